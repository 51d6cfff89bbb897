"""Embedding dim-mode bookkeeping and the additive (t, y, x) coordinate grid.

Mirror of the reference's stemseg/modeling/embedding_utils.py (same public names, incl. the
``creat_`` spelling): get_nb_embedding_dims :4-14, get_nb_free_dims :17-25,
creat_spatiotemporal_grid :28-41, add_spatiotemporal_offset :44-120.
A mode string is a list of axes: 'x','y','t' are coordinate channels (emitted in the order t, y, x),
every 'f' is a free dimension (zero offset).
"""
import torch

_VALID_FOR_DIMS = ("xy", "ff", "xyt", "xyf", "xytf", "xyff", "xytff", "xyfff")
_VALID_FOR_OFFSET = _VALID_FOR_DIMS + ("x", "xyffff")


def get_nb_embedding_dims(mode):
    if mode not in _VALID_FOR_DIMS:
        raise ValueError("Invalid experimental embedding mode: {}".format(mode))
    return len(mode)


def get_nb_free_dims(mode):
    return mode.count("f") if mode in ("xyf", "xytf", "xyff", "xytff", "xyfff") else 0


def grid_axes(mode):
    """Per embedding channel: 0 = no offset, 1 = t, 2 = y, 3 = x (the C-ABI's grid_axis codes)."""
    if mode not in _VALID_FOR_OFFSET:
        raise ValueError("Invalid experimental embedding mode: {}".format(mode))
    if mode == "ff":
        return [0, 0]
    if mode == "x":
        return [3]
    axes = ([1] if "t" in mode else []) + [2, 3]
    return axes + [0] * mode.count("f")


@torch.no_grad()
def creat_spatiotemporal_grid(height, width, time, t_scale, dtype=torch.float32, device="cpu"):
    """Returns broadcast-expanded (t, y, x) grids of shape [time, height, width]; the 1-D vectors are fp32
    linspace over [-max(1, W/H), max(1, W/H)], [-max(1, H/W), ...] and [-t_scale, t_scale]."""
    t, y, x = grid_vectors(height, width, time, t_scale, dtype, device)
    shape = (time, height, width)
    return t[:, None, None].expand(shape), y[None, :, None].expand(shape), x[None, None, :].expand(shape)


@torch.no_grad()
def grid_vectors(height, width, time, t_scale, dtype=torch.float32, device="cpu"):
    xa, ya = max(1., width / float(height)), max(1., height / float(width))
    # linspace is evaluated on the host in fp32 exactly like the reference (:35-37), then moved
    mk = lambda a, n: torch.linspace(-a, a, n, dtype=torch.float32).to(dtype=dtype, device=device)  # noqa: E731
    return mk(float(t_scale), time), mk(ya, height), mk(xa, width)


def add_spatiotemporal_offset(embeddings, time_scale, mode):
    """embeddings [N, C, T, H, W] + coordinate grid on the leading channels (torch ops; the fused HIP heads
    kernel does this in its epilogue -- this function exists for API parity and for callers outside the decoder)."""
    N, C, T, H, W = embeddings.shape
    axes = grid_axes(mode)
    if not any(axes):
        return embeddings
    t, y, x = creat_spatiotemporal_grid(H, W, T, float(time_scale), embeddings.dtype, embeddings.device)
    planes = {1: t, 2: y, 3: x}
    grid = torch.stack([planes[a] if a else torch.zeros_like(x) for a in axes], 0)
    if mode == "x":
        return embeddings + grid[None]
    return embeddings + grid[None].expand(N, -1, -1, -1, -1)

"""Build-time guard for the register hazard of DESIGN.md section 10: inside the MFMA streams of the split-staged convolution tiles no packed
(v_pk_*, *_mix*) and no 64-bit VALU instruction may write a VGPR.  Both classes were seen to land in registers that MFMAs issued just
before still read (results then differ run to run once several pipelines are in flight); plain VALU and LDS returns inside the streams
have been bit-stable all round.  Compiles four representative tiles to gfx950 assembly (hipcc cross-compiles without a GPU)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import isa_lint  # noqa: E402

TILES = ["3, 3, 3, 4, 4, 2, 1, 8, 1, false, 3",        # f16x3, big 3x3x3 tile (two weight phases)
         "1, 1, 1, 32, 4, 2, 2, 4, 8, false, 3",       # f16x3, 256-channel 1x1 tile (lookahead weight staging)
         "1, 3, 3, 8, 4, 2, 1, 8, 1, false, 2",        # bf16x6, big 2-D tile
         "1, 1, 1, 32, 2, 2, 2, 2, 4, false, 2"]       # bf16x6, 128-voxel 1x1 tile


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("cfg", TILES)
def test_no_packed_or_64bit_valu_inside_the_mfma_streams(cfg):
    worst, counts = isa_lint.lint(cfg)
    assert counts.get("streams", 0) >= 2, "no MFMA streams found in the tile's assembly: %s" % (counts,)
    bad = {k: (counts[k], worst.get(k)) for k in counts if k in ("valu-packed", "valu-wide")}
    assert not bad, "hazard-prone VALU inside the MFMA streams of ConvCfg<%s>: %s" % (cfg, bad)

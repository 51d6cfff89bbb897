// Shared host/device helpers for libstemseg_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/stemseg_hip.h"

#if defined(__HIP_DEVICE_COMPILE__) && defined(__AMDGCN_WAVEFRONT_SIZE) && __AMDGCN_WAVEFRONT_SIZE != 64
#error "libstemseg_hip is written for 64-lane wavefronts (gfx950): shuffles, ballots and MFMA layouts assume it"
#endif

namespace stemseg {

void set_error(const char* fmt, ...);

#define SS_CHECK_ARG(cond, ...)                                     \
    do {                                                            \
        if (!(cond)) {                                              \
            ::stemseg::set_error(__VA_ARGS__);                      \
            return STEMSEG_E_INVALID;                               \
        }                                                           \
    } while (0)

#define SS_HIP(call)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            ::stemseg::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),         \
                                 __FILE__, __LINE__);                                           \
            return STEMSEG_E_HIP;                                                               \
        }                                                                                       \
    } while (0)

#define SS_LAUNCH_CHECK() SS_HIP(hipGetLastError())

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// ReLU / max that PROPAGATE NaN like torch's (fmaxf returns the other operand): an overflow upstream -- e.g. an activation beyond
// the f16x3 range -- must reach the head outputs as a non-finite value, never be flushed to a plausible 0 on the way
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float max_keep_nan(float a, float b) { return (a > b || a != a) ? a : b; }

// zero-haloed layout of a logical [C][T][H][W] volume (see stemseg_hip_padded_geometry)
struct PaddedGeom {
    int64_t pitch, ts, cs, total, interior;
    PaddedGeom() = default;
    PaddedGeom(int C, int T, int H, int W) {
        pitch = round_up((int64_t)W + 2, 4);
        ts = (int64_t)(H + 2) * pitch;
        cs = (int64_t)(T + 2) * ts;
        total = (int64_t)C * cs + 64;  // tail slack: tile over-reads past the last row stay inside
        interior = ts + pitch + 1;
    }
};

static inline StemsegVolume make_volume(float* ptr, int64_t cs, int64_t ts, int64_t ys, int C, int T, int H, int W,
                                        int64_t limit) {
    StemsegVolume v;
    v.ptr = ptr; v.c_stride = cs; v.t_stride = ts; v.y_stride = ys; v.C = C; v.T = T; v.H = H; v.W = W; v.limit = limit;
    return v;
}
// dense [C][T][H][W]
static inline StemsegVolume dense_volume(float* ptr, int C, int T, int H, int W) {
    return make_volume(ptr, (int64_t)T * H * W, (int64_t)H * W, W, C, T, H, W, (int64_t)C * T * H * W);
}
// the haloed view (what conv3d reads): extents T+2, H+2, W+2, origin at the buffer base
static inline StemsegVolume padded_halo_view(float* base, int C, int T, int H, int W) {
    PaddedGeom g(C, T, H, W);
    return make_volume(base, g.cs, g.ts, g.pitch, C, T + 2, H + 2, W + 2, g.total);
}
// the interior view (what producers write)
static inline StemsegVolume padded_interior_view(float* base, int C, int T, int H, int W) {
    PaddedGeom g(C, T, H, W);
    return make_volume(base + g.interior, g.cs, g.ts, g.pitch, C, T, H, W, g.total - g.interior);
}

// ---- workspace canaries (with 288 GB nothing is aliased, so nothing but a guard would notice an out-of-slice
// write): every slice of the encoder / decoder workspaces is followed by a guard block of WS_GUARD_FLOATS words holding WS_CANARY (a
// quiet-NaN bit pattern: a stray write of data will not reproduce it, a stray READ poisons whatever consumed it);
// *_init_workspace writes them, stemseg_hip_{encoder,decoder}_check_workspace counts the words that no longer hold it.
constexpr int WS_GUARD_FLOATS = 64;
constexpr uint32_t WS_CANARY = 0x7fc5ca7au;
constexpr int WS_MAX_GUARDS = 64;
struct GuardList {
    int n = 0;
    int64_t off[WS_MAX_GUARDS];   // float offset of every guard block inside the workspace
    bool full = false;            // a plan asked for more slices than the list holds (the plan functions fail on it)
    void push(int64_t o) { if (n < WS_MAX_GUARDS) off[n++] = o; else full = true; }
};
int launch_canary_fill(float* ws, const GuardList& g, hipStream_t s);
// synchronises `s`; *n_bad_host = clobbered guard words, *first_bad_host = float offset of the first one (-1: none)
int canary_check(const float* ws, const GuardList& g, int32_t* n_bad_host, int64_t* first_bad_host, hipStream_t s);

// ---- optional in-library profiler: hipEvent pairs around tagged launches (off by default) -------
void* profile_begin(int tag, double work, hipStream_t s);   // returns NULL when profiling is off
void profile_end(void* handle, hipStream_t s);

// ---- internal kernel launchers shared between api.hip and decoder.hip -------------------------
struct ConvEpilogue {            // fused into the conv epilogue (or the split-K reduce): v = acc + bias (+ res) ; relu
    int relu = 0;
    const float* res = nullptr;  // residual, addressed with (res_cs, res_ts, res_ys) at the output's (c, t, y, x)
    int64_t res_cs = 0, res_ts = 0, res_ys = 0;
    int dec_H = 0, dec_W = 0;    // > 0: 1x1x1 conv launched on a flat [C][V] input; voxel v -> (v / (H*W), (v / W) % H, v % W)
    int precision = 0;           // STEMSEG_PRECISION_F32 | _BF16X6 | _F16X3 (the weights packed for it)
    // Planning shape (conv_igemm.hip, PlanCtx): this launch holds `frames` frames (its T axis, or its flat voxel count / the
    // per-frame voxel count); tile shape and split-K factor are decided as if it held `plan_frames`, with `plan_scratch_floats` of
    // split-K scratch to count on -- so the summation order of every output does not depend on the batch.  0: decide on the real shape.
    int frames = 0, plan_frames = 0;
    int64_t plan_scratch_floats = 0;
    // GroupNorm statistics of the output in the same pass (decoder stages): see launch_conv3d_gn
    double* gn_part = nullptr;   // [Cout / gn_cpg][gn_cap][2] partial (sum, sum of squares) table
    int gn_cpg = 0, gn_cap = 0;
    int* gn_used = nullptr;      // host counter: slots filled by this conv's launches
    // clip batch (decoder stages; blockIdx.y of the launch): nb clips whose input / output volumes and gn_part tables lie in_bs / out_bs
    // floats and gn_bs doubles apart.  Tile shape and split-K factor are decided on one clip, so a clip's result is the same in any batch;
    // the split-K scratch must hold nb times a single clip's slabs.
    int nb = 1;
    int64_t in_bs = 0, out_bs = 0, gn_bs = 0;
    // f16x3: write the output as fp16 pair planes (the fused bottleneck tail's operand form, conv_igemm.hip ConvKParams::out_p16) into p16_out
    // INSTEAD of fp32 into the output volume's memory; *p16_done = 1 when the launch did so, 0 when it fell back to the plain fp32 output
    // (a split-K plan).  Decided on the planning shape like every launch decision.
    unsigned int* p16_out = nullptr;
    int* p16_done = nullptr;
};
int launch_conv3d(const StemsegVolume& in, const float* packed_w, const float* bias, const StemsegVolume& out,
                  int kt, int kh, int kw, int tile_cfg, hipStream_t s, float* splitk_scratch = nullptr, int64_t splitk_scratch_floats = 0,
                  const ConvEpilogue* epi = nullptr);
// conv3 of a bottleneck block (+ bias + identity + ReLU) and conv1 of the next (+ bias + ReLU) in one launch (bottleneck_fused.hip, f16x3)
bool fused_tail_supported(int mid);
int launch_fused_tail(int mid, const unsigned int* x16, const float* w3, const float* b3, const float* res, float* y, const float* w1, const float* b1,
                      const StemsegVolume& z, int dec_H, int dec_W, int64_t V, int form, hipStream_t s);
int launch_gn_stats(const float* x, int C, int64_t S, int groups, float eps, float* stats, double* scratch, hipStream_t s);
// conv + GroupNorm statistics of its output in one pass: the conv's epilogue (or its split-K reduce) leaves per-tile partial
// sums in `gn_scratch` (>= gn_scratch_doubles(Cout, groups) doubles), one more tiny launch turns them into stats[2g] = mean,
// stats[2g+1] = rstd.  Falls back to conv + launch_gn_stats when the group size is not 4 or 8 channels.
constexpr int GN_SLOT_CAP = 32768;      // slots per group (x 16 B x groups = 33.5 MB at 64 groups); larger launches take the separate pass
static inline int64_t gn_scratch_doubles(int Cout, int groups) { return (int64_t)groups * GN_SLOT_CAP * 2 > (int64_t)groups * 128 ? (int64_t)groups * GN_SLOT_CAP * 2 : (int64_t)groups * 128; }
int launch_conv3d_gn(const StemsegVolume& in, const float* packed_w, const float* bias, const StemsegVolume& out, int kt, int kh, int kw,
                     int tile_cfg, hipStream_t s, float* splitk_scratch, int64_t splitk_scratch_floats, const ConvEpilogue* epi, int groups,
                     float eps, float* stats, double* gn_scratch, int64_t stats_bs = 0);     // (stats_bs: floats between the clips' stats of a clip batch, epi->nb)
int launch_gn_identity_stats(float* stats, int groups, hipStream_t s);
// Clip batches (the decoders run all clips of a step in one launch per stage): `nb` clips whose buffers lie at fixed strides (floats;
// doubles for the partial-sum tables) from clip 0's; the clip is the launch's last grid dimension.
struct ClipBatch {
    int nb = 1;
    int64_t in_bs = 0, out_bs = 0, stats_bs = 0;
};
int launch_gn_finalize_slots(const double* part, int groups, int cap, int used, double group_elems, float eps, float* stats, hipStream_t s,
                             int nb = 1, int64_t part_bs = 0, int64_t stats_bs = 0);
int launch_gn_relu_pool(const float* x, int C, int T, int H, int W, int groups, const float* stats, const float* gamma,
                        const float* beta, int pool, const StemsegVolume& out, hipStream_t s, const ClipBatch& cb = ClipBatch());
int launch_upsample(const float* in, int C, int T, int H, int W, int st, int sy, int sx, const StemsegVolume& out,
                    hipStream_t s, const ClipBatch& cb = ClipBatch());
int launch_copy_to_volume(const float* in, int layout, const StemsegVolume& out, hipStream_t s);
int launch_copy_strided(const float* in, int64_t in_c_stride, int64_t in_t_stride, const StemsegVolume& out, hipStream_t s);
struct HeadSpec {
    int n_out;
    int act[STEMSEG_MAX_EMB_DIMS * 2];
    int grid_axis[STEMSEG_MAX_EMB_DIMS * 2];
};
struct HeadsGN {               // the heads read a RAW conv output and apply GroupNorm + affine + ReLU on the fly (stats: (mean, rstd) per group, per clip)
    const float* stats;
    const float* gamma;
    const float* beta;
    int cpg;
    int64_t stats_bs;
};
int launch_heads(const float* x, int Cin, int T, int H, int W, const float* w, const float* bias, const HeadSpec& hs,
                 const float* gt, const float* gy, const float* gx, float* out, hipStream_t s, const ClipBatch& cb = ClipBatch(),
                 const float* add = nullptr, int64_t add_bs = 0, const HeadsGN* gn = nullptr);
int launch_level_head(const float* x, int Cin, int64_t V, const float* w, int n_out, const float* add, float* out, hipStream_t s, int nb, int64_t x_bs,
                      int64_t add_bs, int64_t out_bs);

}  // namespace stemseg

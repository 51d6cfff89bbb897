// Trilinear up-sampling and the layout copy into (zero-haloed) volumes.  HBM-bound streaming kernels.
//
// Reference: UpsampleTrilinear3D.forward = F.interpolate(mode='trilinear', align_corners=False)
// (/root/reference/stemseg/modeling/common.py:77-78, used at embedding_decoder.py:64-79) and the x4 resize of
// OnlineChainer.resize_tensors (inference/online_chainer.py:127-140).  Per axis, for integer scale s:
//   src = max((dst + 0.5) / s - 0.5, 0) ; i0 = floor(src) ; i1 = min(i0 + 1, n - 1) ; w1 = src - i0.
#include "common.h"
#include <algorithm>

namespace stemseg {

struct UpParams {
    const float* in;
    float* out;
    int64_t out_cs, out_ts, out_ys;
    int64_t in_bs, out_bs;   // clip batch (the launch's last grid dimension = clip)
    int C, T, H, W, To, Ho, Wo;
    float rt, ry, rx;   // 1/scale
};

__device__ __forceinline__ void src_index(int dst, float rscale, int n, int& i0, int& i1, float& w1) {
    // ATen area_pixel_compute_source_index(align_corners=false): scale*(dst+0.5)-0.5, clamped at 0
    float src = __fsub_rn(__fmul_rn(rscale, __fadd_rn((float)dst, 0.5f)), 0.5f);
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + ((i0 < n - 1) ? 1 : 0);
    w1 = __fsub_rn(src, (float)i0);
}

__global__ __launch_bounds__(256) void upsample_trilinear_kernel(UpParams p) {
    p.in += (int64_t)blockIdx.y * p.in_bs; p.out += (int64_t)blockIdx.y * p.out_bs;
    const int64_t HWo = (int64_t)p.Ho * p.Wo;
    const int64_t per_c = (int64_t)p.To * HWo;
    const int64_t total = per_c * p.C;
    const int64_t HW = (int64_t)p.H * p.W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / per_c);
        int64_t r = i - (int64_t)c * per_c;
        const int to = (int)(r / HWo);
        r -= (int64_t)to * HWo;
        const int yo = (int)(r / p.Wo), xo = (int)(r - (int64_t)yo * p.Wo);
        int t0, t1, y0, y1, x0, x1;
        float wt, wy, wx;
        src_index(to, p.rt, p.T, t0, t1, wt);
        src_index(yo, p.ry, p.H, y0, y1, wy);
        src_index(xo, p.rx, p.W, x0, x1, wx);
        const float* b = p.in + (int64_t)c * p.T * HW;
        const float* p00 = b + (int64_t)t0 * HW + (int64_t)y0 * p.W;
        const float* p01 = b + (int64_t)t0 * HW + (int64_t)y1 * p.W;
        const float* p10 = b + (int64_t)t1 * HW + (int64_t)y0 * p.W;
        const float* p11 = b + (int64_t)t1 * HW + (int64_t)y1 * p.W;
        const float ut = __fsub_rn(1.f, wt), uy = __fsub_rn(1.f, wy), ux = __fsub_rn(1.f, wx);
        // ATen's CPU kernel: x innermost, then y, then t; each level evaluates  a * wa + b * wb  with the SECOND product rounded
        // and the first fused into the add -- fma(a, wa, round(b * wb)) -- which is what its compiled code does (found by
        // matching all 27 contraction patterns against torch-CPU: this one reproduces it BIT for bit, the unfused form differs
        // in 29 % of the values).  It matters: the clusterer's arg-max over a resized seediness plateau sees the last bit.
        auto lerp = [](float a, float wa, float b, float wb) { return __fmaf_rn(a, wa, __fmul_rn(b, wb)); };
        const float r00 = lerp(p00[x0], ux, p00[x1], wx), r01 = lerp(p01[x0], ux, p01[x1], wx);
        const float r10 = lerp(p10[x0], ux, p10[x1], wx), r11 = lerp(p11[x0], ux, p11[x1], wx);
        const float v = lerp(lerp(r00, uy, r01, wy), ut, lerp(r10, uy, r11, wy), wt);
        p.out[(int64_t)c * p.out_cs + (int64_t)to * p.out_ts + (int64_t)yo * p.out_ys + xo] = v;
    }
}

// 16-B-store form for the x2 / x4 up-samplings of the path (decoder stages: (1|2, 2, 2); --resize_embeddings: (1, 4, 4)):
// one thread = 4 consecutive outputs of one row.  For an integer x-scale SX the 4 outputs 4j .. 4j+3 read the inputs
// j*4/SX - 1 .. (SX = 2: 2j-1 .. 2j+2, four of them; SX = 4: j-1 .. j+1, three), so each of the 4 source rows (t0|t1 x y0|y1)
// is loaded ONCE into registers with clamped indices; the interpolation weights still come from src_index() and every
// lerp keeps the contraction pattern of the scalar kernel, so the result is the same bit for bit (at the borders the
// clamped neighbour enters with weight exactly 0 or duplicates the edge value, as in ATen).  grid.y = (channel, t_out):
// 32-bit index math, one division per thread.
template <int SX>
__global__ __launch_bounds__(256) void upsample_vec4_kernel(UpParams p, unsigned wq, unsigned n_items) {
    static_assert(SX == 2 || SX == 4, "x scale 2 or 4");
    const unsigned item = blockIdx.x * 256u + threadIdx.x;
    if (item >= n_items) return;
    p.in += (int64_t)blockIdx.z * p.in_bs; p.out += (int64_t)blockIdx.z * p.out_bs;
    const int c = blockIdx.y / p.To, to = blockIdx.y - c * p.To;
    const int yo = (int)(item / wq), j = (int)(item - (unsigned)yo * wq);
    int t0, t1, y0, y1;
    float wt, wy;
    src_index(to, p.rt, p.T, t0, t1, wt);
    src_index(yo, p.ry, p.H, y0, y1, wy);
    const int HW = p.H * p.W;
    const float* b = p.in + (int64_t)c * p.T * HW;
    const float* rows[4] = {b + t0 * HW + y0 * p.W, b + t0 * HW + y1 * p.W, b + t1 * HW + y0 * p.W, b + t1 * HW + y1 * p.W};
    constexpr int NV = SX == 2 ? 4 : 3;
    const int xb = (SX == 2 ? 2 * j : j) - 1;                 // first input column this thread needs
    float v[4][NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int xx = min(max(xb + k, 0), p.W - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r][k] = rows[r][xx];
    }
    const float ut = __fsub_rn(1.f, wt), uy = __fsub_rn(1.f, wy);
    auto lerp = [](float a, float wa, float b2, float wb) { return __fmaf_rn(a, wa, __fmul_rn(b2, wb)); };
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int x0, x1;
        float wx;
        src_index(4 * j + k, p.rx, p.W, x0, x1, wx);
        const float ux = __fsub_rn(1.f, wx);
        // position of x0 in v[]: SX = 2 -> outputs 0,1,2,3 start at 2j-1, 2j, 2j, 2j+1; SX = 4 -> j-1, j-1, j, j
        constexpr int r2[4] = {0, 1, 1, 2}, r4[4] = {0, 0, 1, 1};
        const int i0 = SX == 2 ? r2[k] : r4[k];
        const float r00 = lerp(v[0][i0], ux, v[0][i0 + 1], wx), r01 = lerp(v[1][i0], ux, v[1][i0 + 1], wx);
        const float r10 = lerp(v[2][i0], ux, v[2][i0 + 1], wx), r11 = lerp(v[3][i0], ux, v[3][i0 + 1], wx);
        o[k] = lerp(lerp(r00, uy, r01, wy), ut, lerp(r10, uy, r11, wy), wt);
    }
    *reinterpret_cast<float4*>(p.out + (int64_t)c * p.out_cs + (int64_t)to * p.out_ts + (int64_t)yo * p.out_ys + 4 * j) =
        make_float4(o[0], o[1], o[2], o[3]);
}

// Block form of the (1 | 2, 2, 2) up-samplings of the decoders: one thread = output rows 2 my - 1 and 2 my (they interpolate between the SAME two
// input rows) of output planes 2 mt - 1 and 2 mt (ST = 2: the same two input planes; ST = 1: one plane) x 4 columns -- sixteen loads for four
// (ST = 1: two) 16-byte stores, where the one-row form issues sixteen loads per store: the x2 up-sampling of a 60 x 108 map was bound by its
// load instructions, not by its bytes (2.2 TB/s).  Indices and weights come from src_index() per output, every lerp is the scalar kernel's
// fma(a, wa, round(b * wb)): the same bits.  my = 0 holds row 0 alone, my = H row 2 H - 1 alone; likewise the planes.  grid.y = (channel, mt).
template <int ST>
__global__ __launch_bounds__(256) void upsample2_blk_kernel(UpParams p, unsigned wq, unsigned n_items) {
    static_assert(ST == 1 || ST == 2, "temporal scale 1 or 2");
    const unsigned item = blockIdx.x * 256u + threadIdx.x;
    if (item >= n_items) return;
    p.in += (int64_t)blockIdx.z * p.in_bs; p.out += (int64_t)blockIdx.z * p.out_bs;
    constexpr int NP = ST;                                    // output planes per thread
    const int ntp = ST == 2 ? p.T + 1 : p.To;
    const int c = blockIdx.y / ntp, mt = blockIdx.y - c * ntp;
    const int my = (int)(item / wq), j = (int)(item - (unsigned)my * wq);
    int to_[NP], yo_[2];
    bool tv[NP], yv[2];
    if (ST == 2) { to_[0] = 2 * mt - 1; to_[NP - 1] = 2 * mt; tv[0] = mt >= 1; tv[NP - 1] = mt < p.T; }
    else { to_[0] = mt; tv[0] = true; }
    yo_[0] = 2 * my - 1; yo_[1] = 2 * my; yv[0] = my >= 1; yv[1] = my < p.H;
    int t0 = 0, t1 = 0, y0 = 0, y1 = 0;
    float wt[NP], wy[2];
#pragma unroll
    for (int a = NP - 1; a >= 0; --a) {
        wt[a] = 0.f;
        if (tv[a]) src_index(to_[a], p.rt, p.T, t0, t1, wt[a]);      // (both valid: the same (t0, t1))
    }
#pragma unroll
    for (int b = 1; b >= 0; --b) {
        wy[b] = 0.f;
        if (yv[b]) src_index(yo_[b], p.ry, p.H, y0, y1, wy[b]);
    }
    const int HW = p.H * p.W;
    const float* base = p.in + (int64_t)c * p.T * HW;
    const float* rows[4] = {base + t0 * HW + y0 * p.W, base + t0 * HW + y1 * p.W, base + t1 * HW + y0 * p.W, base + t1 * HW + y1 * p.W};
    const int xb = 2 * j - 1;                                 // first input column this thread needs
    float v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xx = min(max(xb + k, 0), p.W - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r][k] = rows[r][xx];
    }
    auto lerp = [](float a, float wa, float b2, float wb) { return __fmaf_rn(a, wa, __fmul_rn(b2, wb)); };
    float r00[4], r01[4], r10[4], r11[4];                     // the x lerps: shared by the rows / planes of the block
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int x0, x1;
        float wx;
        src_index(4 * j + k, p.rx, p.W, x0, x1, wx);
        const float ux = __fsub_rn(1.f, wx);
        constexpr int r2[4] = {0, 1, 1, 2};                   // position of x0 in v[]: outputs 0, 1, 2, 3 start at 2j - 1, 2j, 2j, 2j + 1
        const int i0 = r2[k];
        r00[k] = lerp(v[0][i0], ux, v[0][i0 + 1], wx); r01[k] = lerp(v[1][i0], ux, v[1][i0 + 1], wx);
        r10[k] = lerp(v[2][i0], ux, v[2][i0 + 1], wx); r11[k] = lerp(v[3][i0], ux, v[3][i0 + 1], wx);
    }
#pragma unroll
    for (int a = 0; a < NP; ++a) {
        if (!tv[a]) continue;
        const float ut = __fsub_rn(1.f, wt[a]);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (!yv[b]) continue;
            const float uy = __fsub_rn(1.f, wy[b]);
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = lerp(lerp(r00[k], uy, r01[k], wy[b]), ut, lerp(r10[k], uy, r11[k], wy[b]), wt[a]);
            *reinterpret_cast<float4*>(p.out + (int64_t)c * p.out_cs + (int64_t)to_[a] * p.out_ts + (int64_t)yo_[b] * p.out_ys + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

struct CopyParams {
    const float* in;
    float* out;
    int64_t out_cs, out_ts, out_ys;
    int64_t in_c, in_t;   // input strides in floats for (c, t)
    int C, T, H, W;
};

__global__ __launch_bounds__(256) void copy_to_volume_kernel(const CopyParams p) {
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t per_c = (int64_t)p.T * HW;
    const int64_t total = per_c * p.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / per_c);
        int64_t r = i - (int64_t)c * per_c;
        const int t = (int)(r / HW);
        r -= (int64_t)t * HW;
        const int y = (int)(r / p.W), x = (int)(r - (int64_t)y * p.W);
        p.out[(int64_t)c * p.out_cs + (int64_t)t * p.out_ts + (int64_t)y * p.out_ys + x] =
            p.in[(int64_t)c * p.in_c + (int64_t)t * p.in_t + r];
    }
}

int launch_upsample(const float* in, int C, int T, int H, int W, int st, int sy, int sx, const StemsegVolume& out,
                    hipStream_t s, const ClipBatch& cb) {
    SS_CHECK_ARG(in && out.ptr && cb.nb >= 1 && cb.nb <= 65535 && cb.out_bs % 4 == 0, "upsample: null pointer");
    SS_CHECK_ARG(st >= 1 && sy >= 1 && sx >= 1, "upsample: scale factors must be >= 1");
    SS_CHECK_ARG(out.C == C && out.T == T * st && out.H == H * sy && out.W == W * sx,
                 "upsample: output volume (%d,%d,%d,%d) != (%d,%d,%d,%d)", out.C, out.T, out.H, out.W, C, T * st, H * sy, W * sx);
    UpParams p;
    p.in = in; p.out = out.ptr; p.out_cs = out.c_stride; p.out_ts = out.t_stride; p.out_ys = out.y_stride;
    p.C = C; p.T = T; p.H = H; p.W = W; p.To = T * st; p.Ho = H * sy; p.Wo = W * sx;
    p.rt = 1.0f / (float)st; p.ry = 1.0f / (float)sy; p.rx = 1.0f / (float)sx;
    p.in_bs = cb.in_bs; p.out_bs = cb.out_bs;
    const int64_t total = (int64_t)C * p.To * p.Ho * p.Wo;
    void* ev = profile_begin(40, 4.0 * ((double)C * T * H * W + (double)total) * cb.nb, s);
    const bool vec = (sx == 2 || sx == 4) && p.Wo % 4 == 0 && (int64_t)C * p.To <= 65535 && (int64_t)T * H * W < (1ll << 31) &&
                     (int64_t)p.Ho * p.Wo < (1ll << 32) && (reinterpret_cast<uintptr_t>(out.ptr) % 16 == 0) && out.c_stride % 4 == 0 &&
                     out.t_stride % 4 == 0 && out.y_stride % 4 == 0;
    static const bool blk_on = [] { const char* e = getenv("STEMSEG_UPSAMPLE_BLK"); return !(e && e[0] == '0'); }();      // (A/B switch; default on)
    if (vec && blk_on && sx == 2 && sy == 2 && (st == 1 || st == 2) && (int64_t)C * (st == 2 ? T + 1 : p.To) <= 65535) {
        const unsigned wq = (unsigned)(p.Wo / 4), items = wq * (unsigned)(H + 1);
        const dim3 grid((unsigned)ceil_div(items, 256), (unsigned)(C * (st == 2 ? T + 1 : p.To)), (unsigned)cb.nb);
        if (st == 2) hipLaunchKernelGGL(upsample2_blk_kernel<2>, grid, dim3(256), 0, s, p, wq, items);
        else hipLaunchKernelGGL(upsample2_blk_kernel<1>, grid, dim3(256), 0, s, p, wq, items);
    } else if (vec) {
        const unsigned wq = (unsigned)(p.Wo / 4), items = wq * (unsigned)p.Ho;
        const dim3 grid((unsigned)ceil_div(items, 256), (unsigned)(C * p.To), (unsigned)cb.nb);
        if (sx == 2) hipLaunchKernelGGL(upsample_vec4_kernel<2>, grid, dim3(256), 0, s, p, wq, items);
        else hipLaunchKernelGGL(upsample_vec4_kernel<4>, grid, dim3(256), 0, s, p, wq, items);
    } else {
        const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 256 * 16);
        hipLaunchKernelGGL(upsample_trilinear_kernel, dim3(blocks, (unsigned)cb.nb), dim3(256), 0, s, p);
    }
    profile_end(ev, s);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

int launch_copy_to_volume(const float* in, int layout, const StemsegVolume& out, hipStream_t s) {
    SS_CHECK_ARG(in && out.ptr, "copy_to_volume: null pointer");
    SS_CHECK_ARG(layout == 0 || layout == 1, "copy_to_volume: layout must be 0 ([C][T][H][W]) or 1 ([T][C][H][W])");
    CopyParams p;
    p.in = in; p.out = out.ptr; p.out_cs = out.c_stride; p.out_ts = out.t_stride; p.out_ys = out.y_stride;
    p.C = out.C; p.T = out.T; p.H = out.H; p.W = out.W;
    const int64_t HW = (int64_t)out.H * out.W;
    if (layout == 0) { p.in_c = (int64_t)out.T * HW; p.in_t = HW; }
    else { p.in_c = HW; p.in_t = (int64_t)out.C * HW; }
    const int64_t total = (int64_t)out.C * out.T * HW;
    const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 256 * 16);
    hipLaunchKernelGGL(copy_to_volume_kernel, dim3(blocks), dim3(256), 0, s, p);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

// same kernel, source given by its (channel, t) strides: a window of frames of a larger dense [C][T_all][H][W] tensor
int launch_copy_strided(const float* in, int64_t in_c_stride, int64_t in_t_stride, const StemsegVolume& out, hipStream_t s) {
    SS_CHECK_ARG(in && out.ptr, "copy_strided: null pointer");
    CopyParams p;
    p.in = in; p.out = out.ptr; p.out_cs = out.c_stride; p.out_ts = out.t_stride; p.out_ys = out.y_stride;
    p.C = out.C; p.T = out.T; p.H = out.H; p.W = out.W;
    p.in_c = in_c_stride; p.in_t = in_t_stride;
    const int64_t total = (int64_t)out.C * out.T * out.H * out.W;
    const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 256 * 16);
    hipLaunchKernelGGL(copy_to_volume_kernel, dim3(blocks), dim3(256), 0, s, p);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

}  // namespace stemseg

extern "C" int stemseg_hip_upsample_trilinear(const float* in, int32_t C, int32_t T, int32_t H, int32_t W, int32_t st, int32_t sy,
                                              int32_t sx, const StemsegVolume* out, void* stream) {
    using namespace stemseg;
    SS_CHECK_ARG(out, "upsample: null volume");
    return launch_upsample(in, C, T, H, W, st, sy, sx, *out, as_stream(stream));
}

extern "C" int stemseg_hip_copy_to_volume(const float* in, int32_t layout, const StemsegVolume* out, void* stream) {
    using namespace stemseg;
    SS_CHECK_ARG(out, "copy_to_volume: null volume");
    return launch_copy_to_volume(in, layout, *out, as_stream(stream));
}

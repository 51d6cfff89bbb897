// 2-D encoder (ResNet-50/101 + FPN) on the same fp32-MFMA implicit-GEMM core as the decoder.
//
// Reference: /root/reference/stemseg/modeling/backbone/resnet.py:105-113 (ResNet.forward), :263-282 (Bottleneck,
// stride in the first 1x1), :292-304 (stem + max-pool), fpn.py:47-69 (FPN.forward), make_layers.py:51-63
// (FrozenBatchNorm2d, folded into weight/bias by the host once), model_builder.py:154-169 (run_backbone).
//
// MI355X-first choices
//   * a clip's T frames are the T axis of one [C][T][H][W] volume, so every 2-D convolution is the decoder's conv
//     kernel with KT = 1 and the FPN outputs land directly in the [C][T][h][w] (optionally zero-haloed) layout the 3-D
//     decoders consume -- no per-frame launches, no torch.stack, no pad copy;
//   * bias (= folded BN shift), ReLU and the residual add live in the conv epilogue: a bottleneck block is 3 (4 with a
//     projection shortcut) launches instead of conv + BN + ReLU + add chains;
//   * 1x1 convs run on the flat [C][V] view (aligned 16-B rows regardless of W); when their consumer is a 3x3 conv the
//     epilogue decodes the voxel index and writes the interior of a zero-haloed 2-D layout [C][T][h+2][pitch];
//   * stride-2 1x1 convs (first block of layers 2-4, conv1 and the shortcut) read one shared 2x-subsampled copy;
//   * layers with few voxels (layer3/4: 12 960 / 3 240 voxels) use the conv kernel's split-K to fill 256 CUs.
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace stemseg {

// zero-haloed 2-D layout: [C][T][H+2][pitch], halo only in H and W
struct Padded2D {
    int64_t pitch, ts, cs, total, interior;
    Padded2D() = default;
    Padded2D(int C, int T, int H, int W) {
        pitch = round_up((int64_t)W + 2, 4);
        ts = (int64_t)(H + 2) * pitch;
        cs = (int64_t)T * ts;
        total = (int64_t)C * cs + 64;
        interior = pitch + 1;
    }
};
static inline StemsegVolume halo2d_view(float* base, int C, int T, int H, int W) {
    Padded2D g(C, T, H, W);
    return make_volume(base, g.cs, g.ts, g.pitch, C, T, H + 2, W + 2, g.total);
}
static inline StemsegVolume interior2d_view(float* base, int C, int T, int H, int W) {
    Padded2D g(C, T, H, W);
    return make_volume(base + g.interior, g.cs, g.ts, g.pitch, C, T, H, W, g.total - g.interior);
}
static inline StemsegVolume flat_view(float* base, int C, int64_t V) { return make_volume(base, V, 0, 0, C, 1, 1, (int)V, (int64_t)C * V); }

// stem tile: 8 x 64 output pixels of one frame, all 64 channels
constexpr int ST_ROWS = 8, ST_COLS = 64, ST_PR = 2 * ST_ROWS + 5, ST_PC = 2 * ST_COLS + 5;
#ifdef SS_EXPERIMENTS
constexpr int ST_PCP = 136;
// ---- stem as a direct VALU kernel (rounds 1-3).  NOT in the product library: under several concurrently replaying hipGraphs a
// few of its outputs per ~10^3 launches came back wrong -- 16 lanes of one accumulator register off by a product or two
// (tools/soak_probe.py, DESIGN.md section 10: every one of 253 differing lane-rounds started here, none in an MFMA kernel).  The
// mechanism is unknown: rebuilding it without the AGPR-parked accumulators of its first version changed nothing, two stand-alone
// reproducers stay clean.  Built only with -DSS_EXPERIMENTS (STEMSEG_BUILD_DEFINES), where STEMSEG_STEM=valu selects it for the
// probes that study the effect.  thread = 2 pixels (x, x+32) x 64 channels.
__global__ __launch_bounds__(256, 2) void stem_conv7x7_kernel(const float* __restrict__ frames, const float* __restrict__ w_tap_major,
                                                            const float* __restrict__ bias, float* __restrict__ out, int T, int H, int W) {
    __shared__ __attribute__((aligned(16))) float lds[3 * ST_PR * ST_PCP + 147 * 64];
    float* patch = lds;
    float* wl = lds + 3 * ST_PR * ST_PCP;
    const int Ho = H / 2, Wo = W / 2;
    const int tiles_x = (Wo + ST_COLS - 1) / ST_COLS, tiles_y = (Ho + ST_ROWS - 1) / ST_ROWS;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int t = b / tiles_y;
    const int oy0 = ty * ST_ROWS, ox0 = tx * ST_COLS;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    for (int i = threadIdx.x; i < 147 * 64; i += 256) wl[i] = w_tap_major[i];
    for (int i = threadIdx.x; i < 3 * ST_PR * ST_PC; i += 256) {
        const int xx = i % ST_PC;
        int r = i / ST_PC;
        const int yy = r % ST_PR, c = r / ST_PR;
        const int iy = iy0 + yy, ix = ix0 + xx;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = frames[(((int64_t)t * 3 + c) * H + iy) * W + ix];
        patch[(c * ST_PR + yy) * ST_PCP + xx] = v;
    }
    __syncthreads();
    const int py = threadIdx.x >> 5, px = threadIdx.x & 31;    // pixels (py, px) and (py, px + 32)
    const int oy = oy0 + py;
    const int64_t plane = (int64_t)Ho * Wo;
    // Two passes of 32 output channels: 64 accumulators per pass stay in architectural VGPRs (the one-pass form parked 20 of its
    // 128 accumulators in AGPRs; both forms showed the wrong words, at the same rate: the parking is NOT the cause).
#pragma unroll 1
    for (int hc = 0; hc < 2; ++hc) {
        float acc0[32], acc1[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; }
#pragma unroll 1
        for (int cdy = 0; cdy < 21; ++cdy) {                  // (c, dy) rows of the patch, in the summation order c, dy, dx
                const int c = cdy / 7, dy = cdy - 7 * c;
                const float* prow = patch + (c * ST_PR + 2 * py + dy) * ST_PCP + 2 * px;
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    const float v0 = prow[dx], v1 = prow[dx + 64];
                    const float4* w4 = reinterpret_cast<const float4*>(wl + (cdy * 7 + dx) * 64 + hc * 32);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float4 wv = w4[k];      // broadcast LDS read
                        acc0[4 * k + 0] = fmaf(wv.x, v0, acc0[4 * k + 0]); acc1[4 * k + 0] = fmaf(wv.x, v1, acc1[4 * k + 0]);
                        acc0[4 * k + 1] = fmaf(wv.y, v0, acc0[4 * k + 1]); acc1[4 * k + 1] = fmaf(wv.y, v1, acc1[4 * k + 1]);
                        acc0[4 * k + 2] = fmaf(wv.z, v0, acc0[4 * k + 2]); acc1[4 * k + 2] = fmaf(wv.z, v1, acc1[4 * k + 2]);
                        acc0[4 * k + 3] = fmaf(wv.w, v0, acc0[4 * k + 3]); acc1[4 * k + 3] = fmaf(wv.w, v1, acc1[4 * k + 3]);
                    }
                }
            }
        if (oy < Ho) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int ch = hc * 32 + k;
                float* o = out + ((int64_t)ch * T + t) * plane + (int64_t)oy * Wo;
                const float bv = bias[ch];
                if (ox0 + px < Wo) o[ox0 + px] = relu_keep_nan(acc0[k] + bv);
                if (ox0 + px + 32 < Wo) o[ox0 + px + 32] = relu_keep_nan(acc1[k] + bv);
            }
        }
    }
}

#endif  // SS_EXPERIMENTS

// ---- stem on the matrix cores: the 7x7 stride-2 convolution (3 -> 64, + bias + ReLU) as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32
// products, fp32 accumulation).  M = 64 output channels, N = 8 rows x 64 columns of outputs per workgroup, K = 3 x 7 x 8 taps (the
// eighth column tap is zero padding): an MFMA consumes TWO taps per issue -- lane half 0 an even dx, lane half 1 the odd dx next to
// it -- so the input patch is staged DE-INTERLEAVED by column parity: plane p holds the input columns x = 2 i + p, and tap
// (c, dy, dx = 2 q + p) of output column ox is plane[p][c][2 oy + dy][ox + q]: consecutive lanes read consecutive LDS words and
// every tap is a compile-time immediate on one per-lane base (stride-2 reads of an interleaved patch would be 2-way bank
// conflicts).  Four waves; wave w owns output rows 2w, 2w + 1 (four 32-column blocks) x both 32-channel halves: 8 accumulator
// tiles, 6 LDS reads per 8 MFMAs.  (Through round 3 the stem was a VALU kernel -- the step's only VALU-bound one, and the only kernel
// whose results were not bit-stable under concurrently replaying graphs; this form is also 1.5x faster.)
constexpr int SM_PW = 68, SM_ROWS_IN = 2 * ST_ROWS + 6;          // plane row pitch (words); staged input rows (one spare for nothing: 21 used)
constexpr int SM_PLANE = 3 * SM_ROWS_IN * SM_PW;                  // words per parity plane
constexpr int SM_KSTEPS = 3 * 7 * 4;                              // (c, dy, q): two taps each
typedef float f32x16s __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void stem_conv7x7_mfma_kernel(const float* __restrict__ frames, const float* __restrict__ w_tap_major,
                                                                 const float* __restrict__ bias, float* __restrict__ out, int T, int H, int W) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SM_PLANE + SM_KSTEPS * 2 * 64];
    float* const planes = lds;
    float* const wl = lds + 2 * SM_PLANE;
    const int Ho = H / 2, Wo = W / 2;
    const int tiles_x = (Wo + ST_COLS - 1) / ST_COLS, tiles_y = (Ho + ST_ROWS - 1) / ST_ROWS;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int t = b / tiles_y;
    const int oy0 = ty * ST_ROWS, ox0 = tx * ST_COLS;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    // weights: [step = (c * 7 + dy) * 4 + q][half][64] <- w_tap_major[(c * 7 + dy) * 7 + 2 q + half][64], zero for dx = 7
    for (int i = threadIdx.x; i < SM_KSTEPS * 2 * 64; i += 256) {
        const int co = i & 63, half = (i >> 6) & 1, step = i >> 7;
        const int q = step & 3, cdy = step >> 2, dx = 2 * q + half;
        wl[i] = dx < 7 ? w_tap_major[(cdy * 7 + dx) * 64 + co] : 0.f;
    }
    // input patch, 21 rows x 133 columns per channel, split by column parity (columns beyond 133 and the spare rows: zero)
    for (int i = threadIdx.x; i < 2 * SM_PLANE; i += 256) {
        const int par = i / SM_PLANE, r = i - par * SM_PLANE;
        const int xi = r % SM_PW, yy = (r / SM_PW) % SM_ROWS_IN, c = r / (SM_PW * SM_ROWS_IN);
        const int iy = iy0 + yy, ix = ix0 + 2 * xi + par;
        float v = 0.f;
        if (yy < ST_PR && 2 * xi + par < ST_PC && iy >= 0 && iy < H && ix >= 0 && ix < W) v = frames[(((int64_t)t * 3 + c) * H + iy) * W + ix];
        planes[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    f32x16s acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const float* a_ptr = wl + half * 64 + l31;
    const float* b_ptr[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b_ptr[ni] = planes + half * SM_PLANE + (2 * (2 * wave + (ni >> 1))) * SM_PW + (ni & 1) * 32 + l31;
    auto ld = [&](const int step, float (&a)[2], float (&bv)[4]) __attribute__((always_inline)) {
        const int q = step & 3, cdy = step >> 2, c = cdy / 7, dy = cdy - 7 * c;
        const int boff = (c * SM_ROWS_IN + dy) * SM_PW + q;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[mi] = a_ptr[step * 128 + mi * 32];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bv[ni] = b_ptr[ni][boff];
    };
    auto mm = [&](const float (&a)[2], const float (&bv)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bv[ni], acc[mi][ni], 0, 0, 0);
    };
    // software pipeline over the k-steps (operands of step i + 1 requested before the MFMAs of step i issue)
    float a0[2], b0[4], a1[2], b1[4];
    ld(0, a0, b0);
#pragma unroll
    for (int i = 0; i < SM_KSTEPS; i += 2) {
        ld(i + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mm(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 2 < SM_KSTEPS) ld(i + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mm(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: C/D layout col = lane & 31 (output column), row = (r & 3) + 8 * (r >> 2) + 4 * half (channel)
    const int64_t plane = (int64_t)Ho * Wo;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int oy = oy0 + 2 * wave + (ni >> 1), ox = ox0 + (ni & 1) * 32 + l31;
        if (oy < Ho && ox < Wo) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    out[((int64_t)ch * T + t) * plane + (int64_t)oy * Wo + ox] = relu_keep_nan(acc[mi][ni][r] + bias[ch]);
                }
        }
    }
}

// ---- the stem in f16x3 mode: space-to-depth, then a stride-1 4x4 convolution on the split-staged MFMA kernel (conv_igemm.hip, Y4Stem).
// out[oy][ox] = sum w[c][ky][kx] in[c][2 oy + ky - 3][2 ox + kx - 3]; with ky + 1 = 2 a + p, kx + 1 = 2 b + q (a, b in 0..3, p, q in 0..1; the
// index 0 is a zero tap) the input index is 2 (oy - 2 + a) + p: a 4x4 convolution over S[(p, q, c)][Y][X] = in[c][2 Y + p][2 X + q] with two
// halo positions before and one after.  12 channels (one 16-channel chunk), K = 16 taps x 16 = 256 per output in three fp16 products:
// 3/8 of the matrix time of the exact fp32-MFMA stem (84 steps of K = 2), which the default mode no longer needs: every other convolution of
// the mode carries 22-bit operands as well.  One thread = one input pixel pair (q = 0, 1) of one (t, c, row).
__global__ __launch_bounds__(256) void stem_s2d_kernel(const float* __restrict__ frames, float* __restrict__ s2d, int T, int H, int W, int64_t ts, int pitch) {
    const int W2 = W / 2, H2 = H / 2;
    const int64_t total = (int64_t)T * 3 * H * W2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % W2);
        int64_t r = i / W2;
        const int y = (int)(r % H);
        r /= H;
        const int c = (int)(r % 3), t = (int)(r / 3);
        const float2 v = *reinterpret_cast<const float2*>(frames + (((int64_t)t * 3 + c) * H + y) * W + 2 * X);
        const int p = y & 1, Y = y >> 1;
        (void)H2;
        float* o = s2d + ((int64_t)((p * 2 + 0) * 3 + c) * T + t) * ts + (int64_t)(Y + 2) * pitch + X + 2;
        o[0] = v.x;
        o[(int64_t)3 * T * ts] = v.y;                          // q = 1: three channels further
    }
}

// max-pool 3x3 stride 2 pad 1 over every [c][t] plane (resnet.py:303)
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = planes * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo);
        int64_t r = i / Wo;
        const int y = (int)(r % Ho);
        const int64_t pl = r / Ho;
        const float* p = in + pl * H * W;
        float m = -INFINITY;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = 2 * y + dy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = 2 * x + dx;
                if (xx < 0 || xx >= W) continue;
                m = max_keep_nan(m, p[(int64_t)yy * W + xx]);
            }
        }
        out[i] = m;
    }
}

// 16-B forms (W % 8 == 0, aligned planes): one thread = 4 consecutive outputs of one row; 32-bit index arithmetic (n_items =
// output rows over all planes x float4 pieces per row).
__global__ __launch_bounds__(256) void maxpool3x3s2_vec4_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, unsigned n_items) {
    const int Ho = H / 2, Wo = W / 2;
    const unsigned wq = Wo / 4, idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= n_items) return;
    const unsigned row = idx / wq;                         // plane * Ho + y
    const int j = (int)(idx - row * wq);
    const unsigned pl = row / (unsigned)Ho, y = row - pl * (unsigned)Ho;
    const float* p = in + (size_t)pl * H * W;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = 2 * (int)y + dy;
        if (yy < 0 || yy >= H) continue;
        const float* r = p + (size_t)yy * W + 8 * j;       // inputs 8j-1 .. 8j+7 feed outputs 4j .. 4j+3
        const float4 a = *reinterpret_cast<const float4*>(r), b = *reinterpret_cast<const float4*>(r + 4);
        const float left = j > 0 ? r[-1] : -INFINITY;
        m[0] = max_keep_nan(m[0], max_keep_nan(left, max_keep_nan(a.x, a.y)));
        m[1] = max_keep_nan(m[1], max_keep_nan(a.y, max_keep_nan(a.z, a.w)));
        m[2] = max_keep_nan(m[2], max_keep_nan(a.w, max_keep_nan(b.x, b.y)));
        m[3] = max_keep_nan(m[3], max_keep_nan(b.y, max_keep_nan(b.z, b.w)));
    }
    *reinterpret_cast<float4*>(out + (size_t)row * Wo + 4 * j) = make_float4(m[0], m[1], m[2], m[3]);
}
__global__ __launch_bounds__(256) void subsample2_vec4_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, unsigned n_items) {
    const int Ho = H / 2, Wo = W / 2;
    const unsigned wq = Wo / 4, idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= n_items) return;
    const unsigned row = idx / wq;
    const int j = (int)(idx - row * wq);
    const unsigned pl = row / (unsigned)Ho, y = row - pl * (unsigned)Ho;
    const float* r = in + (size_t)pl * H * W + (size_t)(2 * y) * W + 8 * j;
    const float4 a = *reinterpret_cast<const float4*>(r), b = *reinterpret_cast<const float4*>(r + 4);
    *reinterpret_cast<float4*>(out + (size_t)row * Wo + 4 * j) = make_float4(a.x, a.z, b.x, b.z);
}

// out[pl][y][x] = in[pl][2y][2x]   (shared input of a stride-2 block's conv1 and projection shortcut)
__global__ __launch_bounds__(256) void subsample2_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = planes * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo);
        int64_t r = i / Wo;
        const int y = (int)(r % Ho);
        const int64_t pl = r / Ho;
        out[i] = in[pl * H * W + (int64_t)(2 * y) * W + 2 * x];
    }
}

// The bilinear blend of the FPN top-down path with its rounding steps spelled out -- fma(1 - wx, c[x0], wx c[x1]) per coarse row, then
// fma(1 - wy, row0, wy row1): what the compiler's contraction made of the plain expression, fixed here so that the three forms below agree by
// construction, not by the optimiser's choice (the two-row form shares the row blends between two outputs: a value with two uses contracts
// differently).
__device__ __forceinline__ float up2_blend(const float c00, const float c01, const float c10, const float c11, const float wx, const float wy) {
    const float h0 = __fmaf_rn(1.f - wx, c00, __fmul_rn(wx, c01));
    const float h1 = __fmaf_rn(1.f - wx, c10, __fmul_rn(wx, c11));
    return __fmaf_rn(1.f - wy, h0, __fmul_rn(wy, h1));
}

// FPN top-down path (fpn.py:64-66): fine += bilinear_x2(coarse), both zero-haloed 2-D layouts, align_corners = False
__global__ __launch_bounds__(256) void upsample2x_add_kernel(float* __restrict__ fine, const float* __restrict__ coarse, int64_t planes, int H, int W,
                                                              int64_t f_ts, int64_t f_pitch, int64_t c_ts, int64_t c_pitch) {
    const int Hc = H / 2, Wc = W / 2;
    const int64_t total = planes * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        int64_t r = i / W;
        const int y = (int)(r % H);
        const int64_t pl = r / H;
        float sy = 0.5f * ((float)y + 0.5f) - 0.5f, sx = 0.5f * ((float)x + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < Hc - 1 ? 1 : 0), x1 = x0 + (x0 < Wc - 1 ? 1 : 0);
        const float wy = sy - (float)y0, wx = sx - (float)x0;
        const float* c = coarse + pl * c_ts;
        const float v = up2_blend(c[y0 * c_pitch + x0], c[y0 * c_pitch + x1], c[y1 * c_pitch + x0], c[y1 * c_pitch + x1], wx, wy);
        float* fo = fine + pl * f_ts + (int64_t)y * f_pitch + x;
        *fo = __fadd_rn(*fo, v);
    }
}

// 16-B form: one thread = one ALIGNED group of four columns of the zero-haloed fine row (haloed columns 4k .. 4k + 3 = map columns 4k - 1 .. 4k + 2:
// the interior starts at haloed column 1, so aligned groups straddle it by one).  The fine row is read and written as one float4 (a halo word
// in the group is written back unchanged: zero), the two coarse rows contribute three columns each (2k - 1 .. 2k + 1, clamped exactly as the scalar
// form clamps them), and every output evaluates the SAME expression as the scalar form, so the bits are the same.  Needs pitch % 4 == 0 and a
// 16-B aligned plane base (the encoder's zero-haloed buffers are).  n_items = planes x H x groups per row.
__global__ __launch_bounds__(256) void upsample2x_add_vec4_kernel(float* __restrict__ fine_halo, const float* __restrict__ coarse, int H, int W, unsigned gq,
                                                                   unsigned n_items, int64_t f_ts, int f_pitch, int64_t c_ts, int c_pitch) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= n_items) return;
    const unsigned row = idx / gq;                           // plane * H + y
    const int k = (int)(idx - row * gq);
    const unsigned pl = row / (unsigned)H;
    const int y = (int)(row - pl * (unsigned)H);
    const int Hc = H / 2, Wc = W / 2;
    float sy = 0.5f * ((float)y + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    const int y0 = (int)sy, y1 = y0 + (y0 < Hc - 1 ? 1 : 0);
    const float wy = sy - (float)y0;
    const float* c0 = coarse + (int64_t)pl * c_ts + (int64_t)y0 * c_pitch;
    const float* c1 = coarse + (int64_t)pl * c_ts + (int64_t)y1 * c_pitch;
    float4* fp = reinterpret_cast<float4*>(fine_halo + (int64_t)pl * f_ts + (int64_t)(y + 1) * f_pitch + 4 * k);
    float4 f = *fp;
    float* fv = reinterpret_cast<float*>(&f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = 4 * k - 1 + j;
        if (x < 0 || x >= W) continue;
        float sx = 0.5f * ((float)x + 0.5f) - 0.5f;
        sx = sx < 0.f ? 0.f : sx;
        const int x0 = (int)sx, x1 = x0 + (x0 < Wc - 1 ? 1 : 0);
        const float wx = sx - (float)x0;
        fv[j] = __fadd_rn(fv[j], up2_blend(c0[x0], c0[x1], c1[x0], c1[x1], wx, wy));
    }
    *fp = f;
}

// Two fine rows per thread: rows 2m - 1 and 2m interpolate between the SAME coarse rows (m - 1, m), and the four columns of an aligned group
// between coarse columns 2k - 1, 2k, 2k + 1: six coarse loads, two 16-byte loads and two 16-byte stores for eight outputs, where the one-row form
// issues eighteen memory instructions for four (it re-loads all four corners per output).  m = 0 holds row 0 alone (its row - 1 does not exist), m = Hc
// row H - 1 alone.  Every output goes through up2_blend with the (x0, x1, wx) / (y0, y1, wy) the scalar form computes for it -- for even H and W
// these are exact quarters: x = 2n + 1 -> (n, .25), x = 2n + 2 -> (n, .75), x = 0 -> (0, 0), clamped at the far edge -- so the bits are the scalar
// form's.  n_items = planes x (H / 2 + 1) x groups per row.
__global__ __launch_bounds__(256) void upsample2x_add_vec4x2_kernel(float* __restrict__ fine_halo, const float* __restrict__ coarse, int H, int W, unsigned gq,
                                                                     unsigned n_items, int64_t f_ts, int f_pitch, int64_t c_ts, int c_pitch) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= n_items) return;
    const int Hc = H / 2, Wc = W / 2;
    const unsigned rowp = idx / gq;                          // plane * (Hc + 1) + m
    const int k = (int)(idx - rowp * gq);
    const unsigned pl = rowp / (unsigned)(Hc + 1);
    const int m = (int)(rowp - pl * (unsigned)(Hc + 1));
    const bool row_a = m >= 1, row_b = m < Hc;               // rows 2m - 1, 2m
    const int y0 = m >= 1 ? m - 1 : 0, y1 = y0 + (y0 < Hc - 1 ? 1 : 0);
    const float wy_a = 0.25f, wy_b = m >= 1 ? 0.75f : 0.f;
    const float* c0 = coarse + (int64_t)pl * c_ts + (int64_t)y0 * c_pitch;
    const float* c1 = coarse + (int64_t)pl * c_ts + (int64_t)y1 * c_pitch;
    const int xa = min(max(2 * k - 1, 0), Wc - 1), xb = min(2 * k, Wc - 1), xc = min(2 * k + 1, Wc - 1);
    const float a0 = c0[xa], b0 = c0[xb], d0 = c0[xc], a1 = c1[xa], b1 = c1[xb], d1 = c1[xc];
    float* fbase = fine_halo + (int64_t)pl * f_ts + 4 * k;
    float4 fa = {0.f, 0.f, 0.f, 0.f}, fb = {0.f, 0.f, 0.f, 0.f};
    if (row_a) fa = *reinterpret_cast<const float4*>(fbase + (int64_t)(2 * m) * f_pitch);            // (row y sits at haloed row y + 1)
    if (row_b) fb = *reinterpret_cast<const float4*>(fbase + (int64_t)(2 * m + 1) * f_pitch);
    float* va = reinterpret_cast<float*>(&fa);
    float* vb = reinterpret_cast<float*>(&fb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = 4 * k - 1 + j;
        if (x < 0 || x >= W) continue;
        // corners (c[x0], c[x1]) of the two coarse rows and wx: j = 0, 1 -> columns (2k - 1, 2k), j = 2, 3 -> (2k, 2k + 1); x = 0 -> (0, 1) with wx = 0
        const bool first = j < 2 && x > 0;
        const float p0 = first ? a0 : (x == 0 ? a0 : b0), p1 = first ? b0 : d0;
        const float q0 = first ? a1 : (x == 0 ? a1 : b1), q1 = first ? b1 : d1;
        const float wx = x == 0 ? 0.f : ((x & 1) ? 0.25f : 0.75f);
        if (row_a) va[j] = __fadd_rn(va[j], up2_blend(p0, p1, q0, q1, wx, wy_a));
        if (row_b) vb[j] = __fadd_rn(vb[j], up2_blend(p0, p1, q0, q1, wx, wy_b));
    }
    if (row_a) *reinterpret_cast<float4*>(fbase + (int64_t)(2 * m) * f_pitch) = fa;
    if (row_b) *reinterpret_cast<float4*>(fbase + (int64_t)(2 * m + 1) * f_pitch) = fb;
}

// The same two-row form reading the fine term from a DENSE [planes][H][W] map (the lateral conv's output written by the 16-byte epilogue; its
// by-element epilogue into the zero-haloed layout issued four times the store instructions and ran at 2.2 TB/s) and writing the zero-haloed
// consumer layout: the layout change rides on the pass that touches every element anyway.  A group's four map columns 4k - 1 .. 4k + 2 straddle two
// aligned dense groups: two 16-byte loads per row (the first is the neighbouring thread's second: an L1 hit).  Halo words of a group are written
// as zero.  Same expression per output as the in-place form => the same bits.
__global__ __launch_bounds__(256) void upsample2x_add_dense_vec4x2_kernel(float* __restrict__ fine_halo, const float* __restrict__ fine_dense,
                                                                           const float* __restrict__ coarse, int H, int W, unsigned gq, unsigned n_items,
                                                                           int64_t f_ts, int f_pitch, int64_t c_ts, int c_pitch) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= n_items) return;
    const int Hc = H / 2, Wc = W / 2;
    const unsigned rowp = idx / gq;                          // plane * (Hc + 1) + m
    const int k = (int)(idx - rowp * gq);
    const unsigned pl = rowp / (unsigned)(Hc + 1);
    const int m = (int)(rowp - pl * (unsigned)(Hc + 1));
    const bool row_a = m >= 1, row_b = m < Hc;               // rows 2m - 1, 2m
    const int y0 = m >= 1 ? m - 1 : 0, y1 = y0 + (y0 < Hc - 1 ? 1 : 0);
    const float wy_a = 0.25f, wy_b = m >= 1 ? 0.75f : 0.f;
    const float* c0 = coarse + (int64_t)pl * c_ts + (int64_t)y0 * c_pitch;
    const float* c1 = coarse + (int64_t)pl * c_ts + (int64_t)y1 * c_pitch;
    const int xa = min(max(2 * k - 1, 0), Wc - 1), xb = min(2 * k, Wc - 1), xc = min(2 * k + 1, Wc - 1);
    const float a0 = c0[xa], b0 = c0[xb], d0 = c0[xc], a1 = c1[xa], b1 = c1[xb], d1 = c1[xc];
    const float* dbase = fine_dense + (int64_t)pl * H * W;
    float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f};
    // map columns 4k - 1 .. 4k + 2 of rows 2m - 1 / 2m: the last word of dense group k - 1 and the first three of group k (W % 4 == 0)
    const bool g_lo = k >= 1, g_hi = 4 * k < W;
    if (row_a) {
        const float* r = dbase + (int64_t)(2 * m - 1) * W;
        if (g_lo) va[0] = reinterpret_cast<const float4*>(r)[k - 1].w;
        if (g_hi) { const float4 q = reinterpret_cast<const float4*>(r)[k]; va[1] = q.x; va[2] = q.y; va[3] = q.z; }
    }
    if (row_b) {
        const float* r = dbase + (int64_t)(2 * m) * W;
        if (g_lo) vb[0] = reinterpret_cast<const float4*>(r)[k - 1].w;
        if (g_hi) { const float4 q = reinterpret_cast<const float4*>(r)[k]; vb[1] = q.x; vb[2] = q.y; vb[3] = q.z; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = 4 * k - 1 + j;
        if (x < 0 || x >= W) { va[j] = 0.f; vb[j] = 0.f; continue; }      // halo words of the group
        const bool first = j < 2 && x > 0;
        const float p0 = first ? a0 : (x == 0 ? a0 : b0), p1 = first ? b0 : d0;
        const float q0 = first ? a1 : (x == 0 ? a1 : b1), q1 = first ? b1 : d1;
        const float wx = x == 0 ? 0.f : ((x & 1) ? 0.25f : 0.75f);
        if (row_a) va[j] = __fadd_rn(va[j], up2_blend(p0, p1, q0, q1, wx, wy_a));
        if (row_b) vb[j] = __fadd_rn(vb[j], up2_blend(p0, p1, q0, q1, wx, wy_b));
    }
    float* fbase = fine_halo + (int64_t)pl * f_ts + 4 * k;
    if (row_a) *reinterpret_cast<float4*>(fbase + (int64_t)(2 * m) * f_pitch) = make_float4(va[0], va[1], va[2], va[3]);            // (row y sits at haloed row y + 1)
    if (row_b) *reinterpret_cast<float4*>(fbase + (int64_t)(2 * m + 1) * f_pitch) = make_float4(vb[0], vb[1], vb[2], vb[3]);
}

static int grid1d(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, 256), 256 * 16)); }

struct EncoderPlan {
    GuardList guards;          // one guard block behind every slice below (common.h: workspace canaries)
    int T, H, W, nblk[4], total_blocks;
    int h[4], w[4];            // 4x, 8x, 16x, 32x
    int64_t V[4];
    int64_t S0, X1, A, B, Cst[4], M1[4], M2, DS, XS, L[4], FO[4], SK, SKfloats, total;
    int64_t S2D, s2d_ts, s2d_pitch;
    int plan_frames;           // 0: launches decide on their real shape
    int64_t plan_SKfloats;     // split-K scratch a planned launch may count on
};
constexpr int64_t ENC_PLAN_SK_FLOATS = 32ll << 20;

static int make_encoder_plan(const StemsegEncoderDesc* d, EncoderPlan& p) {
    SS_CHECK_ARG(d, "encoder: null descriptor");
    SS_CHECK_ARG(d->struct_bytes == (int32_t)sizeof(StemsegEncoderDesc), "encoder: descriptor size mismatch (%d vs %d): ABI skew",
                 d->struct_bytes, (int)sizeof(StemsegEncoderDesc));
    SS_CHECK_ARG(d->n_clips >= 1 && d->clip_frames >= 0 && d->clip_stride >= 0, "encoder: bad clip layout");
    if (d->clip_frames > 0) {
        SS_CHECK_ARG(d->clip_stride > 0 && (d->n_clips - 1) * d->clip_stride + d->clip_frames == d->T,
                     "encoder: %d windows of %d frames every %d frames do not cover T=%d", d->n_clips, d->clip_frames, d->clip_stride, d->T);
    } else {
        SS_CHECK_ARG(d->T % d->n_clips == 0, "encoder: T=%d is not n_clips=%d whole clips", d->T, d->n_clips);
    }
    SS_CHECK_ARG(d->T >= 1 && d->H >= 32 && d->W >= 32 && d->H % 32 == 0 && d->W % 32 == 0, "encoder: T=%d H=%d W=%d (H, W multiples of 32)", d->T, d->H, d->W);
    p.T = d->T; p.H = d->H; p.W = d->W;
    p.total_blocks = 0;
    for (int i = 0; i < 4; ++i) {
        SS_CHECK_ARG(d->blocks[i] >= 1 && d->blocks[i] <= 64, "encoder: blocks[%d]=%d", i, d->blocks[i]);
        p.nblk[i] = d->blocks[i];
        p.total_blocks += d->blocks[i];
        p.h[i] = d->H >> (2 + i); p.w[i] = d->W >> (2 + i);
        p.V[i] = (int64_t)d->T * p.h[i] * p.w[i];
    }
    SS_CHECK_ARG(p.total_blocks <= STEMSEG_MAX_ENCODER_BLOCKS, "encoder: more than %d bottleneck blocks", STEMSEG_MAX_ENCODER_BLOCKS);
    int64_t off = 0;
    p.guards.n = 0;
    auto take = [&](int64_t floats) {            // a slice + its guard block
        int64_t o = off;
        off += round_up(floats, 64);
        p.guards.push(off);
        off += WS_GUARD_FLOATS;
        return o;
    };
    p.S0 = take(64 * 4 * p.V[0]);
    // space-to-depth image of the frames for the f16x3 stem: [12][T][H/2 + 3][pitch], two halo rows / columns before, one after (zero, written once)
    p.s2d_pitch = round_up(d->W / 2 + 3, 4);
    p.s2d_ts = (int64_t)(d->H / 2 + 3) * p.s2d_pitch;
    p.S2D = take(12 * (int64_t)d->T * p.s2d_ts + 64);
    p.X1 = take(64 * p.V[0]);
    p.A = take(256 * p.V[0]);
    p.B = take(256 * p.V[0]);
    for (int i = 0; i < 4; ++i) {
        p.Cst[i] = take((int64_t)(256 << i) * p.V[i]);
        // one zero-haloed buffer PER STAGE: the halo stays zero only while a buffer keeps one geometry
        p.M1[i] = take(Padded2D(64 << i, p.T, p.h[i], p.w[i]).total);
    }
    p.M2 = take(64 * p.V[0]);
    p.DS = take(256 * p.V[0]);
    p.XS = take(256 * p.V[1]);
    for (int i = 0; i < 4; ++i) p.L[i] = take(Padded2D(256, p.T, p.h[i], p.w[i]).total);
    // overlapping windows: the FPN output convs run ONCE over all frames into dense maps, each clip's window is then copied out
    const bool shared_out = d->clip_frames > 0 && d->clip_stride < d->clip_frames && d->n_clips > 1;
    for (int i = 0; i < 4; ++i) p.FO[i] = shared_out ? take(256 * p.V[i]) : -1;
    SS_CHECK_ARG(d->plan_frames >= 0 && d->plan_frames <= 4096, "encoder: plan_frames=%d", d->plan_frames);
    p.plan_frames = d->plan_frames;
    p.plan_SKfloats = ENC_PLAN_SK_FLOATS;
    // a pass of more frames than planned keeps the plan's K-partition, so its slabs are T / plan_frames times the planned ones
    p.SKfloats = ENC_PLAN_SK_FLOATS * (d->plan_frames > 0 ? std::max<int64_t>(1, ceil_div(d->T, d->plan_frames)) : 1);
    p.SK = take(p.SKfloats);
    p.total = off;
    SS_CHECK_ARG(!p.guards.full, "encoder: the plan has more than %d workspace slices (GuardList)", WS_MAX_GUARDS);
    return STEMSEG_OK;
}

}  // namespace stemseg

using namespace stemseg;

extern "C" size_t stemseg_hip_encoder_workspace_bytes(const StemsegEncoderDesc* desc) {
    EncoderPlan p;
    if (make_encoder_plan(desc, p) != STEMSEG_OK) return 0;
    return (size_t)p.total * sizeof(float);
}

// Debugging aid (tests, tools): float offsets of the plan's buffers inside the workspace, in the order S0, X1, A, B, Cst[4], M1[4], M2, DS,
// XS, L[4], FO[4], SK, total (25 values; -1 = not present).
extern "C" int stemseg_hip_encoder_plan_offsets(const StemsegEncoderDesc* desc, int64_t* out25) {
    EncoderPlan p;
    int rc = make_encoder_plan(desc, p);
    if (rc) return rc;
    SS_CHECK_ARG(out25, "encoder_plan_offsets: null pointer");
    int n = 0;
    out25[n++] = p.S0; out25[n++] = p.X1; out25[n++] = p.A; out25[n++] = p.B;
    for (int i = 0; i < 4; ++i) out25[n++] = p.Cst[i];
    for (int i = 0; i < 4; ++i) out25[n++] = p.M1[i];
    out25[n++] = p.M2; out25[n++] = p.DS; out25[n++] = p.XS;
    for (int i = 0; i < 4; ++i) out25[n++] = p.L[i];
    for (int i = 0; i < 4; ++i) out25[n++] = p.FO[i];
    out25[n++] = p.SK; out25[n++] = p.total;
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_encoder_init_workspace(const StemsegEncoderDesc* desc, void* workspace, size_t ws_bytes, void* stream) {
    EncoderPlan p;
    int rc = make_encoder_plan(desc, p);
    if (rc) return rc;
    SS_CHECK_ARG(workspace && (reinterpret_cast<uintptr_t>(workspace) % 256 == 0), "encoder: workspace must be 256-byte aligned");
    if (ws_bytes < (size_t)p.total * sizeof(float)) {
        set_error("encoder: workspace too small (%zu < %zu bytes)", ws_bytes, (size_t)p.total * sizeof(float));
        return STEMSEG_E_WORKSPACE;
    }
    // only the zero-haloed buffers need it, but one memset keeps the rule simple
    SS_HIP(hipMemsetAsync(workspace, 0, (size_t)p.total * sizeof(float), as_stream(stream)));
    return launch_canary_fill(reinterpret_cast<float*>(workspace), p.guards, as_stream(stream));
}

extern "C" int stemseg_hip_encoder_check_workspace(const StemsegEncoderDesc* desc, const void* workspace, size_t ws_bytes, int32_t* n_bad_host,
                                                   int64_t* first_bad_host, void* stream) {
    EncoderPlan p;
    int rc = make_encoder_plan(desc, p);
    if (rc) return rc;
    SS_CHECK_ARG(workspace && ws_bytes >= (size_t)p.total * sizeof(float), "encoder_check_workspace: bad workspace");
    return canary_check(reinterpret_cast<const float*>(workspace), p.guards, n_bad_host, first_bad_host, as_stream(stream));
}

#ifdef SS_EXPERIMENTS
// (experiment builds only) STEMSEG_STEM=valu selects the VALU form for the co-residency probes
static bool stem_on_mfma() {
    static const bool on = [] { const char* e = getenv("STEMSEG_STEM"); return !(e && e[0] == 'v'); }();
    return on;
}
#define SS_LAUNCH_STEM(blocks, s, ...)                                                                                   \
    do {                                                                                                                 \
        if (stem_on_mfma()) hipLaunchKernelGGL(stem_conv7x7_mfma_kernel, dim3((unsigned)(blocks)), dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL(stem_conv7x7_kernel, dim3((unsigned)(blocks)), dim3(256), 0, s, __VA_ARGS__);             \
    } while (0)
#else
#define SS_LAUNCH_STEM(blocks, s, ...) hipLaunchKernelGGL(stem_conv7x7_mfma_kernel, dim3((unsigned)(blocks)), dim3(256), 0, s, __VA_ARGS__)
#endif

extern "C" int stemseg_hip_stem_conv(const float* frames, const float* w_tap_major, const float* bias, float* out, int32_t T, int32_t H, int32_t W,
                                     void* stream) {
    SS_CHECK_ARG(frames && w_tap_major && bias && out, "stem_conv: null pointer");
    SS_CHECK_ARG(T >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "stem_conv: T=%d H=%d W=%d (H, W even)", T, H, W);
    const int Ho = H / 2, Wo = W / 2;
    const int64_t blocks = ceil_div(Wo, ST_COLS) * ceil_div(Ho, ST_ROWS) * T;
    SS_CHECK_ARG(blocks < (1ll << 31), "stem_conv: too many tiles");
    SS_LAUNCH_STEM(blocks, as_stream(stream), frames, w_tap_major, bias, out, T, H, W);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_encoder_forward(const StemsegEncoderDesc* desc, const StemsegEncoderWeights* wts, const float* frames,
                                           const StemsegVolume* out, void* workspace, size_t ws_bytes, void* stream) {
    EncoderPlan p;
    int rc = make_encoder_plan(desc, p);
    if (rc) return rc;
    SS_CHECK_ARG(wts && frames && out && workspace, "encoder_forward: null pointer");
    if (ws_bytes < (size_t)p.total * sizeof(float)) {
        set_error("encoder: workspace too small (%zu < %zu bytes)", ws_bytes, (size_t)p.total * sizeof(float));
        return STEMSEG_E_WORKSPACE;
    }
    SS_CHECK_ARG(wts->stem_w && wts->stem_b, "encoder_forward: null stem weights");
    for (int i = 0; i < 4; ++i) {
        for (int c = 0; c < desc->n_clips; ++c) {
            const StemsegVolume& o = out[4 * c + i];
            const int Tc_ = desc->clip_frames > 0 ? desc->clip_frames : p.T / desc->n_clips;
            SS_CHECK_ARG(o.ptr && o.C == desc->out_channels && o.T == Tc_ && o.H == p.h[i] && o.W == p.w[i],
                         "encoder_forward: output volume %d of clip %d must be [%d][%d][%d][%d]", i, c, desc->out_channels, Tc_, p.h[i], p.w[i]);
        }
        SS_CHECK_ARG(wts->fpn_inner_w[i] && wts->fpn_inner_b[i] && wts->fpn_layer_w[i] && wts->fpn_layer_b[i], "encoder_forward: null FPN weights %d", i);
    }
    SS_CHECK_ARG(desc->out_channels == 256, "encoder: out_channels must be 256");
    SS_CHECK_ARG(desc->precision == STEMSEG_PRECISION_F32 || desc->precision == STEMSEG_PRECISION_BF16X6 || desc->precision == STEMSEG_PRECISION_F16X3,
                 "encoder: precision must be 0 (f32), 2 (bf16x6) or 3 (f16x3)");
    const int prec = desc->precision;
    // every convolution of the pass: this precision, and its launch decisions on the planning frame count (conv_igemm.hip, PlanCtx)
    auto epi_for = [&](int frames) {
        ConvEpilogue e;
        e.precision = prec;
        if (p.plan_frames > 0) { e.frames = frames; e.plan_frames = p.plan_frames; e.plan_scratch_floats = p.plan_SKfloats; }
        return e;
    };
    hipStream_t s = as_stream(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    const int T = p.T;

    // stem (resnet.py:292-304)
    {
        const int Ho = p.H / 2, Wo = p.W / 2;
        const int blocks = (int)(ceil_div(Wo, ST_COLS) * ceil_div(Ho, ST_ROWS) * T);
        void* ev = profile_begin(47, 4.0 * ((double)3 * T * p.H * p.W + 64.0 * T * Ho * Wo), s);
        if (wts->stem_w_s2d && prec == STEMSEG_PRECISION_F16X3 && p.W % 2 == 0) {
            // space-to-depth + 4x4 convolution on the split-staged kernel (see stem_s2d_kernel)
            hipLaunchKernelGGL(stem_s2d_kernel, dim3(grid1d((int64_t)T * 3 * p.H * (p.W / 2))), dim3(256), 0, s, frames, ws + p.S2D, T, p.H, p.W, p.s2d_ts, (int)p.s2d_pitch);
            SS_LAUNCH_CHECK();
            const StemsegVolume in = make_volume(ws + p.S2D, (int64_t)T * p.s2d_ts, p.s2d_ts, p.s2d_pitch, 12, T, Ho + 3, Wo + 3, 12 * (int64_t)T * p.s2d_ts + 64);
            ConvEpilogue es = epi_for(T);
            es.relu = 1;
            rc = launch_conv3d(in, wts->stem_w_s2d, wts->stem_b, dense_volume(ws + p.S0, 64, T, Ho, Wo), 1, 4, 4, 0, s, nullptr, 0, &es);
            if (rc) return rc;
        } else {
            SS_LAUNCH_STEM(blocks, s, frames, wts->stem_w, wts->stem_b, ws + p.S0, T, p.H, p.W);
        }
        profile_end(ev, s);
        SS_LAUNCH_CHECK();
        ev = profile_begin(48, 4.0 * 64.0 * T * ((double)Ho * Wo + (double)p.V[0] / T), s);
        const int64_t mp_items = (int64_t)64 * T * (Ho / 2) * (Wo / 8);
        if (Wo % 8 == 0 && mp_items < (1ll << 31))
            hipLaunchKernelGGL(maxpool3x3s2_vec4_kernel, dim3((unsigned)ceil_div(mp_items, 256)), dim3(256), 0, s, (const float*)(ws + p.S0), ws + p.X1, Ho,
                               Wo, (unsigned)mp_items);
        else
            hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid1d(64 * p.V[0])), dim3(256), 0, s, (const float*)(ws + p.S0), ws + p.X1, (int64_t)64 * T, Ho, Wo);
        profile_end(ev, s);
        SS_LAUNCH_CHECK();
    }
    // residual stages (resnet.py:105-113)
    float* x = ws + p.X1;
    int cin = 64, bi = 0;
    bool conv1_done = false;            // this block's conv1 came out of the previous block's fused tail (bottleneck_fused.hip)
    for (int st = 0; st < 4; ++st) {
        const int mid = 64 << st, cout = 256 << st, h = p.h[st], w = p.w[st];
        const int64_t V = p.V[st];
        for (int b = 0; b < p.nblk[st]; ++b, ++bi) {
            const bool first = (b == 0), stride2 = first && st > 0;
            SS_CHECK_ARG(wts->conv1_w[bi] && wts->conv1_b[bi] && wts->conv2_w[bi] && wts->conv2_b[bi] && wts->conv3_w[bi] && wts->conv3_b[bi],
                         "encoder_forward: null weights for block %d", bi);
            SS_CHECK_ARG(!first || (wts->down_w[bi] && wts->down_b[bi]), "encoder_forward: block %d needs a projection shortcut", bi);
            float* xin = x;
            if (stride2) {
                void* ev = profile_begin(49, 4.0 * 2.0 * (double)cin * V, s);       // (the kept quarter is read, sector granularity aside)
                const int64_t ss_items = (int64_t)cin * T * h * (w / 4);
                if (w % 4 == 0 && ss_items < (1ll << 31))
                    hipLaunchKernelGGL(subsample2_vec4_kernel, dim3((unsigned)ceil_div(ss_items, 256)), dim3(256), 0, s, (const float*)x, ws + p.XS, 2 * h,
                                       2 * w, (unsigned)ss_items);
                else
                    hipLaunchKernelGGL(subsample2_kernel, dim3(grid1d((int64_t)cin * V)), dim3(256), 0, s, (const float*)x, ws + p.XS, (int64_t)cin * T, 2 * h, 2 * w);
                profile_end(ev, s);
                SS_LAUNCH_CHECK();
                xin = ws + p.XS;
            }
            float* y = (b == p.nblk[st] - 1) ? ws + p.Cst[st] : ((b & 1) ? ws + p.B : ws + p.A);
            if (!conv1_done) {
                ConvEpilogue e1 = epi_for(T);       // conv1 + bn1 + relu -> zero-haloed 2-D layout (input of the 3x3)
                e1.relu = 1; e1.dec_H = h; e1.dec_W = w;
                rc = launch_conv3d(flat_view(xin, cin, V), wts->conv1_w[bi], wts->conv1_b[bi], interior2d_view(ws + p.M1[st], mid, T, h, w), 1, 1, 1, 0, s,
                                   ws + p.SK, p.SKfloats, &e1);
                if (rc) return rc;
            }
            // Fused tail (f16x3): conv3 of this block and conv1 of the next in one back-to-back kernel.  conv2 then writes its output as the
            // fp16 operand planes that kernel stages by LDS-DMA (same bytes, the M2 buffer) -- unless its plan splits K, in which case it
            // says so and the block runs its three launches (a function of the planning shape: the same choice for every batch).
            const bool want_fuse = ((desc->fuse_tail >> st) & 1) && prec == STEMSEG_PRECISION_F16X3 && b + 1 < p.nblk[st] && fused_tail_supported(mid) && V <= (1ll << 27) && (int64_t)T * (h + 2) * (w + 4) <= (1ll << 27);
            int p16_done = 0;
            ConvEpilogue e2 = epi_for(T);       // conv2 (3x3) + bn2 + relu -> dense
            e2.relu = 1;
            if (want_fuse) { e2.p16_out = reinterpret_cast<unsigned int*>(ws + p.M2); e2.p16_done = &p16_done; }
            rc = launch_conv3d(halo2d_view(ws + p.M1[st], mid, T, h, w), wts->conv2_w[bi], wts->conv2_b[bi], dense_volume(ws + p.M2, mid, T, h, w), 1, 3, 3, 0, s,
                               ws + p.SK, p.SKfloats, &e2);
            if (rc) return rc;
            const float* idt = xin;
            ConvEpilogue ed = epi_for(T);
            if (first) {                        // projection shortcut: 1x1 (stride folded into xin) + bn
                rc = launch_conv3d(flat_view(xin, cin, V), wts->down_w[bi], wts->down_b[bi], flat_view(ws + p.DS, cout, V), 1, 1, 1, 0, s, ws + p.SK, p.SKfloats,
                                   &ed);
                if (rc) return rc;
                idt = ws + p.DS;
            }
            if (want_fuse && p16_done) {
                SS_CHECK_ARG(wts->conv1_w[bi + 1] && wts->conv1_b[bi + 1], "encoder_forward: null weights for block %d", bi + 1);
                rc = launch_fused_tail(mid, reinterpret_cast<const unsigned int*>(ws + p.M2), wts->conv3_w[bi], wts->conv3_b[bi], idt, y, wts->conv1_w[bi + 1],
                                       wts->conv1_b[bi + 1], interior2d_view(ws + p.M1[st], mid, T, h, w), h, w, V, (desc->fuse_tail >> 3) & 3, s);
                if (rc) return rc;
                conv1_done = true;
            } else {
                ConvEpilogue e3 = epi_for(T);       // conv3 + bn3 + identity + relu
                e3.relu = 1; e3.res = idt; e3.res_cs = V; e3.res_ts = 0; e3.res_ys = 0;
                rc = launch_conv3d(flat_view(ws + p.M2, mid, V), wts->conv3_w[bi], wts->conv3_b[bi], flat_view(y, cout, V), 1, 1, 1, 0, s, ws + p.SK, p.SKfloats, &e3);
                if (rc) return rc;
                conv1_done = false;
            }
            x = y;
            cin = cout;
        }
    }
    // FPN (fpn.py:47-69), coarsest level first
    for (int k = 3; k >= 0; --k) {
        const int h = p.h[k], w = p.w[k];
        ConvEpilogue e = epi_for(T), el = epi_for(T);
        e.dec_H = h; e.dec_W = w;
        // Levels with a top-down term: the lateral conv writes a DENSE map (16-byte epilogue: its by-element epilogue into the zero-haloed layout
        // issued four times the store instructions -- 770 us for the 4x level's 1.7 GB) into the idle block buffer A, and the add pass, which
        // touches every element anyway, writes the zero-haloed layout.  Same tile, same sums: the same bits (STEMSEG_FPN_LATERAL_DENSE=0: in place).
        Padded2D gf0(256, T, h, w);
        static const bool lat_dense_on = [] { const char* e = getenv("STEMSEG_FPN_LATERAL_DENSE"); return !(e && e[0] == '0'); }();
        const unsigned gq0 = (unsigned)((w + 1) / 4 + 1);
        const bool lat_dense = lat_dense_on && k < 3 && h % 2 == 0 && w % 4 == 0 && p.h[k + 1] == h / 2 && p.w[k + 1] == w / 2 && gf0.pitch % 4 == 0 &&
                               (int64_t)4 * gq0 <= gf0.pitch && (int64_t)256 * T * (h / 2 + 1) * gq0 < (1ll << 32) - 256 &&
                               (reinterpret_cast<uintptr_t>(ws + p.L[k]) % 16 == 0) && gf0.ts % 4 == 0 && (reinterpret_cast<uintptr_t>(ws + p.A) % 16 == 0);
        if (lat_dense) {
            ConvEpilogue ed = epi_for(T);
            rc = launch_conv3d(flat_view(ws + p.Cst[k], 256 << k, p.V[k]), wts->fpn_inner_w[k], wts->fpn_inner_b[k], flat_view(ws + p.A, 256, p.V[k]), 1, 1, 1, 0, s,
                               ws + p.SK, p.SKfloats, &ed);
            if (rc) return rc;
            Padded2D gc0(256, T, p.h[k + 1], p.w[k + 1]);
            void* ev = profile_begin(50, 4.0 * 256.0 * (2.0 * p.V[k] + p.V[k + 1]), s);
            const int64_t items = (int64_t)256 * T * (h / 2 + 1) * gq0;
            hipLaunchKernelGGL(upsample2x_add_dense_vec4x2_kernel, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, s, ws + p.L[k], (const float*)(ws + p.A),
                               (const float*)(ws + p.L[k + 1] + gc0.interior), h, w, gq0, (unsigned)items, gf0.ts, (int)gf0.pitch, gc0.ts, (int)gc0.pitch);
            profile_end(ev, s);
            SS_LAUNCH_CHECK();
        } else {
        rc = launch_conv3d(flat_view(ws + p.Cst[k], 256 << k, p.V[k]), wts->fpn_inner_w[k], wts->fpn_inner_b[k], interior2d_view(ws + p.L[k], 256, T, h, w), 1, 1, 1, 0, s,
                           ws + p.SK, p.SKfloats, &e);
        if (rc) return rc;
        }
        if (k < 3 && !lat_dense) {
            // (Measured in round 5 and not kept: the add fused into the lateral conv's epilogue -- four gathered coarse loads per output
            // element in the by-element epilogue of a flat launch.  The separate pass goes (-0.22 ms per clip), the 1x1 class pays +0.38 ms
            // and every tile's kernel arguments grow: the step 103.0 vs 104.8 clips/s, interleaved on one box.)
            Padded2D gf(256, T, h, w), gc(256, T, p.h[k + 1], p.w[k + 1]);
            void* ev = profile_begin(50, 4.0 * 256.0 * (2.0 * p.V[k] + p.V[k + 1]), s);         // fine read + written, coarse read
            const unsigned gq = (unsigned)((w + 1) / 4 + 1);                              // aligned groups covering haloed columns 1 .. w
            const int64_t ua_items = (int64_t)256 * T * h * gq;
            const bool ua_vec = gf.pitch % 4 == 0 && (int64_t)4 * gq <= gf.pitch && ua_items < (1ll << 32) - 256 && (reinterpret_cast<uintptr_t>(ws + p.L[k]) % 16 == 0) && gf.ts % 4 == 0;
            static const bool ua_two_rows = [] { const char* e = getenv("STEMSEG_FPN_ADD_ROWS"); return !(e && e[0] == '1'); }();      // (A/B switch; default: two rows per thread)
            const int64_t ua2_items = (int64_t)256 * T * (h / 2 + 1) * gq;
            if (ua_vec && ua_two_rows && h % 2 == 0 && w % 2 == 0 && p.h[k + 1] == h / 2 && p.w[k + 1] == w / 2)
                hipLaunchKernelGGL(upsample2x_add_vec4x2_kernel, dim3((unsigned)ceil_div(ua2_items, 256)), dim3(256), 0, s, ws + p.L[k],
                                   (const float*)(ws + p.L[k + 1] + gc.interior), h, w, gq, (unsigned)ua2_items, gf.ts, (int)gf.pitch, gc.ts, (int)gc.pitch);
            else if (ua_vec)
                hipLaunchKernelGGL(upsample2x_add_vec4_kernel, dim3((unsigned)ceil_div(ua_items, 256)), dim3(256), 0, s, ws + p.L[k],
                                   (const float*)(ws + p.L[k + 1] + gc.interior), h, w, gq, (unsigned)ua_items, gf.ts, (int)gf.pitch, gc.ts, (int)gc.pitch);
            else
                hipLaunchKernelGGL(upsample2x_add_kernel, dim3(grid1d(256 * p.V[k])), dim3(256), 0, s, ws + p.L[k] + gf.interior,
                                   (const float*)(ws + p.L[k + 1] + gc.interior), (int64_t)256 * T, h, w, gf.ts, gf.pitch, gc.ts, gc.pitch);
            profile_end(ev, s);
            SS_LAUNCH_CHECK();
        }
        // the output conv runs once per clip: each clip's map goes to its own (usually zero-haloed) consumer volume
        const int Tc = desc->clip_frames > 0 ? desc->clip_frames : T / desc->n_clips;
        const int Ts = desc->clip_frames > 0 ? desc->clip_stride : Tc;          // first frame of clip c within the pass: c * Ts
        if (p.FO[k] >= 0) {
            // ... unless the clips overlap: then once over all T frames into a dense map, and every clip copies its window
            // (a shared frame's output conv is not repeated; the copy is 0.1 ms per clip against ~2.3 ms of convs)
            rc = launch_conv3d(halo2d_view(ws + p.L[k], 256, T, h, w), wts->fpn_layer_w[k], wts->fpn_layer_b[k], dense_volume(ws + p.FO[k], 256, T, h, w), 1, 3, 3,
                               0, s, ws + p.SK, p.SKfloats, &el);
            if (rc) return rc;
            for (int c = 0; c < desc->n_clips; ++c) {
                rc = launch_copy_strided(ws + p.FO[k] + (int64_t)c * Ts * h * w, (int64_t)T * h * w, (int64_t)h * w, out[4 * c + k], s);
                if (rc) return rc;
            }
            continue;
        }
        // ... in ONE launch when the clips' consumer volumes lie a fixed stride apart (the decoders' clip batch allocates them so): the clip
        // is a grid dimension of the conv (decided, like every launch of the pass, on the planning shape: same bits either way)
        bool uniform = desc->n_clips > 1 && desc->n_clips <= 65535;
        const int64_t obs = uniform ? out[4 + k].ptr - out[k].ptr : 0;
        for (int c = 1; c < desc->n_clips && uniform; ++c) {
            const StemsegVolume &a = out[4 * c + k], &b = out[k];
            uniform = a.ptr == b.ptr + c * obs && a.c_stride == b.c_stride && a.t_stride == b.t_stride && a.y_stride == b.y_stride && a.limit == b.limit;
        }
        if (uniform && obs > 0 && obs % 4 == 0) {
            StemsegVolume in = halo2d_view(ws + p.L[k], 256, T, h, w);
            ConvEpilogue ec = epi_for(Tc);
            ec.nb = desc->n_clips;
            ec.in_bs = (int64_t)Ts * in.t_stride;
            ec.out_bs = obs;
            in.limit -= (int64_t)(desc->n_clips - 1) * ec.in_bs;        // (the last clip's room: every clip reads Tc frames from its own origin)
            in.T = Tc;
            rc = launch_conv3d(in, wts->fpn_layer_w[k], wts->fpn_layer_b[k], out[k], 1, 3, 3, 0, s, ws + p.SK, p.SKfloats, &ec);
            if (rc) return rc;
            continue;
        }
        for (int c = 0; c < desc->n_clips; ++c) {
            StemsegVolume in = halo2d_view(ws + p.L[k], 256, T, h, w);
            in.ptr += (int64_t)c * Ts * in.t_stride;
            in.limit -= (int64_t)c * Ts * in.t_stride;
            in.T = Tc;
            const ConvEpilogue ec = epi_for(Tc);
            rc = launch_conv3d(in, wts->fpn_layer_w[k], wts->fpn_layer_b[k], out[4 * c + k], 1, 3, 3, 0, s, ws + p.SK, p.SKfloats, &ec);
            if (rc) return rc;
        }
    }
    return STEMSEG_OK;
}

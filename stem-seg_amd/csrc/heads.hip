// Fused output heads: three 1x1x1 convs + activations + coordinate grid + bandwidth activation in one pass.
//
// Reference: /root/reference/stemseg/modeling/embedding_decoder.py:131-143
//   emb = tanh(0.25 * conv_embedding(x)) + grid ; var = conv_variance(x) (+bias) ; seed = sigmoid(conv_seediness(x))
// seediness_decoder.py:112 (sigmoid(conv_out(x))) and modeling/inference_model.py:148 (bandwidth = exp(var) * 10).
// The reference reads the 128-channel feature tensor three times (three convs) and runs five more elementwise
// kernels; here it is read exactly once (HBM-bound: 4*Cin bytes in, 4*n_out bytes out per voxel).
#include "common.h"
#include <algorithm>

namespace stemseg {

constexpr int HEADS_MAX_OUT = STEMSEG_MAX_EMB_DIMS * 2;

struct HeadsParams {
    const float* x;
    const float* w;
    const float* bias;
    const float* gt;
    const float* gy;
    const float* gx;
    const float* add;              // optional [n_out][V] term added to the linear part before bias / activation (the linear tail of the decoders)
    const float* gn_stats;         // optional: x is a RAW conv output; GroupNorm (mean, rstd per group) + affine + ReLU are applied as it is read --
    const float* gn_gamma;         // relu(fma(x, rstd gamma, beta - mean rstd gamma)), the expressions of gn_relu_stream_kernel (norm_pool.hip), so the
    const float* gn_beta;          // normalised map the heads are the only reader of is never written
    int gn_cpg;
    int64_t gn_stats_bs;
    float* out;
    int Cin, T, H, W;
    int64_t V;
    int64_t x_bs, out_bs, add_bs;  // clip batch (grid.y = clip)
    int act[HEADS_MAX_OUT];
    int axis[HEADS_MAX_OUT];
};

__device__ __forceinline__ float head_act(float z, int act, float grid) {
    switch (act) {
        case 1: return tanhf(0.25f * z) + grid;
        case 2: return 1.0f / (1.0f + expf(-z));
        case 3: return expf(z) * 10.0f;
        case 4: return z + grid;
        default: return z;
    }
}

// one thread = 4 consecutive voxels (float4 loads along W), NOUT accumulators each
template <int NOUT>
__global__ __launch_bounds__(256) void heads_kernel(HeadsParams p) {
    extern __shared__ __attribute__((aligned(16))) float w_lds[];   // [NOUT][Cin] (+ [2][Cin]: the GroupNorm scale / shift per channel)
    p.x += (int64_t)blockIdx.y * p.x_bs; p.out += (int64_t)blockIdx.y * p.out_bs;
    if (p.add) p.add += (int64_t)blockIdx.y * p.add_bs;
    for (int i = threadIdx.x; i < NOUT * p.Cin; i += blockDim.x) w_lds[i] = p.w[i];
    float* ab = w_lds + NOUT * p.Cin;
    if (p.gn_stats) {
        const float* st = p.gn_stats + (int64_t)blockIdx.y * p.gn_stats_bs;
        for (int c = threadIdx.x; c < p.Cin; c += blockDim.x) {
            const int g = c / p.gn_cpg;
            const float a = st[2 * g + 1] * p.gn_gamma[c];
            const float b = p.gn_beta[c] - st[2 * g] * a;
            ab[c] = a;
            ab[p.Cin + c] = b;
        }
    }
    __syncthreads();
    const int64_t nq = p.V / 4;
    const int64_t HW = (int64_t)p.H * p.W;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        float4 acc[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* xp = reinterpret_cast<const float4*>(p.x) + q;
        for (int c = 0; c < p.Cin; c += 4) {
            float4 xv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) xv[k] = xp[(int64_t)(c + k) * nq];
            if (p.gn_stats) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = ab[c + k], b = ab[p.Cin + c + k];
                    xv[k].x = relu_keep_nan(fmaf(xv[k].x, a, b)); xv[k].y = relu_keep_nan(fmaf(xv[k].y, a, b));
                    xv[k].z = relu_keep_nan(fmaf(xv[k].z, a, b)); xv[k].w = relu_keep_nan(fmaf(xv[k].w, a, b));
                }
            }
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float4 wv = *reinterpret_cast<const float4*>(w_lds + o * p.Cin + c);   // LDS broadcast
                acc[o].x += wv.x * xv[0].x; acc[o].y += wv.x * xv[0].y; acc[o].z += wv.x * xv[0].z; acc[o].w += wv.x * xv[0].w;
                acc[o].x += wv.y * xv[1].x; acc[o].y += wv.y * xv[1].y; acc[o].z += wv.y * xv[1].z; acc[o].w += wv.y * xv[1].w;
                acc[o].x += wv.z * xv[2].x; acc[o].y += wv.z * xv[2].y; acc[o].z += wv.z * xv[2].z; acc[o].w += wv.z * xv[2].w;
                acc[o].x += wv.w * xv[3].x; acc[o].y += wv.w * xv[3].y; acc[o].z += wv.w * xv[3].z; acc[o].w += wv.w * xv[3].w;
            }
        }
        // voxel coordinates of the 4 lanes (W % 4 == 0 so they share t and y)
        const int64_t v0 = q * 4;
        const int t = (int)(v0 / HW);
        const int64_t r = v0 - (int64_t)t * HW;
        const int y = (int)(r / p.W), x0 = (int)(r - (int64_t)y * p.W);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float b = p.bias ? p.bias[o] : 0.f;
            if (p.add) {
                const float4 a4 = reinterpret_cast<const float4*>(p.add + (int64_t)o * p.V)[q];
                acc[o].x += a4.x; acc[o].y += a4.y; acc[o].z += a4.z; acc[o].w += a4.w;
            }
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            const int ax = p.axis[o];
            if (ax == 1) { g[0] = g[1] = g[2] = g[3] = p.gt[t]; }
            else if (ax == 2) { g[0] = g[1] = g[2] = g[3] = p.gy[y]; }
            else if (ax == 3) { g[0] = p.gx[x0]; g[1] = p.gx[x0 + 1]; g[2] = p.gx[x0 + 2]; g[3] = p.gx[x0 + 3]; }
            float4 o4;
            o4.x = head_act(acc[o].x + b, p.act[o], g[0]);
            o4.y = head_act(acc[o].y + b, p.act[o], g[1]);
            o4.z = head_act(acc[o].z + b, p.act[o], g[2]);
            o4.w = head_act(acc[o].w + b, p.act[o], g[3]);
            reinterpret_cast<float4*>(p.out + (int64_t)o * p.V)[q] = o4;
        }
    }
}

template <int NOUT>
static int launch_heads_n(const HeadsParams& p, int nb, hipStream_t s) {
    const int64_t nq = p.V / 4;
    const int blocks = (int)std::min<int64_t>(ceil_div(nq, 256), 256 * 8);
    hipLaunchKernelGGL(heads_kernel<NOUT>, dim3(blocks, (unsigned)nb), dim3(256), (size_t)(NOUT + 2) * p.Cin * sizeof(float), s, p);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

// A level of the decoders' LINEAR TAIL (decoder.hip): z = M x (+ add), M [NOUT][Cin], x dense [Cin][V], no bias, no activation -- the coarse
// levels' share of the heads, computed at their own resolution.  One thread per voxel (any V: the 32x / 16x maps are 15 x 27 / 30 x 54), the
// weights broadcast from LDS; these maps are tens of thousands of voxels, nothing to optimise.
template <int NOUT>
__global__ __launch_bounds__(256) void level_head_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ add,
                                                          float* __restrict__ out, int Cin, int64_t V, int64_t x_bs, int64_t add_bs, int64_t out_bs) {
    extern __shared__ __attribute__((aligned(16))) float w_lds[];   // [NOUT][Cin]
    x += (int64_t)blockIdx.y * x_bs; out += (int64_t)blockIdx.y * out_bs;
    if (add) add += (int64_t)blockIdx.y * add_bs;
    for (int i = threadIdx.x; i < NOUT * Cin; i += blockDim.x) w_lds[i] = w[i];
    __syncthreads();
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
        float acc[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) acc[o] = 0.f;
        for (int c = 0; c < Cin; ++c) {
            const float xv = x[(int64_t)c * V + v];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) acc[o] += w_lds[o * Cin + c] * xv;
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) out[(int64_t)o * V + v] = add ? acc[o] + add[(int64_t)o * V + v] : acc[o];
    }
}

template <int NOUT>
static int launch_level_head_n(const float* x, int Cin, int64_t V, const float* w, const float* add, float* out, hipStream_t s, int nb, int64_t x_bs,
                               int64_t add_bs, int64_t out_bs) {
    const int blocks = (int)std::min<int64_t>(ceil_div(V, 256), 256 * 8);
    hipLaunchKernelGGL(level_head_kernel<NOUT>, dim3(blocks, (unsigned)nb), dim3(256), (size_t)NOUT * Cin * sizeof(float), s, x, w, add, out, Cin, V, x_bs, add_bs, out_bs);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

int launch_level_head(const float* x, int Cin, int64_t V, const float* w, int n_out, const float* add, float* out, hipStream_t s, int nb, int64_t x_bs,
                      int64_t add_bs, int64_t out_bs) {
    SS_CHECK_ARG(x && w && out && nb >= 1 && nb <= 65535 && Cin > 0 && V > 0, "level_head: bad arguments");
    SS_CHECK_ARG(n_out >= 1 && n_out <= STEMSEG_MAX_HEAD_OUT && (size_t)n_out * Cin * sizeof(float) <= 48 * 1024, "level_head: n_out=%d, Cin=%d unsupported", n_out, Cin);
    void* ev = profile_begin(44, 4.0 * (double)V * (Cin + n_out) * nb, s);
    int rc;
    switch (n_out) {
        case 1: rc = launch_level_head_n<1>(x, Cin, V, w, add, out, s, nb, x_bs, add_bs, out_bs); break;
        case 2: rc = launch_level_head_n<2>(x, Cin, V, w, add, out, s, nb, x_bs, add_bs, out_bs); break;
        case 3: rc = launch_level_head_n<3>(x, Cin, V, w, add, out, s, nb, x_bs, add_bs, out_bs); break;
        case 4: rc = launch_level_head_n<4>(x, Cin, V, w, add, out, s, nb, x_bs, add_bs, out_bs); break;
        case 5: rc = launch_level_head_n<5>(x, Cin, V, w, add, out, s, nb, x_bs, add_bs, out_bs); break;
        case 6: rc = launch_level_head_n<6>(x, Cin, V, w, add, out, s, nb, x_bs, add_bs, out_bs); break;
        case 7: rc = launch_level_head_n<7>(x, Cin, V, w, add, out, s, nb, x_bs, add_bs, out_bs); break;
        case 8: rc = launch_level_head_n<8>(x, Cin, V, w, add, out, s, nb, x_bs, add_bs, out_bs); break;
        case 9: rc = launch_level_head_n<9>(x, Cin, V, w, add, out, s, nb, x_bs, add_bs, out_bs); break;
        default: rc = launch_level_head_n<10>(x, Cin, V, w, add, out, s, nb, x_bs, add_bs, out_bs); break;
    }
    profile_end(ev, s);
    return rc;
}

int launch_heads(const float* x, int Cin, int T, int H, int W, const float* w, const float* bias, const HeadSpec& hs,
                 const float* gt, const float* gy, const float* gx, float* out, hipStream_t s, const ClipBatch& cb, const float* add, int64_t add_bs,
                 const HeadsGN* gn) {
    SS_CHECK_ARG(x && w && out && cb.nb >= 1 && cb.nb <= 65535 && cb.in_bs % 4 == 0 && cb.out_bs % 4 == 0, "heads: null pointer");
    SS_CHECK_ARG(hs.n_out >= 1 && hs.n_out <= STEMSEG_MAX_HEAD_OUT, "heads: n_out=%d unsupported (1..%d)", hs.n_out, STEMSEG_MAX_HEAD_OUT);
    SS_CHECK_ARG(Cin % 4 == 0 && W % 4 == 0, "heads: Cin %% 4 == 0 and W %% 4 == 0 required (Cin=%d, W=%d)", Cin, W);
    SS_CHECK_ARG((reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0), "heads: 16-byte alignment");
    HeadsParams p;
    p.x = x; p.w = w; p.bias = bias; p.gt = gt; p.gy = gy; p.gx = gx; p.out = out;
    p.Cin = Cin; p.T = T; p.H = H; p.W = W; p.V = (int64_t)T * H * W;
    p.x_bs = cb.in_bs; p.out_bs = cb.out_bs;
    p.add = add; p.add_bs = add_bs;
    p.gn_stats = gn ? gn->stats : nullptr; p.gn_gamma = gn ? gn->gamma : nullptr; p.gn_beta = gn ? gn->beta : nullptr;
    p.gn_cpg = gn ? gn->cpg : 1; p.gn_stats_bs = gn ? gn->stats_bs : 0;
    SS_CHECK_ARG(!gn || (gn->stats && gn->gamma && gn->beta && gn->cpg > 0 && Cin % gn->cpg == 0), "heads: bad GroupNorm arguments");
    SS_CHECK_ARG(!add || (reinterpret_cast<uintptr_t>(add) % 16 == 0 && add_bs % 4 == 0), "heads: 16-byte aligned addend");
    const int nb = cb.nb;
    for (int o = 0; o < HEADS_MAX_OUT; ++o) { p.act[o] = 0; p.axis[o] = 0; }
    for (int o = 0; o < hs.n_out; ++o) {
        p.act[o] = hs.act[o];
        p.axis[o] = hs.grid_axis[o];
        SS_CHECK_ARG(p.act[o] >= 0 && p.act[o] <= 4 && p.axis[o] >= 0 && p.axis[o] <= 3, "heads: bad act/axis code for channel %d", o);
        const bool needs_grid = (p.act[o] == 1 || p.act[o] == 4) && p.axis[o] != 0;
        if (!needs_grid) p.axis[o] = 0;
        SS_CHECK_ARG(!needs_grid || (gt && gy && gx), "heads: grid vectors required for channel %d", o);
    }
    void* ev = profile_begin(44, 4.0 * (double)p.V * (Cin + hs.n_out) * nb, s);
    int rc;
    switch (hs.n_out) {
        case 1: rc = launch_heads_n<1>(p, nb, s); break;
        case 2: rc = launch_heads_n<2>(p, nb, s); break;
        case 3: rc = launch_heads_n<3>(p, nb, s); break;
        case 4: rc = launch_heads_n<4>(p, nb, s); break;
        case 5: rc = launch_heads_n<5>(p, nb, s); break;
        case 6: rc = launch_heads_n<6>(p, nb, s); break;
        case 7: rc = launch_heads_n<7>(p, nb, s); break;
        case 8: rc = launch_heads_n<8>(p, nb, s); break;
        case 9: rc = launch_heads_n<9>(p, nb, s); break;          // xytff + seediness (embedding_utils.py:4-25): 5 + 3 + 1
        default: rc = launch_heads_n<10>(p, nb, s); break;
    }
    profile_end(ev, s);
    return rc;
}

// flags[b] = 1 when chunk b of x holds a non-finite value, else 0 -- every flag is (re)written by every launch (no reset, no atomics).
// One workgroup of 1024 threads per chunk, four 16-byte loads in flight per thread: n_flags workgroups are all the parallelism there is (64
// per head output), and with one load per thread in flight a 20 MB class-logit volume (41 classes, 384 x 640) took 310 us per call -- 7 % of a
// YouTube-VIS step.
__global__ __launch_bounds__(1024) void nonfinite_flags_kernel(const float* __restrict__ x, long long n, int* __restrict__ flags) {
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    unsigned int bad = 0;
    auto test = [&](const float v) { bad |= (unsigned int)((__float_as_uint(v) & 0x7f800000u) == 0x7f800000u); };      // exponent all ones: inf or NaN
    long long a = lo + ((4 - (((reinterpret_cast<uintptr_t>(x) >> 2) + lo) & 3)) & 3);
    if (a > hi) a = hi;
    const long long n4 = (hi - a) / 4;
    for (long long i = lo + threadIdx.x; i < a; i += 1024) test(x[i]);
    const float4* x4 = reinterpret_cast<const float4*>(x + a);
    long long i = threadIdx.x;
    for (; i + 3 * 1024 < n4; i += 4 * 1024) {
        const float4 v0 = x4[i], v1 = x4[i + 1024], v2 = x4[i + 2048], v3 = x4[i + 3072];
        test(v0.x); test(v0.y); test(v0.z); test(v0.w);
        test(v1.x); test(v1.y); test(v1.z); test(v1.w);
        test(v2.x); test(v2.y); test(v2.z); test(v2.w);
        test(v3.x); test(v3.y); test(v3.z); test(v3.w);
    }
    for (; i < n4; i += 1024) {
        const float4 v = x4[i];
        test(v.x); test(v.y); test(v.z); test(v.w);
    }
    for (long long k = a + 4 * n4 + threadIdx.x; k < hi; k += 1024) test(x[k]);
    const int any = __syncthreads_or((int)bad);
    if (threadIdx.x == 0) flags[blockIdx.x] = any ? 1 : 0;
}

}  // namespace stemseg

extern "C" int stemseg_hip_nonfinite_flags(const float* x, int64_t n, int32_t* flags, int32_t n_flags, void* stream) {
    using namespace stemseg;
    SS_CHECK_ARG(flags && n >= 0 && n_flags >= 1 && n_flags <= 4096 && (x || n == 0), "nonfinite_flags: bad arguments");
    hipLaunchKernelGGL(nonfinite_flags_kernel, dim3((unsigned)n_flags), dim3(1024), 0, as_stream(stream), x, (long long)n, flags);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_heads(const float* x, int32_t Cin, int32_t T, int32_t H, int32_t W, const float* w, const float* bias,
                                 int32_t n_out, const int32_t* act_host, const int32_t* grid_axis_host, const float* grid_t,
                                 const float* grid_y, const float* grid_x, float* out, void* stream) {
    using namespace stemseg;
    SS_CHECK_ARG(act_host && grid_axis_host && n_out >= 1 && n_out <= STEMSEG_MAX_HEAD_OUT, "heads: bad head spec");
    HeadSpec hs;
    hs.n_out = n_out;
    for (int o = 0; o < n_out; ++o) { hs.act[o] = act_host[o]; hs.grid_axis[o] = grid_axis_host[o]; }
    return launch_heads(x, Cin, T, H, W, w, bias, hs, grid_t, grid_y, grid_x, out, as_stream(stream));
}

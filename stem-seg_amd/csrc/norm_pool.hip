// GroupNorm statistics and the fused GroupNorm-apply + ReLU + temporal-stride-2 average pool.
//
// Reference ops: nn.GroupNorm(32, C) (model_builder.py:32-34, eps 1e-5, biased variance), nn.ReLU and
// nn.AvgPool3d(3, stride=(2,1,1), padding=1) with count_include_pad (divide by 27 always) as instantiated
// in /root/reference/stemseg/modeling/embedding_decoder.py:20-60.  All HBM-bound: the statistics pass reads
// the conv output once; the apply pass reads it once more and writes the (T/2) result straight into the next
// consumer's layout (zero-haloed conv input or a channel slice of a concat buffer), so ReLU and the pool
// never touch HBM on their own.
#include "common.h"
#include <algorithm>

namespace stemseg {

constexpr int GN_SPLIT = 64;   // partial sums per group (deterministic two-level reduction, no atomics)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// grid = (GN_SPLIT, groups).  A group is a contiguous block of cpg*S floats.
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int64_t group_elems,
                                                         double* __restrict__ partial) {
    const int g = blockIdx.y, sp = blockIdx.x;
    const int64_t per = (group_elems + GN_SPLIT - 1) / GN_SPLIT;
    const int64_t per4 = (per + 3) & ~int64_t(3);
    const int64_t beg = (int64_t)sp * per4;
    const int64_t end = (beg + per4 < group_elems) ? beg + per4 : group_elems;
    const float* base = x + (int64_t)g * group_elems;
    float s = 0.f, ss = 0.f;
    const bool vec = ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
    if (vec) {
        const int64_t n4 = (end > beg) ? (end - beg) / 4 : 0;
        const float4* b4 = reinterpret_cast<const float4*>(base + beg);
        for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
            const float4 v = b4[i];
            s += (v.x + v.y) + (v.z + v.w);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        for (int64_t i = beg + n4 * 4 + threadIdx.x; i < end; i += blockDim.x) {
            const float v = base[i];
            s += v;
            ss += v * v;
        }
    } else {
        for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
            const float v = base[i];
            s += v;
            ss += v * v;
        }
    }
    __shared__ double red[2][4];
    double ds = wave_sum((double)s), dss = wave_sum((double)ss);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = ds; red[1][w] = dss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((int64_t)g * GN_SPLIT + sp) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[((int64_t)g * GN_SPLIT + sp) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// one wave per group: fixed-order combine of the GN_SPLIT partials -> mean, rstd
__global__ __launch_bounds__(64) void gn_finalize_kernel(const double* __restrict__ partial, double group_elems, float eps,
                                                         float* __restrict__ stats) {
    const int g = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, ss = 0.0;
    if (lane < GN_SPLIT) {
        s = partial[((int64_t)g * GN_SPLIT + lane) * 2 + 0];
        ss = partial[((int64_t)g * GN_SPLIT + lane) * 2 + 1];
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lane == 0) {
        const double mean = s / group_elems;
        double var = ss / group_elems - mean * mean;
        var = var > 0.0 ? var : 0.0;
        stats[2 * g + 0] = (float)mean;
        stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// one workgroup per group: fixed-order combine of the `used` per-tile partials the conv epilogue / split-K reduce left in
// part[g][slot] (thread k takes slots k, k + 256, ...; then a fixed tree) -> mean, rstd.  Deterministic, no atomics.
__global__ __launch_bounds__(256) void gn_finalize_slots_kernel(const double* __restrict__ part, int cap, int used, double group_elems,
                                                                float eps, float* __restrict__ stats, int64_t part_bs, int64_t stats_bs) {
    const int g = blockIdx.x;
    part += (int64_t)blockIdx.y * part_bs;                     // (grid.y: clip of a clip batch)
    stats += (int64_t)blockIdx.y * stats_bs;
    const double* pg = part + (size_t)g * cap * 2;
    double s = 0.0, ss = 0.0;
    for (int k = threadIdx.x; k < used; k += 256) { s += pg[2 * k]; ss += pg[2 * k + 1]; }
    s = wave_sum(s);
    ss = wave_sum(ss);
    __shared__ double red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        ss = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const double mean = s / group_elems;
        double var = ss / group_elems - mean * mean;
        var = var > 0.0 ? var : 0.0;
        stats[2 * g + 0] = (float)mean;
        stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

struct GnApplyParams {
    const float* x;
    const float* stats;
    const float* gamma;
    const float* beta;
    float* out;
    int64_t out_cs, out_ts, out_ys;
    int64_t x_bs, out_bs, stats_bs;   // clip batch (the launch's last grid dimension = clip)
    int C, T, H, W, To, cpg;
    int pool_max;                  // pooled forms: 0 = AvgPool3d (sum / 27, count_include_pad), 1 = MaxPool3d (values are >= 0 after the ReLU, so the
                                   // -inf padding of nn.MaxPool3d and a zero start give the same maximum)
};

// Scalar form, any destination layout: one thread per output element, x fastest.
template <bool POOL>
__global__ __launch_bounds__(256) void gn_relu_pool_kernel(GnApplyParams p) {
    p.x += (int64_t)blockIdx.y * p.x_bs; p.out += (int64_t)blockIdx.y * p.out_bs; p.stats += (int64_t)blockIdx.y * p.stats_bs;
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t per_c = (int64_t)p.To * HW;
    const int64_t total = per_c * p.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / per_c);
        int64_t r = i - (int64_t)c * per_c;
        const int to = (int)(r / HW);
        r -= (int64_t)to * HW;
        const int y = (int)(r / p.W), x = (int)(r - (int64_t)y * p.W);
        const int g = c / p.cpg;
        const float a = p.stats[2 * g + 1] * p.gamma[c];
        const float b = p.beta[c] - p.stats[2 * g] * a;
        const float* xc = p.x + (int64_t)c * p.T * HW;
        float v;
        if (POOL) {
            float acc = 0.f;
#pragma unroll
            for (int dt = -1; dt <= 1; ++dt) {
                const int t = 2 * to + dt;
                if (t < 0 || t >= p.T) continue;
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    const int yy = y + dy;
                    if (yy < 0 || yy >= p.H) continue;
                    const float* row = xc + (int64_t)t * HW + (int64_t)yy * p.W;
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int xx = x + dx;
                        if (xx < 0 || xx >= p.W) continue;
                        const float rv = relu_keep_nan(fmaf(row[xx], a, b));
                        acc = p.pool_max ? max_keep_nan(acc, rv) : acc + rv;
                    }
                }
            }
            v = p.pool_max ? acc : acc / 27.0f;
        } else {
            v = relu_keep_nan(fmaf(xc[(int64_t)to * HW + (int64_t)y * p.W + x], a, b));
        }
        p.out[(int64_t)c * p.out_cs + (int64_t)to * p.out_ts + (int64_t)y * p.out_ys + x] = v;
    }
}

// Streaming form of the un-pooled apply for dense destinations (block_4x -> its slice of the concat buffer: 106 MB in,
// 106 MB out per decoder at 480p): grid.y = channel, 16-B loads and stores along the contiguous [T][H][W] run of the
// channel, 32-bit indices, the (scale, shift) pair is uniform per workgroup.
__global__ __launch_bounds__(256) void gn_relu_stream_kernel(GnApplyParams p, unsigned n4) {
    p.x += (int64_t)blockIdx.z * p.x_bs; p.out += (int64_t)blockIdx.z * p.out_bs; p.stats += (int64_t)blockIdx.z * p.stats_bs;
    const int c = blockIdx.y, g = c / p.cpg;
    const float a = p.stats[2 * g + 1] * p.gamma[c];
    const float b = p.beta[c] - p.stats[2 * g] * a;
    const float4* src = reinterpret_cast<const float4*>(p.x + (int64_t)c * p.T * p.H * p.W);
    float4* dst = reinterpret_cast<float4*>(p.out + (int64_t)c * p.out_cs);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
        float4 v = src[i];
        v.x = relu_keep_nan(fmaf(v.x, a, b)); v.y = relu_keep_nan(fmaf(v.y, a, b));
        v.z = relu_keep_nan(fmaf(v.z, a, b)); v.w = relu_keep_nan(fmaf(v.w, a, b));
        dst[i] = v;
    }
}

// Pooled apply through LDS: a workgroup owns a band of `band` output rows of one (channel, pooled t) plane.  The 3 input planes x
// (band + 2) rows x W floats it needs are read from HBM ONCE (coalesced along x), normalised / rectified once and staged; the window is
// then separable -- the three planes are folded into one in place, and every output is nine staged values (3 rows x 3 columns).  Round 5's
// form (4 outputs per thread straight from global memory) issued 54 scalar loads per 4 outputs and ran at 0.11 of the HBM roof; this one
// loads each input 3 * (band + 2) / (2 * band) times.  Values are >= 0 after the ReLU, so zero padding serves the average (count_include_pad:
// always / 27) and the maximum alike.  grid = (bands, C * To, clips); the destination may be dense or zero-haloed.
__global__ __launch_bounds__(256) void gn_relu_pool_lds_kernel(GnApplyParams p, int band) {
    extern __shared__ float pl[];                              // [3][band + 2][W]
    p.x += (int64_t)blockIdx.z * p.x_bs; p.out += (int64_t)blockIdx.z * p.out_bs; p.stats += (int64_t)blockIdx.z * p.stats_bs;
    const int c = blockIdx.y / p.To, to = blockIdx.y - c * p.To;
    const int y0 = blockIdx.x * band, rows_out = min(band, p.H - y0), rows_in = rows_out + 2;
    const int g = c / p.cpg;
    const float a = p.stats[2 * g + 1] * p.gamma[c];
    const float b = p.beta[c] - p.stats[2 * g] * a;
    const int HW = p.H * p.W, W = p.W, plane = rows_in * W;
    const float* xc = p.x + (int64_t)c * p.T * HW;
    for (int i = threadIdx.x; i < 3 * plane; i += 256) {
        const int dt = i / plane, r2 = i - dt * plane, r = r2 / W, x = r2 - r * W;
        const int t = 2 * to + dt - 1, yy = y0 + r - 1;
        float v = 0.f;
        if (t >= 0 && t < p.T && yy >= 0 && yy < p.H) v = relu_keep_nan(fmaf(xc[t * HW + yy * W + x], a, b));
        pl[i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < plane; i += 256)             // fold the three planes (each thread its own elements: no hazard)
        pl[i] = p.pool_max ? max_keep_nan(max_keep_nan(pl[i], pl[plane + i]), pl[2 * plane + i]) : (pl[i] + pl[plane + i]) + pl[2 * plane + i];
    __syncthreads();
    for (int i = threadIdx.x; i < rows_out * W; i += 256) {
        const int r = i / W, x = i - r * W;
        float acc = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const float* row = pl + (r + dy) * W + x;
            const float l = x > 0 ? row[-1] : 0.f, m = row[0], rr = x + 1 < W ? row[1] : 0.f;
            acc = p.pool_max ? max_keep_nan(acc, max_keep_nan(max_keep_nan(l, m), rr)) : acc + ((l + m) + rr);
        }
        p.out[(int64_t)c * p.out_cs + (int64_t)to * p.out_ts + (int64_t)(y0 + r) * p.out_ys + x] = p.pool_max ? acc : acc / 27.0f;
    }
}

__global__ void gn_identity_stats_kernel(float* stats, int groups) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < groups) { stats[2 * g] = 0.f; stats[2 * g + 1] = 1.f; }
}
// NORMALIZATION_LAYER 'none' (nn.Identity): mean 0, rstd 1 -- with gamma 1, beta 0 the apply kernels then compute fma(x, 1, 0) = x
int launch_gn_identity_stats(float* stats, int groups, hipStream_t s) {
    SS_CHECK_ARG(stats && groups > 0, "gn_identity_stats: bad arguments");
    hipLaunchKernelGGL(gn_identity_stats_kernel, dim3((unsigned)ceil_div(groups, 64)), dim3(64), 0, s, stats, groups);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

int launch_gn_stats(const float* x, int C, int64_t S, int groups, float eps, float* stats, double* scratch, hipStream_t s) {
    SS_CHECK_ARG(x && stats && scratch, "groupnorm_stats: null pointer");
    SS_CHECK_ARG(groups > 0 && C % groups == 0 && S > 0, "groupnorm_stats: C=%d not divisible by groups=%d", C, groups);
    const int64_t ge = (int64_t)(C / groups) * S;
    void* ev = profile_begin(41, 4.0 * (double)C * (double)S, s);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(GN_SPLIT, groups), dim3(256), 0, s, x, ge, scratch);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups), dim3(64), 0, s, (const double*)scratch, (double)ge, eps, stats);
    profile_end(ev, s);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

int launch_gn_finalize_slots(const double* part, int groups, int cap, int used, double group_elems, float eps, float* stats, hipStream_t s, int nb,
                             int64_t part_bs, int64_t stats_bs) {
    SS_CHECK_ARG(part && stats && groups > 0 && used > 0 && used <= cap && nb >= 1 && nb <= 65535, "gn_finalize_slots: bad arguments (used %d of %d)", used, cap);
    void* ev = profile_begin(41, 16.0 * groups * used * nb, s);
    hipLaunchKernelGGL(gn_finalize_slots_kernel, dim3(groups, nb), dim3(256), 0, s, part, cap, used, group_elems, eps, stats, part_bs, stats_bs);
    profile_end(ev, s);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

int launch_gn_relu_pool(const float* x, int C, int T, int H, int W, int groups, const float* stats, const float* gamma,
                        const float* beta, int pool, const StemsegVolume& out, hipStream_t s, const ClipBatch& cb) {
    SS_CHECK_ARG(x && stats && gamma && beta && out.ptr && cb.nb >= 1 && cb.nb <= 65535, "gn_relu_pool: null pointer");
    const int To = pool ? (T + 1) / 2 : T;   // floor((T + 2 - 3)/2) + 1
    SS_CHECK_ARG(out.C == C && out.T == To && out.H == H && out.W == W,
                 "gn_relu_pool: output volume (%d,%d,%d,%d) != expected (%d,%d,%d,%d)", out.C, out.T, out.H, out.W, C, To, H, W);
    GnApplyParams p;
    p.x = x; p.stats = stats; p.gamma = gamma; p.beta = beta;
    p.out = out.ptr; p.out_cs = out.c_stride; p.out_ts = out.t_stride; p.out_ys = out.y_stride;
    p.C = C; p.T = T; p.H = H; p.W = W; p.To = To; p.cpg = C / groups;
    p.x_bs = cb.in_bs; p.out_bs = cb.out_bs; p.stats_bs = cb.stats_bs;
    const unsigned nb = (unsigned)cb.nb;
    p.pool_max = pool == 2 ? 1 : 0;
    SS_CHECK_ARG(pool >= 0 && pool <= 2, "gn_relu_pool: pool code %d (0 none, 1 average, 2 max)", pool);
    const int64_t total = (int64_t)C * To * H * W;
    const int64_t S = (int64_t)T * H * W;
    const bool small = S < (1ll << 31) && total < (1ll << 40);      // 32-bit offsets inside a channel
    void* ev = profile_begin(pool ? 43 : 42, 4.0 * ((double)C * S + (double)total) * nb, s);
    // band of output rows per workgroup: 3 x (band + 2) x W floats of LDS, at most 48 KB (a function of the plane's shape only)
    const int band = std::min(16, std::max(1, (int)(12288 / (3 * (int64_t)W)) - 2));
    if (pool && small && C * To <= 65535 && 3 * (int64_t)(band + 2) * W <= 12288) {
        const size_t lds = (size_t)3 * (band + 2) * W * sizeof(float);
        hipLaunchKernelGGL(gn_relu_pool_lds_kernel, dim3((unsigned)ceil_div(H, band), (unsigned)(C * To), nb), dim3(256), lds, s, p, band);
    } else if (!pool && C <= 65535 && S % 4 == 0 && S / 4 < (1ll << 31) && out.t_stride == (int64_t)H * W && out.y_stride == W &&
               out.c_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out.ptr) % 16 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0)) {
        const unsigned n4 = (unsigned)(S / 4);
        const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n4, 256 * 4), 4096));   // ~4 float4 per thread
        hipLaunchKernelGGL(gn_relu_stream_kernel, dim3(bx, (unsigned)C, nb), dim3(256), 0, s, p, n4);
    } else {
        const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 256 * 16);
        if (pool) hipLaunchKernelGGL(gn_relu_pool_kernel<true>, dim3(blocks, nb), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(gn_relu_pool_kernel<false>, dim3(blocks, nb), dim3(256), 0, s, p);
    }
    profile_end(ev, s);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

}  // namespace stemseg

extern "C" int stemseg_hip_groupnorm_stats(const float* x, int32_t C, int64_t S, int32_t groups, float eps, float* stats,
                                           double* scratch, void* stream) {
    return stemseg::launch_gn_stats(x, C, S, groups, eps, stats, scratch, stemseg::as_stream(stream));
}

extern "C" int stemseg_hip_gn_relu_pool(const float* x, int32_t C, int32_t T, int32_t H, int32_t W, int32_t groups,
                                        const float* stats, const float* gamma, const float* beta, int32_t pool,
                                        const StemsegVolume* out, void* stream) {
    using namespace stemseg;
    SS_CHECK_ARG(out, "gn_relu_pool: null volume");
    return launch_gn_relu_pool(x, C, T, H, W, groups, stats, gamma, beta, pool, *out, as_stream(stream));
}

// Semantic-segmentation post-processing of the reference's inference loop, HBM-bound elementwise kernels.
//   accumulate: inference_model.py:121-128   per-frame running sum of the clips' class logits
//   masks:      inference_model.py:197-231   mean -> foreground probability + class logits | softmax | argmax
#include "common.h"

#include <algorithm>

using namespace stemseg;

namespace {

struct FrameMap { int f[STEMSEG_MAX_CLIP_FRAMES]; };

// acc[frame(t)][c][p] += clip[c][t][p].  One thread per 4 pixels (16-B accesses when HW % 4 == 0).
template <bool VEC>
__global__ __launch_bounds__(256) void semseg_accumulate_kernel(float* __restrict__ acc, const float* __restrict__ clip, int C, int T, long long HW, FrameMap fm) {
    const int c = blockIdx.y % C, t = blockIdx.y / C;
    if (fm.f[t] < 0) return;                           // slot skipped by the caller
    float* dst = acc + ((long long)fm.f[t] * C + c) * HW;
    const float* src = clip + ((long long)c * T + t) * HW;
    if (VEC) {
        const long long n4 = HW >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
            float4 a = reinterpret_cast<float4*>(dst)[i];
            const float4 b = reinterpret_cast<const float4*>(src)[i];
            a.x = __fadd_rn(a.x, b.x); a.y = __fadd_rn(a.y, b.y); a.z = __fadd_rn(a.z, b.z); a.w = __fadd_rn(a.w, b.w);
            reinterpret_cast<float4*>(dst)[i] = a;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x)
            dst[i] = __fadd_rn(dst[i], src[i]);
    }
}

// One thread per (frame, pixel); the C logits of a pixel are strided by HW (coalesced across the wavefront).
// Two passes over the channels (max, then exp-sum / write) keep the register footprint independent of C.
__global__ __launch_bounds__(256) void semseg_masks_kernel(const float* __restrict__ acc, const float* __restrict__ counts, int F, int C, long long HW,
                                                           int type, float* __restrict__ fg, void* __restrict__ mc) {
    const long long n = (long long)F * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / HW, p = i - f * HW;
        const float cnt = counts[f];
        const float* a = acc + f * C * HW + p;
        if (C == 2) {                                  // softmax over the two channels, channel 1 (inference_model.py:225)
            const float x0 = __fdiv_rn(a[0], cnt), x1 = __fdiv_rn(a[HW], cnt);
            const float m = fmaxf(x0, x1);
            const float e0 = expf(x0 - m), e1 = expf(x1 - m);
            fg[i] = __fdiv_rn(e1, __fadd_rn(e0, e1));
            continue;
        }
        const int K = C - 1;                           // class logits | foreground logit (inference_model.py:212)
        const float xf = __fdiv_rn(a[(long long)K * HW], cnt);
        fg[i] = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-xf)));
        if (type == STEMSEG_SEMSEG_NONE) continue;
        float m = -INFINITY;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            const float x = __fdiv_rn(a[(long long)k * HW], cnt);
            if (x > m || (x != x && m == m)) { m = x; arg = k; }       // first maximum wins; NaN beats numbers (torch argmax)
        }
        if (type == STEMSEG_SEMSEG_ARGMAX) {
            reinterpret_cast<long long*>(mc)[i] = arg;
        } else if (type == STEMSEG_SEMSEG_LOGITS) {
            float* o = reinterpret_cast<float*>(mc) + f * K * HW + p;
            for (int k = 0; k < K; ++k) o[(long long)k * HW] = __fdiv_rn(a[(long long)k * HW], cnt);
        } else {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s = __fadd_rn(s, expf(__fdiv_rn(a[(long long)k * HW], cnt) - m));
            float* o = reinterpret_cast<float*>(mc) + f * K * HW + p;
            for (int k = 0; k < K; ++k) o[(long long)k * HW] = __fdiv_rn(expf(__fdiv_rn(a[(long long)k * HW], cnt) - m), s);
        }
    }
}

// One clip on its own (no cross-clip averaging: count 1, mean = 0. + x = x): foreground probability and the > thr mask straight
// from the decoder's [C][T][HW] logits -- reads the 1 (C > 2) or 2 (C == 2) channels that matter, same arithmetic as
// accumulate + masks.
__global__ __launch_bounds__(256) void semseg_fg_clip_kernel(const float* __restrict__ logits, int C, long long THW, float thr,
                                                             float* __restrict__ prob, unsigned char* __restrict__ mask) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < THW; i += (long long)gridDim.x * blockDim.x) {
        float pr;
        if (C == 2) {
            const float x0 = __fdiv_rn(__fadd_rn(0.f, logits[i]), 1.f), x1 = __fdiv_rn(__fadd_rn(0.f, logits[THW + i]), 1.f);
            const float m = fmaxf(x0, x1);
            const float e0 = expf(x0 - m), e1 = expf(x1 - m);
            pr = __fdiv_rn(e1, __fadd_rn(e0, e1));
        } else {
            const float xf = __fdiv_rn(__fadd_rn(0.f, logits[(long long)(C - 1) * THW + i]), 1.f);
            pr = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-xf)));
        }
        if (prob) prob[i] = pr;
        if (mask) mask[i] = (__fdiv_rn(pr, 1.f) > thr) ? 1 : 0;
    }
}

}  // namespace

extern "C" int stemseg_hip_semseg_fg_clip(const float* clip_logits, int32_t C, int32_t T, int64_t HW, float thr, float* fg_prob,
                                          uint8_t* fg_mask, void* stream) {
    SS_CHECK_ARG(clip_logits && (fg_prob || fg_mask), "semseg_fg_clip: null pointer");
    SS_CHECK_ARG(C >= 2 && T >= 1 && HW >= 1, "semseg_fg_clip: bad dims C=%d T=%d", C, T);
    const long long n = (long long)T * HW;
    hipLaunchKernelGGL(semseg_fg_clip_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(ceil_div(n, 256), 4096))), dim3(256), 0,
                       as_stream(stream), clip_logits, C, n, thr, fg_prob, fg_mask);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_semseg_accumulate(float* acc, const float* clip_logits, int32_t C, int32_t T, int64_t HW,
                                             const int32_t* frame_index, int32_t n_frames, void* stream) {
    SS_CHECK_ARG(acc && clip_logits && frame_index, "semseg_accumulate: null pointer");
    SS_CHECK_ARG(C >= 1 && T >= 1 && T <= STEMSEG_MAX_CLIP_FRAMES && HW >= 1 && (int64_t)C * T <= 65535, "semseg_accumulate: bad dims C=%d T=%d", C, T);
    FrameMap fm;
    for (int t = 0; t < T; ++t) {
        SS_CHECK_ARG(frame_index[t] >= -1 && frame_index[t] < n_frames, "semseg_accumulate: frame index %d outside [-1, %d)", frame_index[t], n_frames);
        for (int u = 0; u < t && frame_index[t] >= 0; ++u)   // two slots adding into the same frame in ONE launch would race
            SS_CHECK_ARG(frame_index[u] != frame_index[t], "semseg_accumulate: frame %d appears twice (repeat frames go in a second call)", frame_index[t]);
        fm.f[t] = frame_index[t];
    }
    const bool vec = (HW % 4 == 0) && (reinterpret_cast<uintptr_t>(acc) % 16 == 0) && (reinterpret_cast<uintptr_t>(clip_logits) % 16 == 0);
    const long long per = vec ? HW / 4 : HW;
    dim3 grid((unsigned)std::max<long long>(1, std::min<long long>(ceil_div(per, 256), 1024)), (unsigned)(C * T));
    if (vec) hipLaunchKernelGGL(semseg_accumulate_kernel<true>, grid, dim3(256), 0, as_stream(stream), acc, clip_logits, C, T, (long long)HW, fm);
    else hipLaunchKernelGGL(semseg_accumulate_kernel<false>, grid, dim3(256), 0, as_stream(stream), acc, clip_logits, C, T, (long long)HW, fm);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_semseg_masks(const float* acc, const float* counts, int32_t F, int32_t C, int64_t HW, int32_t output_type,
                                        float* fg, void* multiclass, void* stream) {
    SS_CHECK_ARG(acc && counts && fg, "semseg_masks: null pointer");
    SS_CHECK_ARG(F >= 0 && C >= 1 && HW >= 1, "semseg_masks: bad dims F=%d C=%d", F, C);
    // C == 1: the lone channel is the foreground logit of a multi-class head (what the clip-parallel path exchanges): sigmoid of its mean
    SS_CHECK_ARG(C >= 2 || output_type == STEMSEG_SEMSEG_NONE, "semseg_masks: a single channel carries no class logits (output_type must be none)");
    SS_CHECK_ARG(output_type >= STEMSEG_SEMSEG_NONE && output_type <= STEMSEG_SEMSEG_ARGMAX, "semseg_masks: output_type %d", output_type);
    SS_CHECK_ARG(C == 2 || output_type == STEMSEG_SEMSEG_NONE || multiclass, "semseg_masks: multiclass output buffer missing");
    if (F == 0) return STEMSEG_OK;
    const long long n = (long long)F * HW;
    hipLaunchKernelGGL(semseg_masks_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(ceil_div(n, 256), 8192))), dim3(256), 0,
                       as_stream(stream), acc, counts, F, C, (long long)HW, output_type, fg, multiclass);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

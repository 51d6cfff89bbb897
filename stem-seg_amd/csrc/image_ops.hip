// Image-space kernels either side of the path, both built on torch's bilinear rule.
//
// (1) Pre-processing (inference_image_loader.py:23-43 + data/common.py:12-30 + structures/image_list.py:93-104):
//     uint8 HWC frame -> bilinear resize -> (/255) -> (x - mean) / std -> (channel flip) -> zero pad to a multiple of 32.
// (2) Mask materialisation right after the stitched labels: the resampling chain shared by the
// reference's three writers (output_utils/davis.py:76-110, youtube_vis.py:118-155, kitti_mots.py:89-130), fused.
//   scatter:  per-point track labels -> dense uint8 map of "kept instance index + 1" at mask resolution
//   resample: one-hot -> bilinear x mask_scale -> crop the network's zero padding -> bilinear resize to the image size ->
//             > 0.5 -> condensed uint8 map, evaluated per output pixel from the <= 16 source pixels it depends on
// The bilinear weights of an output pixel sum to 1 and every source pixel carries ONE label, so at most one instance can
// exceed 0.5: the condensed map holds exactly the information of the reference's K binary planes.
#include "common.h"

#include <algorithm>

using namespace stemseg;

namespace {

__global__ void mask_zero_kernel(unsigned char* dense, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dense[i] = 0;
}

__global__ void mask_scatter_kernel(const long long* __restrict__ ys, const long long* __restrict__ xs, const long long* __restrict__ labels, long long n,
                                    const int* __restrict__ lut, int lut_len, unsigned char* __restrict__ dense, int H, int W) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long y = ys[i], x = xs[i], l = labels[i] + 1;
        if (y < 0 || y >= H || x < 0 || x >= W) continue;
        const int v = (l >= 0 && l < lut_len) ? lut[l] : 0;
        dense[y * W + x] = (unsigned char)(v > 0 && v < 256 ? v : 0);
    }
}

// torch's align_corners=False source coordinate (UpSample.h: area_pixel_compute_source_index + guard_index_and_lambda)
struct Tap { int i0, i1; float w0, w1; };
__device__ __forceinline__ Tap make_tap(float scale, int dst, int in_size) {
    float src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f);
    src = src < 0.f ? 0.f : src;
    int i0 = (int)src;
    i0 = i0 < in_size - 1 ? i0 : in_size - 1;
    float l1 = __fsub_rn(src, (float)i0);
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    Tap t;
    t.i0 = i0; t.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    t.w1 = l1; t.w0 = __fsub_rn(1.f, l1);
    return t;
}

// value = (v00*wx0 + v01*wx1)*wy0 + (v10*wx0 + v11*wx1)*wy1   (UpSampleKernel.cpp Interpolate<2>, unfused)
__device__ __forceinline__ float lerp2(float v00, float v01, float v10, float v11, const Tap& ty, const Tap& tx) {
    const float r0 = __fadd_rn(__fmul_rn(v00, tx.w0), __fmul_rn(v01, tx.w1));
    const float r1 = __fadd_rn(__fmul_rn(v10, tx.w0), __fmul_rn(v11, tx.w1));
    return __fadd_rn(__fmul_rn(r0, ty.w0), __fmul_rn(r1, ty.w1));
}

__global__ __launch_bounds__(256) void mask_resample_kernel(const unsigned char* __restrict__ dense, int h, int w, float up_scale, int up_h, int up_w,
                                                            int crop_h, int crop_w, float sy, float sx, int out_h, int out_w, unsigned char* __restrict__ out) {
    const long long n = (long long)out_h * out_w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int oy = (int)(i / out_w), ox = (int)(i - (long long)oy * out_w);
        // stage 5: cropped up-sampled planes (crop_h x crop_w) -> image size
        const Tap ty = make_tap(sy, oy, crop_h), tx = make_tap(sx, ox, crop_w);
        const int Y[2] = {ty.i0, ty.i1}, X[2] = {tx.i0, tx.i1};
        // stage 3: mask resolution (h x w) -> up-sampled (up_h x up_w); the four points the stage-5 sample reads
        Tap uy[2], ux[2];
        uy[0] = make_tap(up_scale, Y[0], h); uy[1] = make_tap(up_scale, Y[1], h);
        ux[0] = make_tap(up_scale, X[0], w); ux[1] = make_tap(up_scale, X[1], w);
        unsigned char L[2][2][2][2];                     // [Y][X][y-tap][x-tap]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int y0 = uy[a].i0, y1 = uy[a].i1, x0 = ux[b].i0, x1 = ux[b].i1;
                L[a][b][0][0] = dense[y0 * w + x0]; L[a][b][0][1] = dense[y0 * w + x1];
                L[a][b][1][0] = dense[y1 * w + x0]; L[a][b][1][1] = dense[y1 * w + x1];
            }
        const unsigned char* flat = &L[0][0][0][0];
        unsigned char result = 0;
        for (int k = 0; k < 16; ++k) {
            const unsigned char cand = flat[k];
            if (cand == 0) continue;
            bool seen = false;
            for (int j = 0; j < k; ++j) seen = seen || (flat[j] == cand);
            if (seen) continue;
            float s[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    s[a][b] = lerp2(L[a][b][0][0] == cand ? 1.f : 0.f, L[a][b][0][1] == cand ? 1.f : 0.f,
                                    L[a][b][1][0] == cand ? 1.f : 0.f, L[a][b][1][1] == cand ? 1.f : 0.f, uy[a], ux[b]);
            const float v = lerp2(s[0][0], s[0][1], s[1][0], s[1][1], ty, tx);
            if (v > 0.5f && cand > result) result = cand;          // davis.py:104-107: later instances overwrite earlier ones
        }
        out[i] = result;
    }
}

// one thread per output pixel of the PADDED frame, all three channels (the 4 source pixels are 12 contiguous bytes pairs)
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* __restrict__ frames, int T, int H0, int W0, int nh, int nw, int PH, int PW,
                                                         float sy, float sx, float m0, float m1, float m2, float s0, float s1, float s2,
                                                         int unit_scale, int flip, float* __restrict__ out) {
    const long long plane = (long long)PH * PW, n = (long long)T * plane;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i / plane);
        const long long r = i - (long long)t * plane;
        const int y = (int)(r / PW), x = (int)(r - (long long)y * PW);
        float v[3] = {0.f, 0.f, 0.f};
        if (y < nh && x < nw) {
            const Tap ty = make_tap(sy, y, H0), tx = make_tap(sx, x, W0);
            const unsigned char* f = frames + (long long)t * H0 * W0 * 3;
            const unsigned char* p00 = f + ((long long)ty.i0 * W0 + tx.i0) * 3;
            const unsigned char* p01 = f + ((long long)ty.i0 * W0 + tx.i1) * 3;
            const unsigned char* p10 = f + ((long long)ty.i1 * W0 + tx.i0) * 3;
            const unsigned char* p11 = f + ((long long)ty.i1 * W0 + tx.i1) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float a = lerp2((float)p00[c], (float)p01[c], (float)p10[c], (float)p11[c], ty, tx);
                if (unit_scale) a = __fdiv_rn(a, 255.f);
                v[c] = __fdiv_rn(__fsub_rn(a, mean[c]), stdv[c]);
            }
        }
        float* o = out + (long long)t * 3 * plane + r;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(long long)(flip ? 2 - c : c) * plane] = v[c];
    }
}

int grid_for(long long n, int cap) { return (int)std::max<long long>(1, std::min<long long>(ceil_div(n, 256), cap)); }

}  // namespace

extern "C" int stemseg_hip_scatter_instance_index(const int64_t* ys, const int64_t* xs, const int64_t* labels, int64_t n, const int32_t* lut,
                                                  int32_t lut_len, uint8_t* dense, int32_t H, int32_t W, void* stream) {
    SS_CHECK_ARG(dense && H > 0 && W > 0 && n >= 0 && lut_len >= 0, "scatter_instance_index: bad arguments");
    SS_CHECK_ARG(n == 0 || (ys && xs && labels && lut), "scatter_instance_index: null pointer");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(mask_zero_kernel, dim3(grid_for((long long)H * W, 2048)), dim3(256), 0, s, dense, (long long)H * W);
    if (n > 0)
        hipLaunchKernelGGL(mask_scatter_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, s, reinterpret_cast<const long long*>(ys),
                           reinterpret_cast<const long long*>(xs), reinterpret_cast<const long long*>(labels), (long long)n, lut, lut_len, dense, H, W);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_resample_instance_masks(const uint8_t* dense, int32_t h, int32_t w, float mask_scale, int32_t crop_h, int32_t crop_w,
                                                   int32_t out_h, int32_t out_w, uint8_t* out, void* stream) {
    SS_CHECK_ARG(dense && out && h > 0 && w > 0 && out_h > 0 && out_w > 0 && mask_scale > 0.f, "resample_instance_masks: bad arguments");
    // F.interpolate(scale_factor=s): output size floor(in * s), coordinate scale 1 / s (UpSample.h compute_scales_value)
    const int up_h = (int)floorf((float)h * mask_scale), up_w = (int)floorf((float)w * mask_scale);
    SS_CHECK_ARG(crop_h >= 1 && crop_w >= 1 && crop_h <= up_h && crop_w <= up_w,
                 "resample_instance_masks: network input dims without padding (%d, %d) should be <= padded dims (%d, %d)", crop_w, crop_h, up_w, up_h);
    const float up_scale = (float)(1.0 / (double)mask_scale);
    const float sy = (float)crop_h / (float)out_h, sx = (float)crop_w / (float)out_w;   // size-driven resize: in / out
    hipLaunchKernelGGL(mask_resample_kernel, dim3(grid_for((long long)out_h * out_w, 4096)), dim3(256), 0, as_stream(stream), dense, h, w, up_scale,
                       up_h, up_w, crop_h, crop_w, sy, sx, out_h, out_w, out);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_preprocess_frames(const uint8_t* frames, int32_t T, int32_t H0, int32_t W0, int32_t new_h, int32_t new_w,
                                             int32_t pad_h, int32_t pad_w, const float mean[3], const float std[3], int32_t unit_scale,
                                             int32_t flip_channels, float* out, void* stream) {
    SS_CHECK_ARG(frames && out && mean && std, "preprocess_frames: null pointer");
    SS_CHECK_ARG(T >= 1 && H0 >= 1 && W0 >= 1 && new_h >= 1 && new_w >= 1 && pad_h >= new_h && pad_w >= new_w,
                 "preprocess_frames: bad dims (%d, %d, %d) -> (%d, %d) padded (%d, %d)", T, H0, W0, new_h, new_w, pad_h, pad_w);
    SS_CHECK_ARG(std[0] != 0.f && std[1] != 0.f && std[2] != 0.f, "preprocess_frames: zero std");
    const float sy = (float)H0 / (float)new_h, sx = (float)W0 / (float)new_w;       // size-driven bilinear: scale = in / out
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((long long)T * pad_h * pad_w, 8192)), dim3(256), 0, as_stream(stream), frames, T, H0, W0,
                       new_h, new_w, pad_h, pad_w, sy, sx, mean[0], mean[1], mean[2], std[0], std[1], std[2], unit_scale, flip_channels, out);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

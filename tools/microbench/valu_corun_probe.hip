// Co-residency probe for the run-to-run differences of round 3 / 4.
//
// Observation it reproduces stand-alone (tools/soak_probe.py on the real pipeline): with three captured pipelines replaying on
// their own HIP streams, ~1 % of the steps differed from the lone replay, and EVERY such step started with 5-13 wrong words
// inside one 16-word run of the stem convolution's output -- one accumulator register, lanes 48..63 of one wave, off by a few
// products.  The stem was the only VALU-bound kernel of the step and its inner loop was all v_pk_fma_f32.
//
// This program runs a "victim" kernel whose result is known exactly -- the stem's inner loop (LDS broadcast reads of the weights,
// two pixels per thread, 64 accumulators), with the FMA issued either as v_pk_fma_f32 or as v_fma_f32 -- on one stream while an
// "aggressor" runs on a second stream:
//     none | mfma (back-to-back v_mfma_f32_32x32x16_f16) | valu (a second victim) | copy (a streaming copy) |
//     lds + mfma (eight-wave workgroups streaming ds_read_b128 fragments into MFMAs, barriers and LDS refills: the convs' k-loop)
// and counts victim outputs that differ from the exact value, with the lane of every wrong word.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/microbench/bin/valu_corun_probe tools/microbench/valu_corun_probe.hip
// Run:   tools/microbench/bin/valu_corun_probe [launches per configuration, default 200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int TAPS = 147, ROWS = 8, PR = 2 * ROWS + 5, PCP = 136;     // the stem's geometry: 8 x 64 outputs per workgroup

// weights[tap][64] and the input patch are small integers: every product and partial sum is exact in fp32, so the expected
// output is the same whatever the summation order
__device__ __host__ inline float wval(int tap, int k) { return (float)(((tap * 7 + k * 3) % 5) - 2); }
__device__ __host__ inline float pval(int c, int y, int x) { return (float)(((c * 5 + y * 3 + x) % 7) - 3); }

template <int PACKED>
__global__ __launch_bounds__(256, 2) void victim(float* __restrict__ out, int reps) {
    __shared__ __attribute__((aligned(16))) float lds[3 * PR * PCP + TAPS * 64];
    float* patch = lds;
    float* wl = lds + 3 * PR * PCP;
    for (int i = threadIdx.x; i < TAPS * 64; i += 256) wl[i] = wval(i / 64, i % 64);
    for (int i = threadIdx.x; i < 3 * PR * PCP; i += 256) { const int x = i % PCP, r = i / PCP; patch[i] = pval(r / PR, r % PR, x + (int)(blockIdx.x % 13)); }
    __syncthreads();
    const int py = threadIdx.x >> 5, px = threadIdx.x & 31;
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 1
        for (int hc = 0; hc < 2; ++hc) {
            v2f acc[32];                                    // (pixel 0, pixel 1) of 32 channels
#pragma unroll
            for (int k = 0; k < 32; ++k) acc[k] = (v2f){0.f, 0.f};
#pragma unroll 1
            for (int cdy = 0; cdy < 21; ++cdy) {
                const int c = cdy / 7, dy = cdy - 7 * c;
                const float* prow = patch + (c * PR + 2 * py + dy) * PCP + 2 * px;
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    const v2f v = (v2f){prow[dx], prow[dx + 64]};
                    const float4* w4 = reinterpret_cast<const float4*>(wl + (cdy * 7 + dx) * 64 + hc * 32);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float4 wv = w4[k];
                        const float ws[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (PACKED) {
                                const v2f w2 = (v2f){ws[j], ws[j]};
                                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[4 * k + j]) : "v"(w2), "v"(v));
                            } else {
                                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[4 * k + j].x) : "v"(ws[j]), "v"(v.x));
                                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[4 * k + j].y) : "v"(ws[j]), "v"(v.y));
                            }
                        }
                    }
                }
            }
            if (rep == reps - 1) {
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    float* o = out + (((size_t)blockIdx.x * 64 + hc * 32 + k) * ROWS + py) * 64;
                    o[px] = acc[k].x;
                    o[px + 32] = acc[k].y;
                }
            } else {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 32; ++k) s += acc[k].x + acc[k].y;
                if (s == 1.2345e30f) out[0] = s;            // (keeps the earlier repetitions alive)
            }
        }
    }
}

static float expected(int block, int ch, int py, int x) {      // x in 0..63
    double s = 0;
    for (int c = 0; c < 3; ++c)
        for (int dy = 0; dy < 7; ++dy)
            for (int dx = 0; dx < 7; ++dx)
                s += (double)wval((c * 7 + dy) * 7 + dx, ch) * (double)pval(c, 2 * py + dy, 2 * (x & 31) + dx + (x >= 32 ? 64 : 0) + block % 13);
    return (float)s;
}

__global__ __launch_bounds__(256) void aggr_mfma(float* sink, int iters) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(float)((threadIdx.x + j) & 3); b[j] = (_Float16)(float)((threadIdx.x * 3 + j) & 1); }
    f16v c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

// the split-staged convolutions' k-loop in miniature: an eight-wave workgroup streaming ds_read_b128 fragments out of 96 KB of LDS
// into back-to-back MFMAs (the LDS pipe ~80 % busy), with a barrier and an LDS refill per "chunk"
__global__ __launch_bounds__(512, 2) void aggr_lds_mfma(float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[24576];
    for (int i = threadIdx.x; i < 24576; i += 512) smem[i] = (float)(i & 7);
    __syncthreads();
    f16v c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    const char* base = reinterpret_cast<const char*>(smem) + (threadIdx.x & 63) * 16;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const h8 a = *reinterpret_cast<const h8*>(base + ((g * 4 + 0) & 47) * 2048);
            const h8 b = *reinterpret_cast<const h8*>(base + ((g * 4 + 1) & 47) * 2048);
            const h8 a2 = *reinterpret_cast<const h8*>(base + ((g * 4 + 2) & 47) * 2048);
            const h8 b2 = *reinterpret_cast<const h8*>(base + ((g * 4 + 3) & 47) * 2048);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b2, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, c3, 0, 0, 0);
        }
        __syncthreads();
        smem[(threadIdx.x * 4 + i) % 24576] = c0[0] * 0.f + (float)(i & 3);
        __syncthreads();
    }
    sink[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ __launch_bounds__(256) void aggr_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

// device-side comparison with the exact values: wrong words by lane quarter + the first few indices
__global__ __launch_bounds__(256) void check(const float* __restrict__ got, const float* __restrict__ want, size_t n, unsigned long long* hist, unsigned long long* first) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (got[i] != want[i]) {
            const int x = (int)(i % 64), py = (int)((i / 64) % ROWS);
            const int lane = (py & 1) * 32 + (x & 31);
            atomicAdd(&hist[lane >> 4], 1ull);
            const unsigned long long slot = atomicAdd(&hist[4], 1ull);
            if (slot < 8) first[slot] = i;
        }
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    const int NB = 6720;                                        // the stem's grid at 4 x 8 frames of 480 x 864
    const size_t n_out = (size_t)NB * 64 * ROWS * 64;
    float *out, *out2, *sink;
    float4 *cin, *cout;
    const size_t ncopy = (size_t)64 << 20;                      // 1 GB in, 1 GB out
    CK(hipMalloc(&out, n_out * 4)); CK(hipMalloc(&out2, n_out * 4)); CK(hipMalloc(&sink, 4096 * 512 * 4));
    CK(hipMalloc(&cin, ncopy * 16)); CK(hipMalloc(&cout, ncopy * 16)); CK(hipMemset(cin, 1, ncopy * 16));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    std::vector<float> want(n_out);
    float* want_d;
    unsigned long long *hist_d, *first_d;
    CK(hipMalloc(&want_d, n_out * 4)); CK(hipMalloc(&hist_d, 5 * 8)); CK(hipMalloc(&first_d, 8 * 8));
    for (int b = 0; b < NB; ++b)
        for (int ch = 0; ch < 64; ++ch)
            for (int py = 0; py < ROWS; ++py)
                for (int x = 0; x < 64; ++x) want[(((size_t)b * 64 + ch) * ROWS + py) * 64 + x] = expected(b, ch, py, x);
    CK(hipMemcpy(want_d, want.data(), n_out * 4, hipMemcpyHostToDevice));
    const char* vname[2] = {"v_fma_f32   ", "v_pk_fma_f32"};
    const char* aname[5] = {"none", "mfma", "valu (second victim)", "copy", "lds + mfma (conv-like)"};
    for (int v = 1; v >= 0; --v)
        for (int a = 0; a < 5; ++a) {
            long long bad_words = 0, bad_launches = 0;
            int reported = 0;
            long long lane_hist[4] = {0, 0, 0, 0};              // wrong words by lane quarter (0-15, 16-31, 32-47, 48-63)
            for (int l = 0; l < launches; ++l) {
                CK(hipMemsetAsync(out, 0xff, n_out * 4, s1));
                CK(hipStreamSynchronize(s1));
                if (a == 1) hipLaunchKernelGGL(aggr_mfma, dim3(4096), dim3(256), 0, s2, sink, 6000);
                if (a == 2) { if (v) hipLaunchKernelGGL(victim<1>, dim3(NB), dim3(256), 0, s2, out2, 1); else hipLaunchKernelGGL(victim<0>, dim3(NB), dim3(256), 0, s2, out2, 1); }
                if (a == 4) hipLaunchKernelGGL(aggr_lds_mfma, dim3(1024), dim3(512), 0, s2, sink, 150);
                if (a == 3) hipLaunchKernelGGL(aggr_copy, dim3(2048), dim3(256), 0, s2, cin, cout, ncopy);
                if (v) hipLaunchKernelGGL(victim<1>, dim3(NB), dim3(256), 0, s1, out, 1); else hipLaunchKernelGGL(victim<0>, dim3(NB), dim3(256), 0, s1, out, 1);
                CK(hipDeviceSynchronize());
                CK(hipMemset(hist_d, 0, 5 * 8));
                hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, s1, out, want_d, n_out, hist_d, first_d);
                unsigned long long h[5], f[8];
                CK(hipMemcpy(h, hist_d, 5 * 8, hipMemcpyDeviceToHost));
                if (h[4]) {
                    ++bad_launches;
                    bad_words += (long long)h[4];
                    for (int q = 0; q < 4; ++q) lane_hist[q] += (long long)h[q];
                    CK(hipMemcpy(f, first_d, 8 * 8, hipMemcpyDeviceToHost));
                    for (unsigned long long k = 0; k < h[4] && k < 8 && reported < 12; ++k, ++reported) {
                        const size_t i = f[k];
                        float g;
                        CK(hipMemcpy(&g, out + i, 4, hipMemcpyDeviceToHost));
                        const int x = (int)(i % 64), py = (int)((i / 64) % ROWS);
                        printf("    wrong word: block %zu ch %zu row %d x %d (lane %d): got %g, exact %g\n", i / (64 * ROWS * 64), (i / (ROWS * 64)) % 64, py, x,
                               (py & 1) * 32 + (x & 31), g, want[i]);
                    }
                }
            }
            printf("victim %s, aggressor %-22s: %lld of %d launches wrong, %lld wrong words; by lane quarter [0-15 16-31 32-47 48-63] = %lld %lld %lld %lld\n",
                   vname[v], aname[a], bad_launches, launches, bad_words, lane_hist[0], lane_hist[1], lane_hist[2], lane_hist[3]);
            fflush(stdout);
        }
    return 0;
}

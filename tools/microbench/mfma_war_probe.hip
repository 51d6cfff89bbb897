// Hazard probe: a run of MFMAs reading a VGPR quad as the A operand, immediately followed by a global_load_dwordx4 INTO that quad
// (L2-hot data).  If a queued MFMA picked up its operands after the load returned, its result would contain the loaded values.
// Two waves per SIMD keep the matrix pipe contended.  Build: hipcc --offload-arch=gfx950 -O2 mfma_war_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM>
__global__ __launch_bounds__(512, 2) void victim(const f32x4* __restrict__ g, unsigned long long* bad, int iters) {
    const int tid = threadIdx.x;
    unsigned long long nbad = 0;
    h8 b;
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)1.0f;
    for (int it = 0; it < iters; ++it) {
        f16v acc[NM];
        for (int m = 0; m < NM; ++m)
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        h8 a;
        for (int j = 0; j < 8; ++j) a[j] = (_Float16)1.0f;                    // A = ones, B = ones: every output = 16
        const f32x4* src = g + ((blockIdx.x * 512 + tid + it * 64) & 65535);
        if constexpr (NM == 6)
            asm volatile("v_mfma_f32_32x32x16_f16 %1, %0, %7, %1\n\t" "v_mfma_f32_32x32x16_f16 %2, %0, %7, %2\n\t"
                         "v_mfma_f32_32x32x16_f16 %3, %0, %7, %3\n\t" "v_mfma_f32_32x32x16_f16 %4, %0, %7, %4\n\t"
                         "v_mfma_f32_32x32x16_f16 %5, %0, %7, %5\n\t" "v_mfma_f32_32x32x16_f16 %6, %0, %7, %6\n\t"
                         "global_load_dwordx4 %0, %8, off\n\t"
                         "s_waitcnt vmcnt(0)\n\t" "s_nop 15\n\t" "s_nop 15"
                         : "+v"(a), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]) : "v"(b), "v"(src) : "memory");
        for (int m = 0; m < NM; ++m)
            for (int r = 0; r < 16; ++r)
                if (acc[m][r] != 16.0f) { ++nbad; break; }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    f32x4* g; unsigned long long* bad;
    hipMalloc(&g, 65536 * 16 + 4096 * 16); hipMemset(g, 0x3c, 65536 * 16 + 4096 * 16);       // fp16 0x3c3c = 1.06: a late operand read gives != 16
    hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(bad, 0, 8);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 10; ++rep) {
            hipLaunchKernelGGL(victim<6>, dim3(512), dim3(512), 0, s1, g, bad, 2000);
            if (mode == 1) hipLaunchKernelGGL(victim<6>, dim3(512), dim3(512), 0, s2, g, bad, 2000);
        }
        hipDeviceSynchronize();
        unsigned long long h = 0; (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        printf("%s: %llu wrong accumulator sets\n", mode ? "two streams" : "one stream", h);
    }
    return 0;
}

// Hazard probe: ds_write_b128 of a VGPR quad immediately followed by a global_load_dwordx4 INTO the same quad.  If the LDS unit picks
// up the store data later than the (L2-hot) load returns, LDS ends up holding the loaded values instead of the stored ones.
// A second kernel on another stream hammers the LDS pipe of the same CUs.  Build: hipcc --offload-arch=gfx950 -O2 lds_war_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void victim(const f32x4* __restrict__ g, unsigned long long* bad, int iters, int pre_reads) {
    __shared__ f32x4 lds[512 * 9];
    const int tid = threadIdx.x;
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        // queue some LDS reads in front of the store (their results are consumed later)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        f32x4 keep[8];
        for (int r = 0; r < 8; ++r) keep[r] = (r < pre_reads) ? lds[512 * (r + 1) + ((tid * 7 + r) & 511)] : acc;
        f32x4 v = {(float)(it + 1), (float)tid, 3.0f, 4.0f};
        const f32x4* src = g + ((blockIdx.x * 512 + tid + it * 64) & 65535);
        const unsigned int addr = (unsigned int)(uintptr_t)(&lds[tid]);          // LDS byte address
        asm volatile("ds_write_b128 %1, %0\n\t"
                     "global_load_dwordx4 %0, %2, off\n\t"
                     "s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(v) : "v"(addr), "v"(src) : "memory");
        __syncthreads();
        const f32x4 back = lds[tid];
        if (back.x != (float)(it + 1) || back.y != (float)tid || back.z != 3.0f || back.w != 4.0f) ++nbad;
        for (int r = 0; r < 8; ++r) acc += keep[r];
        if (acc.x == 12345.f) lds[512 + tid] = v + acc;                           // keep everything alive
        __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
}

__global__ __launch_bounds__(256) void hammer(float* out, int iters) {
    __shared__ f32x4 lds[256 * 16];
    const int tid = threadIdx.x;
    for (int i = 0; i < 16; ++i) lds[i * 256 + tid] = f32x4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += lds[i * 256 + ((tid + it) & 255)];
        lds[(it & 15) * 256 + tid] = acc;
    }
    if (acc.x == 12345.f) out[tid] = acc.y;
}

int main(int argc, char** argv) {
    const int pre = argc > 1 ? atoi(argv[1]) : 8;
    f32x4* g; unsigned long long* bad; float* out;
    hipMalloc(&g, 65536 * 16 + 4096 * 16); hipMemset(g, 0x7f, 65536 * 16 + 4096 * 16);       // loaded values: 0x7f7f7f7f (not the stored ones)
    hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    hipMalloc(&out, 4096);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(bad, 0, 8);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 20; ++rep) {
            if (mode == 1) hipLaunchKernelGGL(hammer, dim3(512), dim3(256), 0, s2, out, 20000);
            hipLaunchKernelGGL(victim, dim3(256), dim3(512), 0, s1, g, bad, 2000, pre);
        }
        hipDeviceSynchronize();
        unsigned long long h = 0; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        printf("pre_reads %d, %s: %llu corrupted LDS quads of %llu\n", pre, mode ? "with LDS hammer on a second stream" : "alone", h, 20ull * 256 * 512 * 2000);
    }
    return 0;
}

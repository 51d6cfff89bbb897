// Back-to-back GEMM for the bottleneck tail (f16x3 mode): conv3 of block b (1x1, MID -> 4 MID, + bias + identity + ReLU) and conv1 of
// block b + 1 (1x1, 4 MID -> MID, + bias + ReLU) in ONE launch.
//
// Reference: /root/reference/stemseg/modeling/backbone/resnet.py:262-282 (Bottleneck.forward: conv1 -> conv2 -> conv3 -> += identity ->
// relu), two consecutive blocks of a stage (:105-113).  As two launches of conv_igemm.hip the 4 MID-channel block output is written by
// conv3 and read again by the next conv1 (and later by the next conv3 as its identity): at layer 3 (MID = 256, 32 frames) 212 MB of the
// 742 MB the pair moves.  Here a workgroup owns 256 positions and walks ALL 4 MID output channels of conv3 in co-tiles of CT: the co-tile's
// accumulators get scale / bias / identity / ReLU in registers, are written to HBM once (the next block's identity needs them) and -- split
// into the fp16 (hi, lo * 2^11) pairs -- ARE the B operand of conv1's next K-chunk: the MFMA C/D layout gives a lane 16 of a 32-channel
// tile's rows for its own position, a v_permlane32_swap between the lane halves turns them into the two 16-deep k-groups in the standalone
// kernel's k order.  conv1's MID x 256 accumulators stay in registers through the whole launch.
//
// Same operands, same split arithmetic (split_pair_f16), same k order per accumulator as conv_igemm_kernel's 1x1 tiles (32-channel chunks, two
// k-groups of one tap x 16 channels, products lo_w*hi_x, (hi_w 2^-11)*(lo_x 2^11), hi_w*hi_x) => BIT-IDENTICAL to the two launches whenever
// those run without split-K (tests/test_gpu_fused_tail.py).
//
// Data movement: everything staged by LDS-DMA (global_load_lds_dwordx4, no staging registers, no VALU):
//   * conv3's input -- conv2's output -- arrives ALREADY SPLIT: conv2's epilogue writes the fp16 pair planes ("P16", conv_igemm.hip) instead
//     of fp32, same 4 bytes per value, in octets: word [plane][channel / 8][position][(channel % 8) / 2] -- a lane's eight k-values of a
//     k-group are ONE ds_read_b128;
//   * weights in the packed f16x3 layout of stemseg_hip_pack_conv_weight_prec, unchanged: conv3's [chunk][grp][plane][half][4 MID][16 B]
//     has the co-tile's CT channels as one 16 CT-byte run per row, conv1's chunks [..][MID][16 B] for a co-tile are one contiguous block.
// Stages: per co-tile MID / 32 conv3 sub-stages (x chunk + w3 chunk double-buffered, a share of the co-tile's w1 block streamed in beside
// them), then ONE epilogue + conv1 stage.  One barrier per stage, vmcnt(0) in front of it.
#include "common.h"

// Timing probes / A-B switches: EXPERIMENT builds only (-DSS_EXPERIMENTS), as in conv_igemm.hip.  SS_FT_PROBE (results WRONG for any value but 0):
// 1 no identity loads and no block-output stores (the kernel's HBM traffic), 2 no LDS-DMA of the input chunks, 3 no MFMAs, 4 (R1 form) no weight DMA,
// 5 (R1 form) identity loads but no output stores, 6 (R1 form) output stores but no identity loads.
// SS_FT_STAGGER n: workgroup b starts (b % 16) * n * ~0.5 us late (spreads the workgroups' HBM bursts; same results).
#ifndef SS_EXPERIMENTS
#if defined(SS_FT_PROBE) || defined(SS_FT_STAGGER)
#error "SS_FT_PROBE / SS_FT_STAGGER are experiment switches: add -DSS_EXPERIMENTS"
#endif
#define SS_FT_PROBE 0
#define SS_FT_STAGGER 0
#else
#ifndef SS_FT_PROBE
#define SS_FT_PROBE 0
#endif
#ifndef SS_FT_STAGGER
#define SS_FT_STAGGER 0
#endif
#endif

namespace stemseg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct FusedTailParams {
    const unsigned int* x16;     // conv2's output as fp16 pair planes: [2][MID / 8][V][4] words
    const char* w3;              // packed f16x3 weights of conv3 (Cout = 4 MID, Cin = MID, 1 tap) + float inv[4 MID] behind the slabs
    const float* b3;
    const float* res;            // identity: dense [4 MID][V]
    float* y;                    // block output: dense [4 MID][V]
    const char* w1;              // packed f16x3 weights of the next block's conv1 (Cout = MID, Cin = 4 MID) + float inv[MID]
    const float* b1;
    float* z;                    // conv1 output (+ bias, ReLU) at (t, y, x) of a [MID][T][h + 2][pitch] zero-haloed interior view
    int64_t z_cs, z_ts, z_ys;
    int dec_H, dec_W;
    int V;                       // positions: T * h * w
};

// the split of conv_igemm.hip (split_pair_f16), bit for bit: hi = fp16(x / 4), lo = fp16((x / 4 - hi) * 2^11) of a channel pair
__device__ __forceinline__ void ft_split_pair(const float x0, const float x1, unsigned int& hw, unsigned int& lw) {
    const float qs = 0.25f, ks = 2048.0f;
    unsigned int h, l;
    float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "s"(qs));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "s"(qs));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "s"(qs), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "s"(qs), "v"(h));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(l) : "v"(r0), "s"(ks));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(l) : "v"(r1), "s"(ks));
    hw = h;
    lw = l;
}

template <int MID_, int CT_, int P_>
struct FusedTailCfg {
    static constexpr int MID = MID_, CT = CT_, COUT = 4 * MID_;
    static constexpr int P = P_, NW = P_ / 32, NTHREADS = 64 * NW;   // positions per workgroup; one 32-position column block per wave
    static constexpr int NC3 = MID / 32;                   // conv3 K-chunks (32 channels)
    static constexpr int NCT = COUT / CT;                  // co-tiles
    static constexpr int MI3 = CT / 32, MI1 = MID / 32;    // 32-row accumulator tiles per wave: conv3 co-tile, conv1
    static constexpr int NH = 2 * MI3;                     // conv1 K-halves (k-groups of 16 channels) per co-tile
    static constexpr int X_BYTES = 2 * 4 * P * 16;          // x chunk: [plane][octet][position][4 words]
    static constexpr int W3_BYTES = 8 * CT * 16;            // w3 chunk of the co-tile: [grp][plane][half][CT][16 B]
    static constexpr int W1_HALF = 4 * MID * 16;            // one k-group of w1: [plane][half][MID][16 B]; a co-tile's NH of them are contiguous in the packed blob
    static constexpr int LDS_BYTES = 2 * X_BYTES + 2 * W3_BYTES + 2 * W1_HALF;
    static constexpr int NQ = P / 64;                       // 64-position quarters of an x row
    static constexpr int X_PIECES = X_BYTES / 1024, W3_PIECES = W3_BYTES / 1024, W1_PIECES = W1_HALF / 1024;     // 1 KB wave-instructions
    static_assert(P % 64 == 0 && MID % 32 == 0 && CT % 32 == 0 && COUT % CT == 0, "tile shapes");
    static_assert(LDS_BYTES <= (P <= 128 ? 80 : 160) * 1024, "LDS: two 128-position workgroups, or one 256-position workgroup, per CU");
    static_assert(W3_BYTES % 1024 == 0 && W1_HALF % 1024 == 0 && X_PIECES % NW == 0, "whole DMA pieces");
    static_assert(MI1 * 16 + MI3 * 16 <= 176, "accumulators must leave room for fragments at 256 registers");
};

// Stages of a workgroup, one barrier (behind a vmcnt(0)) each; every stage first requests what the NEXT stage reads (LDS-DMA into the idle
// buffer of its kind), then computes:
//   per co-tile j:  S(c), c < MID / 32   conv3 MFMAs of K-chunk c (x chunk c + w3 chunk (j, c)); the last one also requests w1 half 0 and the
//                                        identity values of the co-tile's first 32-channel tile
//                   E(h), h < 2 CT / 32  h even: scale / bias / identity / ReLU / store / split of 32-channel tile h / 2, its two k-groups as B
//                                        fragments; then conv1's MFMAs of k-group h (w1 half h); the last one requests the next co-tile's S(0)
// A stage is 0.4-0.8 us of matrix work against ~1 us from a DMA request to its landing, so one workgroup alone spends most of its time at the
// barrier (measured: the 256-position form ran at 34 % MFMA-pipe occupancy, and removing ALL its MFMAs saved 9 %).  The 128-position form is
// therefore the default: 80 KB of LDS, four waves -- TWO workgroups per CU, each filling the other's waits.
template <class C>
__global__ __launch_bounds__(C::NTHREADS, 2) void fused_tail_kernel(const FusedTailParams p) {
#pragma clang fp contract(off)       // scale, bias and identity are separately rounded steps in the standalone kernels: no multiply-add contraction here
    __shared__ __attribute__((aligned(1024))) char smem[C::LDS_BYTES];
    char* const xbuf = smem;                               // 2 buffers
    char* const w3buf = smem + 2 * C::X_BYTES;             // 2 buffers
    char* const w1buf = smem + 2 * C::X_BYTES + 2 * C::W3_BYTES;      // 2 buffers (k-group halves)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int pos0 = blockIdx.x * C::P;
    const int pos = pos0 + wave * 32 + l31;                // this lane's position (column of every accumulator tile of the wave)
    const bool pos_ok = pos < p.V;
    const int pos_c = min(pos, p.V - 1);                    // (loads of a column past V read the last valid one: the column is never stored)
    const int64_t V = p.V;
    // Addressing: every global access below is (wave-uniform 64-bit base in SGPRs) + (ONE loop-invariant 32-bit per-lane byte offset), so that
    // no 64-bit per-lane pointer stays live across the stages (the first build kept ~50 of them and spilled them around every DMA issue).
    // The channel of accumulator register r of lane half h is base + (r & 3) + 8 (r >> 2) + 4 h: the 4 h part goes into the lane offset.
    const unsigned int lane_off = (unsigned int)(((int64_t)(4 * half) * V + pos_c) * 4);      // bytes: (4 half) channels down, this lane's column
    auto opaque = [](unsigned int v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };   // (keeps address arithmetic inside its stage)

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float* inv3 = reinterpret_cast<const float*>(p.w3 + (int64_t)C::NC3 * 8 * C::COUT * 16);
    const float* inv1 = reinterpret_cast<const float*>(p.w1 + (int64_t)(C::COUT / 32) * 8 * C::MID * 16);

    // ---- DMA issue helpers (each call = this wave's share; 1 KB per wave instruction, LDS image lane-linear) -------------------------
    const int xpos = min(pos0 + lane, p.V - 1);             // (clamped: columns past V are computed and never stored)
    auto dma_x = [&](const int c, const int buf) __attribute__((always_inline)) {       // x chunk c: (plane, octet, 64-position quarter) pieces
#if SS_FT_PROBE == 2
        return;
#endif
#pragma unroll
        for (int k = 0; k < C::X_PIECES / C::NW; ++k) {
            const int idx = wave * (C::X_PIECES / C::NW) + k, q = idx % C::NQ, o = (idx / C::NQ) & 3, pl = idx / (4 * C::NQ);
            const unsigned int xp = (unsigned int)min((int)opaque((unsigned int)xpos) + q * 64, p.V - 1);
            const char* src = reinterpret_cast<const char*>(p.x16) + ((int64_t)(pl * (C::MID / 8) + 4 * c + o)) * V * 16 + (size_t)(xp * 16u);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xbuf + buf * C::X_BYTES + ((pl * 4 + o) * C::P + q * 64) * 16), 16, 0, 0);
        }
    };
    auto dma_w3 = [&](const int j, const int c, const int buf) __attribute__((always_inline)) {      // w3 chunk (co-tile j, K-chunk c): 8 rows of CT x 16 B
        constexpr int PER_ROW = C::CT * 16 / 1024;
#pragma unroll
        for (int k = 0; k < (C::W3_PIECES + C::NW - 1) / C::NW; ++k) {
            const int idx = wave + k * C::NW;
            if (idx < C::W3_PIECES) {
                const int row = idx / PER_ROW, part = idx % PER_ROW;
                const char* src = p.w3 + ((int64_t)c * 8 + row) * (C::COUT * 16) + (int64_t)j * (C::CT * 16) + part * 1024 + (size_t)(opaque((unsigned int)lane) * 16u);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(w3buf + buf * C::W3_BYTES + idx * 1024), 16, 0, 0);
            }
        }
    };
    auto dma_w1 = [&](const int j, const int h, const int buf) __attribute__((always_inline)) {      // k-group half h of co-tile j's w1 block
#pragma unroll
        for (int k = 0; k < (C::W1_PIECES + C::NW - 1) / C::NW; ++k) {
            const int idx = wave + k * C::NW;
            if (idx < C::W1_PIECES) {
                const char* src = p.w1 + ((int64_t)j * C::NH + h) * C::W1_HALF + (int64_t)idx * 1024 + (size_t)(opaque((unsigned int)lane) * 16u);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(w1buf + buf * C::W1_HALF + idx * 1024), 16, 0, 0);
            }
        }
    };
    // the co-tile's per-channel (1 / scale, bias): CT + CT floats into the head of the w3 buffer that idles through the epilogue stages
    // (lanes 0 .. CT / 4 - 1 of wave 0 carry the scales, the next CT / 4 lanes the biases)
    auto dma_tab = [&](const int j, const int buf) __attribute__((always_inline)) {
        static_assert(2 * (C::CT / 4) <= 64 && 2 * C::CT * 4 <= C::W3_BYTES, "the table is one DMA piece");
        if (wave == 0 && lane < 2 * (C::CT / 4)) {
            const unsigned int l = opaque((unsigned int)lane);
            const float* src = (l < C::CT / 4 ? inv3 : p.b3 - C::CT) + (int64_t)j * C::CT + (size_t)(l * 4u);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(w3buf + buf * C::W3_BYTES), 16, 0, 0);
        }
    };
    // conv1's own (1 / scale, bias): MID + MID floats into the (by then idle) x buffer, requested in the very last stage
    auto dma_tail_tab = [&]() __attribute__((always_inline)) {
        constexpr int PIECES = (2 * C::MID + 255) / 256;
        if (wave < PIECES) {
            const unsigned int f0 = (unsigned int)wave * 256u + opaque((unsigned int)lane) * 4u;
            if (f0 < 2u * C::MID) {
                const float* src = (f0 < (unsigned int)C::MID ? inv1 : p.b1 - C::MID) + (size_t)f0;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xbuf + wave * 1024), 16, 0, 0);
            }
        }
    };
    auto stage_sync = [&]() __attribute__((always_inline)) {
        // vmcnt(0): this wave's DMA pieces (and loads / stores) have landed.  The BUILTIN, not inline assembly: the compiler's wait-count
        // pass must see it, or it guards the first use of a value loaded a stage earlier with a vmcnt(0) of its own -- behind the DMA
        // just issued for the next stage
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    };
    auto load_identity = [&](const int co_first, float (&rres)[16]) __attribute__((always_inline)) {   // rows of the 32-channel tile at co_first, this lane's column
#pragma unroll
        for (int r = 0; r < 16; ++r)
#if SS_FT_PROBE == 1
            rres[r] = 0.f;
#else
            rres[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.res + (int64_t)(co_first + (r & 3) + 8 * (r >> 2)) * V) + (size_t)lane_off);
#endif
    };

    f32x16 acc1[C::MI1];
#pragma unroll
    for (int m = 0; m < C::MI1; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[m][r] = 0.f;

    const f16x8 k2048 = {(_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f),
                         (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f)};

    // three products per (A, B) fragment pair, smallest first -- the order of conv_igemm.hip's f16x3 stream
    auto mma3 = [&](f32x16& acc, const f16x8 a_hi, const f16x8 a_lo, const f16x8 b_hi, const f16x8 b_lo) __attribute__((always_inline)) {
        const f16x8 a_his = a_hi * k2048;                    // hi_w * 2^-11 (exact: a power of two on a normal number)
#if SS_FT_PROBE == 3
        asm volatile("" ::"v"(a_his), "v"(a_lo), "v"(b_hi), "v"(b_lo));
        return;
#endif
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_his, b_lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc, 0, 0, 0);
    };

#if SS_FT_STAGGER > 0
    for (int i = 0; i < (int)(blockIdx.x % 16) * SS_FT_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);
#endif
    dma_x(0, 0);
    dma_w3(0, 0, 0);
    stage_sync();
    int sb = 0;                                              // x / w3 buffer of the conv3 sub-stage about to run
#pragma unroll 1
    for (int j = 0; j < C::NCT; ++j) {
        f32x16 acc3[C::MI3];
#pragma unroll
        for (int m = 0; m < C::MI3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[m][r] = 0.f;
        const int co_j = j * C::CT;
        float rres[16];                                      // identity values of the 32-channel tile whose epilogue comes next
#pragma unroll 1
        for (int c = 0; c < C::NC3; ++c) {
            if (c + 1 < C::NC3) { dma_x(c + 1, sb ^ 1); dma_w3(j, c + 1, sb ^ 1); }
            else { dma_w1(j, 0, 0); dma_tab(j, sb ^ 1); load_identity(co_j, rres); }
            const char* xb = xbuf + sb * C::X_BYTES + (half * C::P + wave * 32 + l31) * 16;
            const char* wb = w3buf + sb * C::W3_BYTES + (half * C::CT + l31) * 16;
            // software pipeline over the (k-group, 32-row tile) steps: the A fragments of step s + 1 are requested before the MFMAs of step s
            // issue; fenced, or the scheduler hoists every fragment of the stage to its top (and spills) or sinks each to its use
            constexpr int NS = 2 * C::MI3;
            f16x8 b_hi[2], b_lo[2], a_hi[2], a_lo[2];
            auto ld_b = [&](const int g, const int k) __attribute__((always_inline)) {
                b_hi[k] = *reinterpret_cast<const f16x8*>(xb + ((0 * 4 + 2 * g) * C::P) * 16);
                b_lo[k] = *reinterpret_cast<const f16x8*>(xb + ((1 * 4 + 2 * g) * C::P) * 16);
            };
            auto ld_a = [&](const int st, const int k) __attribute__((always_inline)) {
                const int g = st / C::MI3, m = st % C::MI3;
                a_hi[k] = *reinterpret_cast<const f16x8*>(wb + (((g * 2 + 0) * 2) * C::CT + m * 32) * 16);
                a_lo[k] = *reinterpret_cast<const f16x8*>(wb + (((g * 2 + 1) * 2) * C::CT + m * 32) * 16);
            };
            ld_b(0, 0);
            ld_b(1, 1);
            ld_a(0, 0);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                if (st + 1 < NS) ld_a(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                mma3(acc3[st % C::MI3], a_hi[st & 1], a_lo[st & 1], b_hi[st / C::MI3], b_lo[st / C::MI3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage_sync();
            sb ^= 1;
        }
        // ---- epilogue of the co-tile + its K-halves of conv1 ---------------------------------------------------------------------------
        f16x8 b_hi[2], b_lo[2];                              // the two k-groups of the current 32-channel tile as B fragments
#pragma unroll
        for (int h = 0; h < C::NH; ++h) {
            const int m = h >> 1, g = h & 1;
            if (g == 0) {
                const int co_m = co_j + m * 32;
                // scale back, + bias, + identity, ReLU, store; C/D layout: register r of lane (half, l31) = row (r & 3) + 8 (r >> 2) + 4 half
                unsigned int hw[8], lw[8];
                const float* tab = reinterpret_cast<const float*>(w3buf + sb * C::W3_BYTES);      // (dma_tab: requested in the last conv3 sub-stage)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co4 = co_m + 8 * q + 4 * half;
                    const float4 sc = *reinterpret_cast<const float4*>(tab + (co4 - co_j));
                    const float4 bv = *reinterpret_cast<const float4*>(tab + C::CT + (co4 - co_j));
                    // (explicitly rounded steps: the standalone kernel scales its accumulators in one place and adds bias / identity in another)
                    float v0 = __fmul_rn(acc3[m][4 * q + 0], sc.x), v1 = __fmul_rn(acc3[m][4 * q + 1], sc.y), v2 = __fmul_rn(acc3[m][4 * q + 2], sc.z),
                          v3 = __fmul_rn(acc3[m][4 * q + 3], sc.w);
                    v0 = relu_keep_nan(__fadd_rn(__fadd_rn(v0, bv.x), rres[4 * q + 0]));
                    v1 = relu_keep_nan(__fadd_rn(__fadd_rn(v1, bv.y), rres[4 * q + 1]));
                    v2 = relu_keep_nan(__fadd_rn(__fadd_rn(v2, bv.z), rres[4 * q + 2]));
                    v3 = relu_keep_nan(__fadd_rn(__fadd_rn(v3, bv.w), rres[4 * q + 3]));
                    if (pos_ok && SS_FT_PROBE != 1) {
                        char* yo = reinterpret_cast<char*>(p.y + (int64_t)(co_m + 8 * q) * V);       // (uniform; the 4 half rows are in lane_off)
                        *reinterpret_cast<float*>(yo + (size_t)lane_off) = v0;
                        *reinterpret_cast<float*>(yo + V * 4 + (size_t)lane_off) = v1;
                        *reinterpret_cast<float*>(yo + V * 8 + (size_t)lane_off) = v2;
                        *reinterpret_cast<float*>(yo + V * 12 + (size_t)lane_off) = v3;
                    }
                    ft_split_pair(v0, v1, hw[2 * q], lw[2 * q]);
                    ft_split_pair(v2, v3, hw[2 * q + 1], lw[2 * q + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                // k-groups of 16 channels in the standalone kernel's order: lane half 0 channels 0..7, half 1 channels 8..15 of the group.  The
                // lane halves hold (q even: ch 0-3 | 4-7), (q odd: 8-11 | 12-15): swapping the upper half of the q-even words with the lower half
                // of the q-odd words gives (0-3, 4-7) to half 0 and (8-11, 12-15) to half 1.
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    unsigned int xh[2], yh[2], xl[2], yl[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        auto sh = __builtin_amdgcn_permlane32_swap(hw[4 * gg + i], hw[4 * gg + 2 + i], false, false);
                        xh[i] = sh[0]; yh[i] = sh[1];
                        auto sl = __builtin_amdgcn_permlane32_swap(lw[4 * gg + i], lw[4 * gg + 2 + i], false, false);
                        xl[i] = sl[0]; yl[i] = sl[1];
                    }
                    const u32x4 bh = {xh[0], xh[1], yh[0], yh[1]}, bl = {xl[0], xl[1], yl[0], yl[1]};
                    b_hi[gg] = __builtin_bit_cast(f16x8, bh);
                    b_lo[gg] = __builtin_bit_cast(f16x8, bl);
                }
            } else if (m + 1 < C::MI3) {
                load_identity(co_j + (m + 1) * 32, rres);        // (they land under this k-group's MFMAs)
            }
            __builtin_amdgcn_sched_barrier(0);
            // what the next stage reads (requested behind the epilogue: its table reads are LDS reads the compiler cannot tell from the DMA's target)
            if (h + 1 < C::NH) dma_w1(j, h + 1, (h + 1) & 1);
            else if (j + 1 < C::NCT) { dma_x(0, sb); dma_w3(j + 1, 0, sb); }
            else dma_tail_tab();
            const char* w1b = w1buf + (h & 1) * C::W1_HALF + (half * C::MID + l31) * 16;
            f16x8 a_hi[2], a_lo[2];
            auto ld_a1 = [&](const int m1, const int k) __attribute__((always_inline)) {
                a_hi[k] = *reinterpret_cast<const f16x8*>(w1b + ((0 * 2) * C::MID + m1 * 32) * 16);
                a_lo[k] = *reinterpret_cast<const f16x8*>(w1b + ((1 * 2) * C::MID + m1 * 32) * 16);
            };
            ld_a1(0, 0);
#pragma unroll
            for (int m1 = 0; m1 < C::MI1; ++m1) {
                if (m1 + 1 < C::MI1) ld_a1(m1 + 1, (m1 + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                mma3(acc1[m1], a_hi[m1 & 1], a_lo[m1 & 1], b_hi[g], b_lo[g]);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage_sync();
        }
    }
    // ---- conv1's epilogue: scale back, + bias, ReLU, flat position -> (t, y, x) of the zero-haloed consumer layout -------------------------
    if (pos_ok) {
        const int hwp = p.dec_H * p.dec_W;
        const int t2 = pos / hwp, r2 = pos - t2 * hwp, y2 = r2 / p.dec_W, x2 = r2 - y2 * p.dec_W;
        const unsigned int z_off = (unsigned int)(((int64_t)(4 * half) * p.z_cs + (int64_t)t2 * p.z_ts + (int64_t)y2 * p.z_ys + x2) * 4);      // bytes
#pragma unroll
        for (int m1 = 0; m1 < C::MI1; ++m1)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co4 = m1 * 32 + 8 * q + 4 * half;
                const float4 sc = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xbuf) + co4);
                const float4 bv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xbuf) + C::MID + co4);
                char* zo = reinterpret_cast<char*>(p.z + (int64_t)(m1 * 32 + 8 * q) * p.z_cs);     // (uniform)
                *reinterpret_cast<float*>(zo + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][4 * q + 0], sc.x), bv.x));
                *reinterpret_cast<float*>(zo + p.z_cs * 4 + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][4 * q + 1], sc.y), bv.y));
                *reinterpret_cast<float*>(zo + p.z_cs * 8 + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][4 * q + 2], sc.z), bv.z));
                *reinterpret_cast<float*>(zo + p.z_cs * 12 + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][4 * q + 3], sc.w), bv.w));
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The 16-column form (MID = 256: layer 3, 22 of the 27 fused launches of an R-101 pass).  What bounds the 32-column form there is not its
// matrix work but its memory-side traffic (profiles/r06e_pmc_fused_tail_256_p128.txt: FETCH + WRITE 1.5 GB per launch against 0.53 GB of
// tensors): conv3's input tile -- 1 KB per position -- is re-read once per co-tile, 16 times, and with every position of the map in flight
// at once those re-reads can only come from the Infinity Cache.  It cannot stay in LDS next to the weight streams (128 KB for 128 positions),
// and as B fragments of 32 positions it is 128 registers beside 160 of accumulators.  With v_mfma_f32_16x16x32_f16 a wave owns 16 positions:
// conv1's accumulators are 64 registers, a 128-channel co-tile's 32, and the wave's WHOLE conv3 input -- 256 channels x 16 positions as
// fp16 pairs -- is 64 registers, loaded once.  LDS then holds weights only (two conv3 chunks and one conv1 k-step per stage, double
// buffered: 128 KB), a stage is 48 MFMAs per wave, and the kernel's traffic is its tensors'.
//   * one MFMA consumes a whole 32-channel chunk (k = 32): lane (n, kb) carries channels 8 kb .. 8 kb + 7 of the chunk -- for conv3 exactly the
//     packed weights' (k-group kb / 2, half kb % 2) vector and the producer's octet 4 s + kb;
//   * C/D layout: lane (n, rb) holds rows 4 rb .. 4 rb + 3 of a 16-row tile.  Two consecutive tiles (32 channels) of the finished co-tile become
//     conv1's next B fragment -- lane (n, kb): channels 8 kb .. 8 kb + 7, the packed w1's own order -- by one v_permlane32_swap and one
//     v_permlane16_swap per word.  (Leaving the rows where they are and fetching the matching A fragment as two 8-byte halves was measured first:
//     the compiler fuses the halves into ds_read2_b64, whose 16-lane groups hit 32 banks at a 16-byte stride -- 39 % of the LDS cycles were
//     bank conflicts, profiles/r06f_pmc_fused_tail16_first.txt.)
// The summation order inside an MFMA differs from the 32x32x16 kernels' (32 channels per instruction instead of two groups of 16), so this
// form agrees with the separate launches to fp32 round-off, not bit for bit; like every kernel here it has ONE summation order whatever the
// batch (no split-K, no dependence on the launch shape).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MID_>
struct FusedTail16Cfg {
    static constexpr int MID = MID_, COUT = 4 * MID_, CT = 128;
    static constexpr int NW = 8, NTHREADS = 512, P = 16 * NW;
    static constexpr int NC3 = MID / 32, NCT = COUT / CT;
    static constexpr int NT3 = CT / 16, NT1 = MID / 16;     // 16-row accumulator tiles: conv3 co-tile, conv1
    static constexpr int SC = 2;                            // conv3 chunks per stage
    static constexpr int NS = NC3 / SC, NU = CT / 32;       // conv3 stages / conv1 k-steps per co-tile
    static constexpr int W3_CHUNK = 8 * CT * 16;            // [grp][plane][half][CT][16 B]
    static constexpr int W3_BYTES = SC * W3_CHUNK;
    static constexpr int W1_BYTES = 8 * MID * 16;           // one 32-channel k-step of conv1: [grp][plane][half][MID][16 B]
    static constexpr int TAB_BYTES = 2 * CT * 4;
    static constexpr int LDS_BYTES = 2 * W3_BYTES + 2 * W1_BYTES + 2 * TAB_BYTES;
    static_assert(NC3 % SC == 0 && W3_BYTES % (1024 * NW) == 0 && W1_BYTES % (1024 * NW) == 0 && TAB_BYTES == 1024, "whole DMA pieces per wave");
    static_assert(LDS_BYTES <= 160 * 1024 && 2 * MID * 4 <= W3_BYTES, "LDS");
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS, 2) void fused_tail16_kernel(const FusedTailParams p) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(1024))) char smem[C::LDS_BYTES];
    char* const w3buf = smem;
    char* const w1buf = smem + 2 * C::W3_BYTES;
    char* const tabbuf = smem + 2 * C::W3_BYTES + 2 * C::W1_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;                 // column (position) of the wave's 16 / k-block (inputs) = row-block (accumulators)
    const int pos = blockIdx.x * C::P + wave * 16 + n;
    const bool pos_ok = pos < p.V;
    const int pos_c = min(pos, p.V - 1);
    const int64_t V = p.V;
    const unsigned int lane_off = (unsigned int)(((int64_t)(4 * kb) * V + pos_c) * 4);       // bytes: rows 4 kb .. of a tile, this lane's column
    auto opaque = [](unsigned int v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float* inv3 = reinterpret_cast<const float*>(p.w3 + (int64_t)C::NC3 * 8 * C::COUT * 16);
    const float* inv1 = reinterpret_cast<const float*>(p.w1 + (int64_t)(C::COUT / 32) * 8 * C::MID * 16);

    auto dma_w3 = [&](const int j, const int st, const int buf) __attribute__((always_inline)) {      // chunks SC st .. of co-tile j
        constexpr int PER_ROW = C::CT * 16 / 1024, PER_CHUNK = C::W3_CHUNK / 1024;
#pragma unroll
        for (int k = 0; k < C::W3_BYTES / 1024 / C::NW; ++k) {
            const int idx = wave + k * C::NW, sc = idx / PER_CHUNK, r = idx % PER_CHUNK, row = r / PER_ROW, part = r % PER_ROW;
            const char* src = p.w3 + ((int64_t)(st * C::SC + sc) * 8 + row) * (C::COUT * 16) + (int64_t)j * (C::CT * 16) + part * 1024 + (size_t)(opaque((unsigned int)lane) * 16u);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(w3buf + buf * C::W3_BYTES + idx * 1024), 16, 0, 0);
        }
    };
    auto dma_w1 = [&](const int cc, const int buf) __attribute__((always_inline)) {                   // conv1 k-step (32-channel chunk) cc
#pragma unroll
        for (int k = 0; k < C::W1_BYTES / 1024 / C::NW; ++k) {
            const int idx = wave + k * C::NW;
            const char* src = p.w1 + (int64_t)cc * C::W1_BYTES + (int64_t)idx * 1024 + (size_t)(opaque((unsigned int)lane) * 16u);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(w1buf + buf * C::W1_BYTES + idx * 1024), 16, 0, 0);
        }
    };
    auto dma_tab = [&](const int j) __attribute__((always_inline)) {        // co-tile j's (1 / scale | bias): lanes 0-31 scales, 32-63 biases
        if (wave == 0) {
            const unsigned int l = opaque((unsigned int)lane);
            const float* src = (l < 32 ? inv3 : p.b3 - C::CT) + (int64_t)j * C::CT + (size_t)(l * 4u);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tabbuf + (j & 1) * C::TAB_BYTES), 16, 0, 0);
        }
    };
    auto dma_tail_tab = [&]() __attribute__((always_inline)) {            // conv1's own (1 / scale | bias) into the idle w3 buffer
        constexpr int PIECES = (2 * C::MID + 255) / 256;
        if (wave < PIECES) {
            const unsigned int f0 = (unsigned int)wave * 256u + opaque((unsigned int)lane) * 4u;
            if (f0 < 2u * C::MID) {
                const float* src = (f0 < (unsigned int)C::MID ? inv1 : p.b1 - C::MID) + (size_t)f0;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(w3buf + wave * 1024), 16, 0, 0);
            }
        }
    };
    auto stage_sync = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0x0f70);                  // vmcnt(0), as a builtin: see fused_tail_kernel
        __syncthreads();
    };
    auto load_identity = [&](const int co_first, float (&rres)[8]) __attribute__((always_inline)) {    // two 16-row tiles from co_first: rows 4 kb + r
#pragma unroll
        for (int i = 0; i < 8; ++i)
#if SS_FT_PROBE == 1
            rres[i] = 0.f;
#else
            rres[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.res + (int64_t)(co_first + 16 * (i >> 2) + (i & 3)) * V) + (size_t)lane_off);
#endif
    };
    const f16x8 k2048 = {(_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f),
                         (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f)};
    // two 16-row tiles per step, their MFMAs alternating: consecutive MFMAs never wait for each other's accumulator (same products, same order
    // per accumulator as mma3)
    auto mma3x2 = [&](f32x4& c0, f32x4& c1, const f16x8 a0_hi, const f16x8 a0_lo, const f16x8 a1_hi, const f16x8 a1_lo, const f16x8 b_hi, const f16x8 b_lo)
        __attribute__((always_inline)) {
        const f16x8 a0_his = a0_hi * k2048, a1_his = a1_hi * k2048;
#if SS_FT_PROBE == 3
        asm volatile("" ::"v"(a0_his), "v"(a0_lo), "v"(a1_his), "v"(a1_lo), "v"(b_hi), "v"(b_lo));
        return;
#endif
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0_lo, b_hi, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1_lo, b_hi, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0_his, b_lo, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1_his, b_lo, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0_hi, b_hi, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1_hi, b_hi, c1, 0, 0, 0);
    };

    // the first stage's weights, then the wave's whole conv3 input: chunk s, lane (n, kb) = octet 4 s + kb of both planes at its position
    dma_w3(0, 0, 0);
    f16x8 xh[C::NC3], xl[C::NC3];
    {
        const char* xs = reinterpret_cast<const char*>(p.x16) + (size_t)((unsigned int)pos_c * 16u);
#pragma unroll
        for (int s = 0; s < C::NC3; ++s) {
            xh[s] = *reinterpret_cast<const f16x8*>(xs + ((int64_t)(4 * s + kb)) * V * 16);
            xl[s] = *reinterpret_cast<const f16x8*>(xs + ((int64_t)(C::MID / 8 + 4 * s + kb)) * V * 16);
        }
    }
    f32x4 acc1[C::NT1];
#pragma unroll
    for (int m = 0; m < C::NT1; ++m) acc1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    stage_sync();
    int sb = 0, eb = 0;                                      // buffers of the conv3 stage / the conv1 k-step about to run
#pragma unroll 1
    for (int j = 0; j < C::NCT; ++j) {
        f32x4 acc3[C::NT3];
#pragma unroll
        for (int t = 0; t < C::NT3; ++t) acc3[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int co_j = j * C::CT;
        float rres[8];
#pragma unroll
        for (int st = 0; st < C::NS; ++st) {
            if (st + 1 < C::NS) dma_w3(j, st + 1, sb ^ 1);
            else { dma_w1(j * C::NU, eb); dma_tab(j); load_identity(co_j, rres); }
            const char* wb = w3buf + sb * C::W3_BYTES + (((kb >> 1) * 2 * 2 + (kb & 1)) * C::CT + n) * 16;       // row (g = kb / 2, plane 0, h = kb % 2), this lane's tile row
            constexpr int NSTEP = C::SC * C::NT3 / 2;            // steps of two tiles
            f16x8 a_hi[2][2], a_lo[2][2];                       // (two sets here: the fragments of step i + 1 are requested before the MFMAs of step i issue)
            auto ld_a = [&](const int i, const int k) __attribute__((always_inline)) {
                const int sc = (2 * i) / C::NT3, t = (2 * i) % C::NT3;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    a_hi[k][e] = *reinterpret_cast<const f16x8*>(wb + sc * C::W3_CHUNK + (0 * 2 * C::CT + (t + e) * 16) * 16);
                    a_lo[k][e] = *reinterpret_cast<const f16x8*>(wb + sc * C::W3_CHUNK + (1 * 2 * C::CT + (t + e) * 16) * 16);
                }
            };
            ld_a(0, 0);
#pragma unroll
            for (int i = 0; i < NSTEP; ++i) {
                if (i + 1 < NSTEP) ld_a(i + 1, (i + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                const int t = (2 * i) % C::NT3, ch = st * C::SC + (2 * i) / C::NT3;
                mma3x2(acc3[t], acc3[t + 1], a_hi[i & 1][0], a_lo[i & 1][0], a_hi[i & 1][1], a_lo[i & 1][1], xh[ch], xl[ch]);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage_sync();
            sb ^= 1;
        }
        const float* tab = reinterpret_cast<const float*>(tabbuf + (j & 1) * C::TAB_BYTES);
#pragma unroll
        for (int u = 0; u < C::NU; ++u) {
            // ---- tiles 2 u, 2 u + 1 of the co-tile: scale back, + bias, + identity, ReLU, store, split: conv1's B fragment of k-step u -----------
            unsigned int hw[4], lw[4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * u + tt, co4 = 16 * t + 4 * kb;                      // (within the co-tile)
                const float4 sc = *reinterpret_cast<const float4*>(tab + co4);
                const float4 bv = *reinterpret_cast<const float4*>(tab + C::CT + co4);
                float v0 = __fmul_rn(acc3[t][0], sc.x), v1 = __fmul_rn(acc3[t][1], sc.y), v2 = __fmul_rn(acc3[t][2], sc.z), v3 = __fmul_rn(acc3[t][3], sc.w);
                v0 = relu_keep_nan(__fadd_rn(__fadd_rn(v0, bv.x), rres[4 * tt + 0]));
                v1 = relu_keep_nan(__fadd_rn(__fadd_rn(v1, bv.y), rres[4 * tt + 1]));
                v2 = relu_keep_nan(__fadd_rn(__fadd_rn(v2, bv.z), rres[4 * tt + 2]));
                v3 = relu_keep_nan(__fadd_rn(__fadd_rn(v3, bv.w), rres[4 * tt + 3]));
                if (pos_ok && SS_FT_PROBE != 1) {
                    char* yo = reinterpret_cast<char*>(p.y + (int64_t)(co_j + 16 * t) * V);       // (uniform; rows 4 kb .. are in lane_off)
                    *reinterpret_cast<float*>(yo + (size_t)lane_off) = v0;
                    *reinterpret_cast<float*>(yo + V * 4 + (size_t)lane_off) = v1;
                    *reinterpret_cast<float*>(yo + V * 8 + (size_t)lane_off) = v2;
                    *reinterpret_cast<float*>(yo + V * 12 + (size_t)lane_off) = v3;
                }
                ft_split_pair(v0, v1, hw[2 * tt], lw[2 * tt]);
                ft_split_pair(v2, v3, hw[2 * tt + 1], lw[2 * tt + 1]);
            }
            // The lane (n, rb) holds rows 4 rb + 0..3 of both tiles; conv1's k slots of lane (n, kb) are channels 8 kb + 0..7 of the 32 (the packed
            // weights' order: one conflict-free 16-byte read per fragment).  Two half / row exchanges per word move them: after permlane32_swap the
            // row blocks hold (T0 rb0, T0 rb1, T1 rb0, T1 rb1) | (T0 rb2, T0 rb3, T1 rb2, T1 rb3), after permlane16_swap (T0 rb0, T0 rb2, T1 rb0, T1 rb2) |
            // (T0 rb1, T0 rb3, T1 rb1, T1 rb3): row block kb now has (8 kb + 0..3 | 8 kb + 4..7).
            unsigned int bw[2][4];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned int t0 = pl ? lw[i] : hw[i], t1 = pl ? lw[2 + i] : hw[2 + i];
                    auto s32 = __builtin_amdgcn_permlane32_swap(t0, t1, false, false);
                    auto s16 = __builtin_amdgcn_permlane16_swap(s32[0], s32[1], false, false);
                    bw[pl][i] = s16[0];
                    bw[pl][2 + i] = s16[1];
                }
            const u32x4 bh = {bw[0][0], bw[0][1], bw[0][2], bw[0][3]}, bl = {bw[1][0], bw[1][1], bw[1][2], bw[1][3]};
            const f16x8 b_hi = __builtin_bit_cast(f16x8, bh), b_lo = __builtin_bit_cast(f16x8, bl);
            __builtin_amdgcn_sched_barrier(0);
            if (u + 1 < C::NU) { dma_w1(j * C::NU + u + 1, eb ^ 1); load_identity(co_j + 32 * (u + 1), rres); }
            else if (j + 1 < C::NCT) dma_w3(j + 1, 0, sb);
            else dma_tail_tab();
            __builtin_amdgcn_sched_barrier(0);
            // ---- conv1, k-step u ------------------------------------------------------------------------------------------------------------
            const char* w1b = w1buf + eb * C::W1_BYTES + (((kb >> 1) * 2 * 2 + (kb & 1)) * C::MID + n) * 16;      // row (g = kb / 2, plane 0, h = kb % 2)
            f16x8 a_hi[1][2], a_lo[1][2];
            auto ld_a1 = [&](const int i, const int k) __attribute__((always_inline)) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    a_hi[k][e] = *reinterpret_cast<const f16x8*>(w1b + (0 * 2 * C::MID + (2 * i + e) * 16) * 16);
                    a_lo[k][e] = *reinterpret_cast<const f16x8*>(w1b + (1 * 2 * C::MID + (2 * i + e) * 16) * 16);
                }
            };
#pragma unroll
            for (int i = 0; i < C::NT1 / 2; ++i) {
                ld_a1(i, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma3x2(acc1[2 * i], acc1[2 * i + 1], a_hi[0][0], a_lo[0][0], a_hi[0][1], a_lo[0][1], b_hi, b_lo);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage_sync();
            eb ^= 1;
        }
    }
    // ---- conv1's epilogue ------------------------------------------------------------------------------------------------------------------
    if (pos_ok) {
        const int hwp = p.dec_H * p.dec_W;
        const int t2 = pos / hwp, r2 = pos - t2 * hwp, y2 = r2 / p.dec_W, x2 = r2 - y2 * p.dec_W;
        const unsigned int z_off = (unsigned int)(((int64_t)(4 * kb) * p.z_cs + (int64_t)t2 * p.z_ts + (int64_t)y2 * p.z_ys + x2) * 4);      // bytes
        const float* ttab = reinterpret_cast<const float*>(w3buf);
#pragma unroll
        for (int m1 = 0; m1 < C::NT1; ++m1) {
            const int co4 = m1 * 16 + 4 * kb;
            const float4 sc = *reinterpret_cast<const float4*>(ttab + co4);
            const float4 bv = *reinterpret_cast<const float4*>(ttab + C::MID + co4);
            char* zo = reinterpret_cast<char*>(p.z + (int64_t)(m1 * 16) * p.z_cs);     // (uniform)
            *reinterpret_cast<float*>(zo + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][0], sc.x), bv.x));
            *reinterpret_cast<float*>(zo + p.z_cs * 4 + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][1], sc.y), bv.y));
            *reinterpret_cast<float*>(zo + p.z_cs * 8 + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][2], sc.z), bv.z));
            *reinterpret_cast<float*>(zo + p.z_cs * 12 + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][3], sc.w), bv.w));
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The one-wave-per-SIMD form ("R1"; MID = 256).  The 16-column form above is latency-shaped: two 256-register waves per SIMD meet the same
// barrier 64 times, a stage is 0.7 us of matrix work behind LDS reads that one spare fragment set cannot cover, and every stage's vmcnt(0)
// also waits for the identity loads and output stores of that stage (PMC: waves parked 42 %, the MFMA pipe busy 29 % of the launch).  Here
// a workgroup is FOUR waves, one per SIMD, with the whole 512-register file each:
//   * a wave owns 32 positions and keeps, for all of them, conv3's whole input (MID / 16 k-groups x (hi, lo) = 128 registers, loaded once),
//     conv1's MID x 32 accumulators (128) and the co-tile's CT x 32 (64): an A fragment read from LDS feeds a 32-column MFMA, i.e. half the
//     LDS traffic per MFMA cycle of the 16-column form, and the k order per accumulator is the separate launches' (v_mfma_f32_32x32x16_f16,
//     two 16-deep k-groups per 32-channel chunk) => bit-identical to them again;
//   * the spare registers are a ring of A-fragment sets two steps ahead of the MFMAs, and a second identity buffer: the identity rows of a
//     32-channel tile are requested two stages before its epilogue;
//   * the end-of-stage wait is COUNTED: the stage's LDS-DMA pieces are issued first, so `s_waitcnt vmcnt(n)` with n = the identity loads
//     and output stores issued behind them waits for the weights only (vmcnt retires in issue order);
//   * the epilogue of tile u + 1 (scale / bias / identity / ReLU / store / split / lane exchange) is cut into slices that sit between the
//     MFMAs of conv1's k-step u; identity loads and output stores are buffer instructions (row base in an SGPR offset, one loop-invariant
//     lane offset; columns past V are dropped by the descriptor's range check): no address arithmetic, no branch inside a stage.
typedef __amdgpu_buffer_rsrc_t ft_rsrc_t;
#define FT_VMCNT_(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x0F70)
#if defined(SS_EXPERIMENTS) && defined(SS_FT_PROBE) && SS_FT_PROBE == 7      // (probe 7: the stage ends do not wait for their DMA)
#define FT_VMCNT(n) FT_VMCNT_(63)
#else
#define FT_VMCNT(n) FT_VMCNT_(n)
#endif
#define FT_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

template <int MID_, int CT_>
struct FusedTailR1Cfg {
    static constexpr int MID = MID_, CT = CT_, COUT = 4 * MID_;
    static constexpr int NW = 4, NTHREADS = 64 * NW, P = 32 * NW;
    static constexpr int NC3 = MID / 32, NG = MID / 16, NCT = COUT / CT;
    static constexpr int MI3 = CT / 32, MI1 = MID / 32;        // 32-row accumulator tiles: conv3 co-tile, conv1
    static constexpr int SC = 2;                              // conv3 chunks per stage
    static constexpr int NS = NC3 / SC, NU = CT / 32;         // conv3 stages / conv1 k-steps per co-tile
    static constexpr int NX3 = SC * 2 * (MI3 / 2), NX1 = 2 * (MI1 / 2);     // steps (two tiles x three products) per conv3 stage / conv1 k-step
    static constexpr int W3_CHUNK = 8 * CT * 16, W3_BYTES = SC * W3_CHUNK;
    static constexpr int W1_BYTES = 8 * MID * 16;
    static constexpr int TAB_BYTES = 1024;
    static constexpr int LDS_BYTES = 2 * W3_BYTES + 2 * W1_BYTES + 2 * TAB_BYTES;
    static constexpr int W3_PIECES = W3_BYTES / 1024 / NW, W1_PIECES = W1_BYTES / 1024 / NW;
    // a conv1 k-step of NX1 steps carries the next tile's epilogue: SL slices (rows 8 q4 ..) per step from step S0 on, the lane exchanges behind them;
    // the requests of the next stage go out in the steps before S0, M_PPS pieces per step (MID = 256: 8 steps -- requests 0-1, slices 2-5, exchanges
    // 6-7; MID = 128: 4 steps -- requests 0, slices 1-2, exchanges 3)
    static constexpr int SL = 8 / NX1, S0 = NX1 / 4, M_PPS = 8 / S0;
    static_assert(NC3 % SC == 0 && MI3 % 2 == 0 && MI1 % 2 == 0 && W3_BYTES % (1024 * NW) == 0 && W1_BYTES % (1024 * NW) == 0 && 2 * CT * 4 <= TAB_BYTES, "whole DMA pieces per wave");
    static_assert(LDS_BYTES <= 160 * 1024 && 2 * MID * 4 <= W3_BYTES && (NX1 == 8 || NX1 == 4) && NX3 == 8 && MI3 == 4 && NS >= 2 && NU >= 2 && NU % 2 == 0, "LDS / slice placement / identity buffer parity");
    static_assert(W3_PIECES == 8 && W1_PIECES <= 8, "the pieces of a stage are requested in its first steps");
    static_assert(NG * 8 + MI1 * 16 <= 256, "the conv3 input and conv1's accumulators are the AGPR half of the register file");
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS, 1) void fused_tail_r1_kernel(const FusedTailParams p) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(1024))) char smem[C::LDS_BYTES];
    char* const w3buf = smem;
    char* const w1buf = smem + 2 * C::W3_BYTES;
    char* const tabbuf = smem + 2 * C::W3_BYTES + 2 * C::W1_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int pos = blockIdx.x * C::P + wave * 32 + l31;       // this lane's position (column of every accumulator tile of the wave)
    const bool pos_ok = pos < p.V;
    const int pos_c = min(pos, p.V - 1);
    const int64_t V = p.V;
    const unsigned int V4 = (unsigned int)p.V * 4u;           // bytes per channel row
    // rows (r & 3) + 8 (r >> 2) + 4 half of a 32-channel tile: the 4 half part and the column are the lane offset; loads of a column past V read
    // the last valid one, its stores carry an offset the descriptor's range check drops
    const unsigned int lane_ld = (unsigned int)(4 * half) * V4 + (unsigned int)pos_c * 4u;
    const unsigned int lane_st = pos_ok ? lane_ld : 0xFFFFFF00u;
    const unsigned int tile_bytes = (unsigned int)C::CT * V4;  // a co-tile's rows: the extent of its descriptors
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float* inv3 = reinterpret_cast<const float*>(p.w3 + (int64_t)C::NC3 * 8 * C::COUT * 16);
    const float* inv1 = reinterpret_cast<const float*>(p.w1 + (int64_t)(C::COUT / 32) * 8 * C::MID * 16);

    // ---- LDS-DMA, as inline assembly ---------------------------------------------------------------------------------------------------
    // One piece = 1 KB per wave instruction (global_load_lds_dwordx4: wave-uniform 64-bit base in SGPRs + ONE loop-invariant lane offset, LDS
    // base in M0).  Assembly, not the builtin: the compiler cannot tell an LDS read from the target of a DMA it knows to be in flight, and
    // guards the first table read behind a request with a vmcnt(0); requests it does not see are waited for by the counted s_waitcnt at the
    // end of the stage and by nothing else.  (Its own counts for the identity loads stay safe: unseen requests only make a vmcnt(n) stricter.)
    const unsigned int lane16 = (unsigned int)lane * 16u;
    const unsigned int lds_w3 = (unsigned int)(uintptr_t)(lptr_t)w3buf, lds_w1 = (unsigned int)(uintptr_t)(lptr_t)w1buf;
    auto dma_piece = [&](const char* sbase, const unsigned int lds_addr) __attribute__((always_inline)) {
#if SS_FT_PROBE == 4
        return;
#endif
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(lane16), "s"(sbase) : "memory");
    };
    auto dma_w3 = [&](const int j, const int st, const int buf, const int k) __attribute__((always_inline)) {      // piece k of chunks SC st .. of co-tile j
        constexpr int PER_ROW = C::CT * 16 / 1024, PER_CHUNK = C::W3_CHUNK / 1024;
        const int idx = wave + k * C::NW, sc = idx / PER_CHUNK, r = idx % PER_CHUNK, row = r / PER_ROW, part = r % PER_ROW;
        dma_piece(p.w3 + ((int64_t)(st * C::SC + sc) * 8 + row) * (C::COUT * 16) + (int64_t)j * (C::CT * 16) + part * 1024, lds_w3 + buf * C::W3_BYTES + idx * 1024);
    };
    auto dma_w1 = [&](const int cc, const int buf, const int k) __attribute__((always_inline)) {                   // piece k of conv1 k-step (32-channel chunk) cc
        const int idx = wave + k * C::NW;
        dma_piece(p.w1 + (int64_t)cc * C::W1_BYTES + (int64_t)idx * 1024, lds_w1 + buf * C::W1_BYTES + idx * 1024);
    };
    // co-tile j's (1 / scale | bias) table: lanes 0 .. CT / 4 - 1 of wave 0 fetch the scales, the next CT / 4 the biases (an ordinary load at the
    // top of conv3's last stage), and store them at its end
    float4 tabv = {0.f, 0.f, 0.f, 0.f};
    auto tab_fetch = [&](const int j) __attribute__((always_inline)) {
        if (wave == 0 && lane < 2 * (C::CT / 4))
            tabv = *reinterpret_cast<const float4*>((lane < C::CT / 4 ? inv3 : p.b3 - C::CT) + (int64_t)j * C::CT + lane * 4);
    };
    auto tab_store = [&](const int j) __attribute__((always_inline)) {
        if (wave == 0 && lane < 2 * (C::CT / 4)) *reinterpret_cast<float4*>(tabbuf + (j & 1) * C::TAB_BYTES + lane * 16) = tabv;
    };
    auto rsrc_of = [&](const float* base, const int jj) __attribute__((always_inline)) {        // the CT rows of co-tile jj of a dense [4 MID][V] map
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (int64_t)jj * C::CT * V), 0, (int)tile_bytes, 0x00020000);
    };
    // identity rows of 32-channel tile u of co-tile jj, this lane's column
    auto load_identity = [&](const int jj, const int u, float (&rr)[16]) __attribute__((always_inline)) {
        const ft_rsrc_t rs = rsrc_of(p.res, jj);
#pragma unroll
        for (int r = 0; r < 16; ++r)
#if SS_FT_PROBE == 1 || SS_FT_PROBE == 6
            rr[r] = 0.f;
#else
            rr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane_ld, (unsigned int)(32 * u + (r & 3) + 8 * (r >> 2)) * V4, 0));
#endif
    };
    const f16x8 k2048 = {(_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f),
                         (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f)};
    // two tiles per step, their MFMAs alternating (never two in a row on one accumulator); per accumulator the three products in mma3's order
    auto mma6 = [&](f32x16& c0, f32x16& c1, const f16x8 (&a_hi)[2], const f16x8 (&a_lo)[2], const f16x8 (&a_his)[2], const f16x8 b_hi, const f16x8 b_lo)
        __attribute__((always_inline)) {
#if SS_FT_PROBE == 3
        asm volatile("" ::"v"(a_his[0]), "v"(a_lo[0]), "v"(a_his[1]), "v"(a_lo[1]), "v"(a_hi[0]), "v"(a_hi[1]), "v"(b_hi), "v"(b_lo));
        return;
#endif
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[0], b_hi, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[1], b_hi, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_his[0], b_lo, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_his[1], b_lo, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[0], b_hi, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[1], b_hi, c1, 0, 0, 0);
    };

    // ---- prologue: the first stage's weights, the first two tiles' identity rows, the wave's whole conv3 input --------------------------
#pragma unroll
    for (int k = 0; k < C::W3_PIECES; ++k) dma_w3(0, 0, 0, k);
    float rres[2][16];
    load_identity(0, 0, rres[0]);
    load_identity(0, 1, rres[1]);
    f16x8 xh[C::NG], xl[C::NG];                    // k-group G: lane (half, n) = octet 2 G + half of both planes at its position
    {
        const char* xs = reinterpret_cast<const char*>(p.x16) + (size_t)((unsigned int)pos_c * 16u);
#pragma unroll
        for (int G = 0; G < C::NG; ++G) {
            xh[G] = *reinterpret_cast<const f16x8*>(xs + ((int64_t)(2 * G + half)) * V * 16);
            xl[G] = *reinterpret_cast<const f16x8*>(xs + ((int64_t)(C::MID / 8 + 2 * G + half)) * V * 16);
        }
    }
    f32x16 acc1[C::MI1];
#pragma unroll
    for (int m = 0; m < C::MI1; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[m][r] = 0.f;
    FT_VMCNT(0);
    __syncthreads();
    int sb = 0, eb = 0;                                      // buffers of the conv3 stage / the conv1 k-step about to run
#pragma unroll 1
    for (int j = 0; j < C::NCT; ++j) {
        f32x16 acc3[C::MI3];
#pragma unroll
        for (int m = 0; m < C::MI3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[m][r] = 0.f;
        // ---- the epilogue's pieces (used from conv3's last stage on) -----------------------------------------------------------------------
        const float* tab = reinterpret_cast<const float*>(tabbuf + (j & 1) * C::TAB_BYTES) + 4 * half;
        const ft_rsrc_t ry = rsrc_of(p.y, j);
        unsigned int hw[8], lw[8];
        f16x8 b_hi[2][2], b_lo[2][2];                       // [k-step parity][k-group]: conv1's B fragments
        // rows 8 q4 + 4 half + 0 .. 3 of tile u: scale back, + bias, + identity, ReLU, store, split
        auto e_slice = [&](const int u, const int q4) __attribute__((always_inline)) {
            const int c4 = 32 * u + 8 * q4;
            const float4 sc = *reinterpret_cast<const float4*>(tab + c4);
            const float4 bv = *reinterpret_cast<const float4*>(tab + C::CT + c4);
            float (&rr)[16] = rres[u & 1];
            // (explicitly rounded steps: the standalone kernel scales its accumulators in one place and adds bias / identity in another)
            float v0 = __fmul_rn(acc3[u][4 * q4 + 0], sc.x), v1 = __fmul_rn(acc3[u][4 * q4 + 1], sc.y), v2 = __fmul_rn(acc3[u][4 * q4 + 2], sc.z),
                  v3 = __fmul_rn(acc3[u][4 * q4 + 3], sc.w);
            v0 = relu_keep_nan(__fadd_rn(__fadd_rn(v0, bv.x), rr[4 * q4 + 0]));
            v1 = relu_keep_nan(__fadd_rn(__fadd_rn(v1, bv.y), rr[4 * q4 + 1]));
            v2 = relu_keep_nan(__fadd_rn(__fadd_rn(v2, bv.z), rr[4 * q4 + 2]));
            v3 = relu_keep_nan(__fadd_rn(__fadd_rn(v3, bv.w), rr[4 * q4 + 3]));
#if SS_FT_PROBE != 1 && SS_FT_PROBE != 5
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v0), ry, lane_st, (unsigned int)(c4 + 0) * V4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v1), ry, lane_st, (unsigned int)(c4 + 1) * V4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v2), ry, lane_st, (unsigned int)(c4 + 2) * V4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v3), ry, lane_st, (unsigned int)(c4 + 3) * V4, 0);
#endif
            ft_split_pair(v0, v1, hw[2 * q4], lw[2 * q4]);
            ft_split_pair(v2, v3, hw[2 * q4 + 1], lw[2 * q4 + 1]);
        };
        // k-group gg of the tile as conv1's B fragment: the lane halves hold (q4 even: ch 0-3 | 4-7), (q4 odd: 8-11 | 12-15) of the group; swapping
        // the upper half of the q4-even words with the lower half of the q4-odd words gives (0-3, 4-7) to half 0 and (8-11, 12-15) to half 1
        auto e_perm = [&](const int gg, f16x8& bh, f16x8& bl) __attribute__((always_inline)) {
            unsigned int xh_[2], yh_[2], xl_[2], yl_[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                auto sh = __builtin_amdgcn_permlane32_swap(hw[4 * gg + i], hw[4 * gg + 2 + i], false, false);
                xh_[i] = sh[0]; yh_[i] = sh[1];
                auto sl = __builtin_amdgcn_permlane32_swap(lw[4 * gg + i], lw[4 * gg + 2 + i], false, false);
                xl_[i] = sl[0]; yl_[i] = sl[1];
            }
            const u32x4 bh4 = {xh_[0], xh_[1], yh_[0], yh_[1]}, bl4 = {xl_[0], xl_[1], yl_[0], yl_[1]};
            bh = __builtin_bit_cast(f16x8, bh4);
            bl = __builtin_bit_cast(f16x8, bl4);
        };
        // ---- conv3: NS stages of SC chunks ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int st = 0; st < C::NS; ++st) {
            if (st + 2 == C::NS) tab_fetch(j);
            const char* wb = w3buf + sb * C::W3_BYTES + (half * C::CT + l31) * 16;
            f16x8 a_hi[3][2], a_lo[3][2], a_his[2][2];
            auto ld3 = [&](const int s, const int k) __attribute__((always_inline)) {
                const int mp = s / (2 * C::SC), sc = (s % (2 * C::SC)) / 2, g = s % 2;      // tile pair outermost: tiles 0-1 are finished half a stage early
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    a_hi[k][e] = *reinterpret_cast<const f16x8*>(wb + sc * C::W3_CHUNK + (((g * 2 + 0) * 2) * C::CT + (2 * mp + e) * 32) * 16);
                    a_lo[k][e] = *reinterpret_cast<const f16x8*>(wb + sc * C::W3_CHUNK + (((g * 2 + 1) * 2) * C::CT + (2 * mp + e) * 32) * 16);
                }
            };
            ld3(0, 0);
            ld3(1, 1);
            a_his[0][0] = a_hi[0][0] * k2048;
            a_his[0][1] = a_hi[0][1] * k2048;
#pragma unroll
            for (int s = 0; s < C::NX3; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s < 2) {                                 // what the next stage reads, four pieces in each of the first two steps
#pragma unroll
                    for (int k = 4 * s; k < 4 * s + 4; ++k) {
                        if (st + 1 < C::NS) dma_w3(j, st + 1, sb ^ 1, k);
                        else if (k < C::W1_PIECES) dma_w1(j * C::NU, eb, k);
                    }
                }
                if (s + 2 < C::NX3) ld3(s + 2, (s + 2) % 3);
                if (s + 1 < C::NX3) {
                    a_his[(s + 1) & 1][0] = a_hi[(s + 1) % 3][0] * k2048;
                    a_his[(s + 1) & 1][1] = a_hi[(s + 1) % 3][1] * k2048;
                }
                // the co-tile's LAST stage: tile 0 got its last products in step NX3 / 2 - 1 -- its epilogue, a slice per step, sits between the MFMAs of
                // the other tile pair (nothing of it is left in front of conv1's first k-step but the two lane exchanges)
                if (st + 1 == C::NS && s >= C::NX3 / 2) e_slice(0, s - C::NX3 / 2);
                const int mp = s / (2 * C::SC), sc = (s % (2 * C::SC)) / 2, g = s % 2, G = 2 * (st * C::SC + sc) + g;
                mma6(acc3[2 * mp], acc3[2 * mp + 1], a_hi[s % 3], a_lo[s % 3], a_his[s & 1], xh[G], xl[G]);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    FT_SGB(0x008, 1); FT_SGB(0x100, 1);
                    if (st + 1 == C::NS && s >= C::NX3 / 2) { FT_SGB(0x002, 7); FT_SGB(0x040, 1); }
                    else FT_SGB(0x002, 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 == C::NS) FT_VMCNT(16);               // (tile 0's 16 output stores stay in flight; otherwise nothing was issued behind the requests)
            else FT_VMCNT(0);
            if (st + 2 == C::NS) tab_store(j);
            __syncthreads();
            sb ^= 1;
        }
        // ---- conv1's k-steps, each carrying the next tile's epilogue ---------------------------------------------------------------------------
        e_perm(0, b_hi[0][0], b_lo[0][0]);
        e_perm(1, b_hi[0][1], b_lo[0][1]);
#pragma unroll
        for (int u = 0; u < C::NU; ++u) {
            const char* w1b = w1buf + eb * C::W1_BYTES + (half * C::MID + l31) * 16;
            f16x8 a_hi[3][2], a_lo[3][2], a_his[2][2];
            auto ld1 = [&](const int s, const int k) __attribute__((always_inline)) {
                const int g = s / (C::MI1 / 2), mp = s % (C::MI1 / 2);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    a_hi[k][e] = *reinterpret_cast<const f16x8*>(w1b + (((g * 2 + 0) * 2) * C::MID + (2 * mp + e) * 32) * 16);
                    a_lo[k][e] = *reinterpret_cast<const f16x8*>(w1b + (((g * 2 + 1) * 2) * C::MID + (2 * mp + e) * 32) * 16);
                }
            };
            ld1(0, 0);
            ld1(1, 1);
            a_his[0][0] = a_hi[0][0] * k2048;
            a_his[0][1] = a_hi[0][1] * k2048;
#pragma unroll
            for (int s = 0; s < C::NX1; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                // Steps before S0: what the next stage reads.  Step S0: the identity rows of tile u + 2 (tile u's buffer: its epilogue ran a stage ago) --
                // a stage ahead of the slices that add them, and BEHIND this stage's requests, so that the counted wait leaves them in flight.
                // From step S0: the next tile's epilogue, SL slices (rows 8 q4 ..) per step; behind them its two lane exchanges.
                if (s < C::S0) {
#pragma unroll
                    for (int k = C::M_PPS * s; k < C::M_PPS * s + C::M_PPS; ++k) {
                        if (u + 1 < C::NU) { if (k < C::W1_PIECES) dma_w1(j * C::NU + u + 1, eb ^ 1, k); }
                        else if (j + 1 < C::NCT) dma_w3(j + 1, 0, sb, k);
                    }
                }
                if (s == C::S0) load_identity(u + 2 < C::NU ? j : min(j + 1, C::NCT - 1), (u + 2) % C::NU, rres[u & 1]);      // (past the last tile: the last co-tile's rows again, never used)
                if (s + 2 < C::NX1) ld1(s + 2, (s + 2) % 3);
                if (s + 1 < C::NX1) {
                    a_his[(s + 1) & 1][0] = a_hi[(s + 1) % 3][0] * k2048;
                    a_his[(s + 1) & 1][1] = a_hi[(s + 1) % 3][1] * k2048;
                }
                if (u + 1 < C::NU) {
                    constexpr int SE = C::S0 + 4 / C::SL;     // first step behind the slices
                    if (s >= C::S0 && s < SE) {
#pragma unroll
                        for (int q = 0; q < C::SL; ++q) e_slice(u + 1, (s - C::S0) * C::SL + q);
                    }
                    if (C::SL == 1 ? s == SE : s == SE) e_perm(0, b_hi[(u + 1) & 1][0], b_lo[(u + 1) & 1][0]);
                    if (C::SL == 1 ? s == SE + 1 : s == SE) e_perm(1, b_hi[(u + 1) & 1][1], b_lo[(u + 1) & 1][1]);
                }
                const int g = s / (C::MI1 / 2), mp = s % (C::MI1 / 2);
                mma6(acc1[2 * mp], acc1[2 * mp + 1], a_hi[s % 3], a_lo[s % 3], a_his[s & 1], b_hi[u & 1][g], b_lo[u & 1][g]);
#pragma unroll
                for (int i = 0; i < 6; ++i) { FT_SGB(0x008, 1); FT_SGB(0x100, 1); FT_SGB(0x002, 7 * C::SL); FT_SGB(0x040, C::SL); FT_SGB(0x020, 3); }
            }
            __builtin_amdgcn_sched_barrier(0);
            // behind the requests of steps 0-1 this stage issued 16 identity loads and, if it carried an epilogue, 16 output stores: those stay in flight
            if (u + 1 < C::NU) FT_VMCNT(32);
            else FT_VMCNT(16);
            __syncthreads();
            eb ^= 1;
        }
    }
    // ---- conv1's epilogue: scale back, + bias, ReLU, flat position -> (t, y, x) of the zero-haloed consumer layout -------------------------
    // its (1 / scale | bias) table: waves 0 / 1 put the MID scales / biases into the (idle) w3 buffer
    static_assert(C::MID / 4 <= 64, "one lane per four channels");
    if (wave < 2 && lane < C::MID / 4)
        *reinterpret_cast<float4*>(w3buf + wave * (C::MID * 4) + lane * 16) = *reinterpret_cast<const float4*>((wave == 0 ? inv1 : p.b1) + lane * 4);
    __syncthreads();
    if (pos_ok) {
        const int hwp = p.dec_H * p.dec_W;
        const int t2 = pos / hwp, r2 = pos - t2 * hwp, y2 = r2 / p.dec_W, x2 = r2 - y2 * p.dec_W;
        const unsigned int z_off = (unsigned int)(((int64_t)(4 * half) * p.z_cs + (int64_t)t2 * p.z_ts + (int64_t)y2 * p.z_ys + x2) * 4);      // bytes
        const float* ttab = reinterpret_cast<const float*>(w3buf);
#pragma unroll
        for (int m1 = 0; m1 < C::MI1; ++m1)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co4 = m1 * 32 + 8 * q + 4 * half;
                const float4 sc = *reinterpret_cast<const float4*>(ttab + co4);
                const float4 bv = *reinterpret_cast<const float4*>(ttab + C::MID + co4);
                char* zo = reinterpret_cast<char*>(p.z + (int64_t)(m1 * 32 + 8 * q) * p.z_cs);     // (uniform)
                *reinterpret_cast<float*>(zo + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][4 * q + 0], sc.x), bv.x));
                *reinterpret_cast<float*>(zo + p.z_cs * 4 + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][4 * q + 1], sc.y), bv.y));
                *reinterpret_cast<float*>(zo + p.z_cs * 8 + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][4 * q + 2], sc.z), bv.z));
                *reinterpret_cast<float*>(zo + p.z_cs * 12 + (size_t)z_off) = relu_keep_nan(__fadd_rn(__fmul_rn(acc1[m1][4 * q + 3], sc.w), bv.w));
            }
    }
}

template <class C>
static int launch_fused_r1_cfg(const FusedTailParams& p, hipStream_t s) {
    const double flops = 2.0 * 2.0 * (double)C::MID * C::COUT * (double)p.V;
    void* ev = profile_begin(19, flops, s);
    hipLaunchKernelGGL(fused_tail_r1_kernel<C>, dim3((unsigned)ceil_div(p.V, C::P)), dim3(C::NTHREADS), 0, s, p);
    profile_end(ev, s);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

template <class C>
static int launch_fused16_cfg(const FusedTailParams& p, hipStream_t s) {
    const double flops = 2.0 * 2.0 * (double)C::MID * C::COUT * (double)p.V;
    void* ev = profile_begin(19, flops, s);
    hipLaunchKernelGGL(fused_tail16_kernel<C>, dim3((unsigned)ceil_div(p.V, C::P)), dim3(C::NTHREADS), 0, s, p);
    profile_end(ev, s);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

template <class C>
static int launch_fused_cfg(const FusedTailParams& p, hipStream_t s) {
    const double flops = 2.0 * 2.0 * (double)C::MID * C::COUT * (double)p.V;        // both GEMMs
    void* ev = profile_begin(19, flops, s);
    hipLaunchKernelGGL(fused_tail_kernel<C>, dim3((unsigned)ceil_div(p.V, C::P)), dim3(C::NTHREADS), 0, s, p);
    profile_end(ev, s);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

// (the kernel reads the two-plane weight packing, SS_F16_WPLANES 2: the three-plane A/B build keeps the separate launches)
#if defined(SS_F16_WPLANES) && SS_F16_WPLANES != 2
bool fused_tail_supported(int) { return false; }
#else
bool fused_tail_supported(int mid) { return mid == 64 || mid == 128 || mid == 256; }
#endif

int launch_fused_tail(int mid, const unsigned int* x16, const float* w3, const float* b3, const float* res, float* y, const float* w1, const float* b1,
                      const StemsegVolume& z, int dec_H, int dec_W, int64_t V, int form, hipStream_t s) {
    SS_CHECK_ARG(x16 && w3 && b3 && res && y && w1 && b1 && z.ptr, "fused_tail: null pointer");
    SS_CHECK_ARG(fused_tail_supported(mid), "fused_tail: mid = %d (64, 128 or 256)", mid);
    SS_CHECK_ARG(V > 0 && V <= (1ll << 27) && z.c_stride <= (1ll << 27) && dec_H > 0 && dec_W > 0 && V % ((int64_t)dec_H * dec_W) == 0, "fused_tail: V = %lld positions of %d x %d planes",
                 (long long)V, dec_H, dec_W);
    SS_CHECK_ARG(z.C == mid, "fused_tail: consumer volume has %d channels, conv1 makes %d", z.C, mid);
    SS_CHECK_ARG(reinterpret_cast<uintptr_t>(x16) % 16 == 0 && reinterpret_cast<uintptr_t>(w3) % 16 == 0 && reinterpret_cast<uintptr_t>(w1) % 16 == 0 &&
                 reinterpret_cast<uintptr_t>(b3) % 16 == 0 && reinterpret_cast<uintptr_t>(b1) % 16 == 0, "fused_tail: 16-byte aligned operands");
    FusedTailParams p;
    p.x16 = x16; p.w3 = reinterpret_cast<const char*>(w3); p.b3 = b3; p.res = res; p.y = y;
    p.w1 = reinterpret_cast<const char*>(w1); p.b1 = b1;
    p.z = z.ptr; p.z_cs = z.c_stride; p.z_ts = z.t_stride; p.z_ys = z.y_stride; p.dec_H = dec_H; p.dec_W = dec_W;
    p.V = (int)V;
#if defined(SS_EXPERIMENTS) && defined(SS_FT_P256)
    if (mid == 256) return launch_fused_cfg<FusedTailCfg<256, 64, 256>>(p, s);
    if (mid == 128) return launch_fused_cfg<FusedTailCfg<128, 64, 256>>(p, s);
    return launch_fused_cfg<FusedTailCfg<64, 64, 256>>(p, s);
#else
#if defined(SS_EXPERIMENTS) && defined(SS_FT_W32)
    if (mid == 256) return launch_fused_cfg<FusedTailCfg<256, 64, 128>>(p, s);
#else
    if (mid == 256) {
        // form: 0 = the library's choice (the one-wave-per-SIMD form wherever its co-tile descriptors -- 128 V 4 bytes -- fit), 1 = the 16-column form, 2 = the
        // one-wave-per-SIMD form
        const bool r1_fits = (int64_t)128 * V * 4 < (1ll << 32) - (1 << 20);
        if (form != 1 && r1_fits) return launch_fused_r1_cfg<FusedTailR1Cfg<256, 128>>(p, s);
        return launch_fused16_cfg<FusedTail16Cfg<256>>(p, s);
    }
#endif
#if defined(SS_EXPERIMENTS) && defined(SS_FT_W16_ALL)
    if (mid == 128) return launch_fused16_cfg<FusedTail16Cfg<128>>(p, s);
    return launch_fused16_cfg<FusedTail16Cfg<64>>(p, s);
#endif
#ifndef SS_FT_CT12
#define SS_FT_CT12 128    // co-tile of the 32-column form (stages 1-2): 128 channels halve the re-reads of the input tile (229 / 196 registers, 80 / 72 KB: still two workgroups per CU); stage 2 335 -> 303 us, stage 1 510 -> 495
#endif
    if (mid == 128 && form != 1 && (int64_t)128 * V * 4 < (1ll << 32) - (1 << 20)) return launch_fused_r1_cfg<FusedTailR1Cfg<128, 128>>(p, s);
    if (mid == 128) return launch_fused_cfg<FusedTailCfg<128, SS_FT_CT12, 128>>(p, s);
    return launch_fused_cfg<FusedTailCfg<64, SS_FT_CT12, 128>>(p, s);
#endif
}

}  // namespace stemseg

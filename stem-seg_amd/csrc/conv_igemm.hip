// Implicit-GEMM 3-D convolution on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Replaces nn.Conv3d(k=3, s=1, p=1) of /root/reference/stemseg/modeling/embedding_decoder.py:21-58 and
// the 1x1x1 fuse convs of :68-80 (cuDNN / MKL-DNN calls in the reference).
//
// GEMM view:  D[co][voxel] = sum_{ci,tap} W[co][ci,tap] * X[ci][voxel + off(tap)]
//   M = Cout, N = T*H*W voxels, K = Cin * taps.
// One workgroup (4 waves) owns MT output channels x (ROWS rows x COLS*32 columns) voxels of one t-plane.
// Per chunk of CK input channels it stages into LDS
//   * the input halo tile  [CK][KT][ROWS+KH-1][XP]   -- rows are W-contiguous in HBM, loaded as 16-B
//     pieces (the zero-haloed source layout makes every row start 16-B aligned and removes all
//     boundary predicates from the inner loop), and
//   * the weight slab      [CK/4][taps][4][MT]       -- a plain linear copy of the packed layout,
// then every tap is a *shifted read* of the same LDS tile: the B fragment of lane l for tap (dt,dy,dx)
// is in_lds[c0 + (l>>5)][dt][row+dy][col0 + (l&31) + dx] -- 32 consecutive floats per half-wave, i.e.
// conflict-free ds_read_b32 with a compile-time immediate offset, zero VALU in the loop.
// The A fragment is w_lds[k][co0 + (l&31)], likewise conflict-free.
// fp32 MFMA issues one instruction per 64 cycles per SIMD, so 6 LDS reads per 8 MFMAs (MI=4, NI=2) leave
// the LDS pipe < 15 % busy; two workgroups per CU (<= 80 KB LDS, <= 256 VGPRs each) overlap one group's
// staging with the other's MFMA stream.
//
// The same template also serves the encoder's 2-D convolutions (KT = 1; the frames of a clip are the T axis) and every
// 1x1 convolution (one flat row of voxels).  Variants, all selected per launch by launch_conv3d():
//   PIPE  next chunk prefetched into registers under the current chunk's MFMA stream
//   GL    (with DB: two LDS buffers) the next chunk goes global -> LDS directly (global_load_lds_dwordx4), no staging registers:
//         the big 3x3x3 tile and every 1x1 tile of the fp32-input mode
//   FLAT  N tile = a run of the zero-haloed plane instead of rows x 32 columns  (maps whose width wastes a 32-column tile)
//   BF_   2 = bf16x6, 3 = f16x3: operands split once when a chunk is staged, products on v_mfma_f32_32x32x16_{bf16,f16}
// plus split-K with a deterministic slab reduce, an XCD-aware (and, for 3-D taps, t-fastest) tile order, a launch planner
// (fp32-input mode) that cuts big launches into whole rows + split-K rows so that their workgroups fill whole rounds of the
// chip, and the GroupNorm statistics of the output (decoder stages) as per-tile fp64 partial sums left by the epilogue.
// Every launch decision (tile shape, split-K factor) is a function of the layer's PLANNING shape and the precision only: the
// encoder passes its per-frame shape with a fixed planning frame count, so the K-partition -- and with it every output bit --
// does not depend on how many frames share a launch (ConvEpilogue::plan_frames).
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

// f16x3: activations are split as fp16 terms of x * 2^-2 (see split3)
#define STEMSEG_F16X3_ACT_SCALE 0.25f

#ifndef SS_F16_WPLANES
#define SS_F16_WPLANES 2           // f16x3: staged weight planes.  2: hi, lo -- the third A operand, hi_w * 2^-11 (it meets the input tile's lo * 2^11
                                  // plane), is made in registers, four packed multiplies per k-group step in the shadow of the MFMAs: a third less
                                  // slab, LDS traffic and staging.  3: the operand is a third packed plane (every MFMA operand straight from LDS).
                                  // Round 3 shipped 3 because the run-to-run differences under several streams were blamed on the in-register
                                  // multiply; round 4 traced every one of them to the VALU stem kernel -- with that gone both
                                  // forms are bit-stable in the three-lane soak, and numerically identical to each other.
#endif
// A/B and probe switches of the split-staged chunk loop exist in EXPERIMENT builds only (-DSS_EXPERIMENTS, STEMSEG_BUILD_DEFINES): a product
// build pins them to the shipped values below and refuses any of them on its command line -- it is never one -D away from a kernel that is
// wrong by construction (SS_PROBE) or from an un-soaked schedule.
#ifndef SS_EXPERIMENTS
#if defined(SS_PROBE) || defined(SS_X6_WMODE_SMALLG) || defined(SS_X6_SPREAD) || defined(SS_X6_SPREAD_DIV)
#error "SS_PROBE / SS_X6_WMODE_SMALLG / SS_X6_SPREAD / SS_X6_SPREAD_DIV are experiment switches: add -DSS_EXPERIMENTS"
#endif
#define SS_X6_WMODE_SMALLG -1
#define SS_X6_SPREAD 1
#define SS_X6_SPREAD_DIV 1
#define SS_PROBE 0
#else
#ifndef SS_X6_WMODE_SMALLG
#define SS_X6_WMODE_SMALLG -1      // weight staging of the split-staged tiles with <= 2 k-groups per chunk (1x1 taps), see ConvCfg::WMODE.  -1: f16x3 takes
                                  // mode 1 (one phase, the whole next chunk in registers under the chunk's MFMA stream), bf16x6 mode 2 where its second
                                  // register set fits (mode 1 costs bf16x6 1.2 %, measured); 0 / 1 / 2 force a mode for both (A/B builds)
#endif
#ifndef SS_X6_SPREAD
#define SS_X6_SPREAD 1             // 1: (f16x3) the next chunk's global loads go out a few per (k-group, mi) step instead of at the top of the phase;
                                  // 2: for bf16x6 as well (no gain there); 0: off.  Round 3 measured this 1-3 % faster per class and kept it off: the loads'
                                  // address arithmetic is VALU inside the MFMA stream, which that round blamed for the run-to-run differences under
                                  // several streams.  Round 4 traced those to the VALU stem kernel; with this and WMODE 1 on,
                                  // the three-lane soak stays at 0 differing lane-rounds and the step is 1.0 % faster (A/B on one box, interleaved).
#endif
#ifndef SS_X6_SPREAD_DIV
#define SS_X6_SPREAD_DIV 1          // the spread loads go out over the first 1 / DIV of a phase's (k-group, mi) steps.  A/B on the step, interleaved on one box:
                                  // round 4 (8-channel 1x3x3 chunks) 1 vs 2: +0.4 %, 3: -0.5 %, 4: -1.0 %, and 2 shipped; round 5 (16-channel chunks, nine
                                  // k-groups per phase pair) 1 vs 2: +1.2 / +0.7 % in two interleaved pairs, 3: -0.6 / -0.3 %: over the whole phase now
#endif

#ifndef SS_PROBE
#define SS_PROBE 0                 // timing probes of the split-staged chunk loop (experiment builds; results are WRONG for any value but 0):
                                  // 1 no global loads of the next chunk, 2 loads waited for but neither split nor written to LDS, 3 no MFMAs
                                  // (fragments still read), 4 no LDS fragment reads (MFMAs on stale registers), 5 no barriers
#endif

#endif  // SS_EXPERIMENTS

#if SS_PROBE == 5
#define SS_CHUNK_SYNC() do { } while (0)
#else
#define SS_CHUNK_SYNC() __syncthreads()
#endif

namespace stemseg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// small direct-to-LDS tiles (<= 4 accumulator blocks per wave, LDS <= 35 KB): workgroups per CU the register allocator leaves
// room for (4 -> 128 VGPRs)
#ifndef STEMSEG_MIN_WG4
#define STEMSEG_MIN_WG4 4
#endif
constexpr int MIN_WG4 = STEMSEG_MIN_WG4;

struct ConvKParams {
    const float* in;
    int64_t in_cs, in_ts, in_ys, in_limit;
    int in_H;
    const float* wpk;
    const float* bias;
    float* out;
    int64_t out_cs, out_ts, out_ys;
    int Cin, Cout, T, H, W;
    int tiles_x, tiles_y;
    int vec4;
    int chunks_per_split;        // split-K: blockIdx.z handles channel chunks [z*cps, (z+1)*cps)
    int64_t out_split_stride;    // floats between the partial-sum slabs of consecutive splits
    // fused epilogue (encoder): + residual, ReLU; optional decode of a flat voxel index into (t, y, x)
    int relu;
    const float* res;
    int64_t res_cs, res_ts, res_ys;
    int dec_H, dec_W;
    int vec_epi;                 // 16-B epilogue through an LDS transpose (dense, aligned outputs only)
    int t_fastest;               // tile order: t-planes of one (x, y) tile are neighbours in launch order (3-D taps)
    int flat_t;                  // FLAT tiles of a 2-D conv: the run of flat positions crosses the frames ([T][H+2][pitch] is one run; p.T in the grid = 1)
    int T_all;                   // flat_t: the frames of the volume (p.T is then 1)
    int n_co;                    // output-channel tiles (Cout / MT), the fastest-running part of the workgroup index
    // GroupNorm statistics of the output, taken in the epilogue (decoder stages): per (group, slot) partial sum / sum of squares
    // in fp64, one slot per tile (or per reduce block under split-K), combined in fixed order by gn_finalize_slots_kernel
    double* gn_part;             // [Cout / gn_cpg][gn_cap][2] or NULL
    int gn_cpg, gn_cap, gn_slot0;
    int* gn_used_host;           // host-side slot counter of the current conv (never dereferenced on the device)
    // clip batch (decoder stages): blockIdx.y = clip; the clips' volumes / partial tables lie at fixed strides from clip 0's.  Every
    // launch decision is taken on ONE clip's shape, so a clip's result does not depend on how many share the launch.
    int nb;
    int64_t in_bs, out_bs, gn_bs;   // floats, floats, doubles
    // f16x3: the output as fp16 PAIR PLANES instead of fp32 -- the operand form the fused bottleneck tail (bottleneck_fused.hip) stages by
    // LDS-DMA: word [plane hi | lo * 2^11][channel / 8][position][(channel % 8) / 2] = the split of split_pair_f16 applied by the PRODUCER,
    // same 4 bytes per value.  Dense outputs, by-element epilogue, no split-K (the launcher falls back to fp32 and says so through
    // gn_used_host, which then points at the caller's "done" flag).
    unsigned int* out_p16;
};

// FLAT (PMAX > 0): the N tile is a run of NSEG * 32 consecutive positions of the zero-haloed PLANE (row pitch <= PMAX floats)
// instead of ROWS x 32 columns: a tap is the flat offset dy * pitch + dx, so maps whose width is not a multiple of 32
// (W = 54: 64 columns computed for 54; flat: 56 for 54) lose almost nothing to tile quantisation.  Junk positions (halo
// columns) are computed and not stored.  Scalar epilogue.
// GL (with DB): the next chunk goes global -> LDS directly (global_load_lds_dwordx4, 1 KB per wave instruction, LDS image
// lane-linear = exactly the [piece] order of the staging loops), issued before the chunk's MFMA stream and retired by the
// vmcnt(0) in front of the chunk's one barrier: no staging registers (-40 VGPRs on the big tile), no ds_write pass between
// the MFMA stream and the barrier.  Out-of-range pieces (tile columns past the row end, the run before the plane's first
// row) are fetched from a clamped in-bounds address instead of being zero-filled: they only feed positions that are never
// stored.  Needs Cin % CK == 0 and Cout % MT == 0 (the launcher falls back to the register-staged twin otherwise).
// bf16x6 ("X6", BF_ = 2): every fp32 operand is split EXACTLY into three bf16 terms, x = hi + mid + lo (8 + 8 + 8 significand
// bits; both remainders are exact fp32 subtractions), and a*b is evaluated as the six products of weight >= 2^-16,
//   lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi   (smallest first),
// on v_mfma_f32_32x32x16_bf16 -- products exact, fp32 accumulate.  The three dropped products (mid*lo, lo*mid, lo*lo) are
// <= 2^-23 |a*b|, i.e. below the rounding of the fp32 accumulation itself: the result carries fp32-level error (measured
// against an fp64 convolution in tests/test_gpu_parity.py next to the fp32-MFMA path) at 16/6 = 2.7x the fp32-MFMA rate.
// The split happens ONCE, when a chunk is staged: LDS holds three bf16 planes of the input tile with the
// channels interleaved in PAIRS (one 32-bit word = the same position of channels 2p and 2p+1), so a lane's 8 k-values of a
// k-group are four ds_read_b32 per plane and the k-loop is ds_read + MFMA only.  Weights arrive pre-split; their slab is
// staged in two k-group phases that ping-pong with the MFMA stream (phase A's slots are refilled for the next chunk while
// phase B computes and vice versa, through registers), the input tile is prefetched into registers and written at the chunk end.
// BLK_ (split-staged tiles): the 32 positions of an MFMA column block are 4 ROWS x 8 COLUMNS of the map instead of 32 columns of one row, and a
// tile is TBR = BLK_ block rows x COLS_ block columns of them: 20 rows x 24 columns = 15 blocks for BLK_ = 5, COLS_ = 3.  120 x 216 maps (the
// 4x level of a 480 x 864 frame: block_4x of both decoders, the FPN output conv, layer 1's 3x3 convs) are then tiled with NO junk position --
// 16 x 32 tiles compute 128 x 224 -- and on a power-bound kernel MFMAs not issued are time.  The tile's LDS pitch is
// == 8 (mod 32) words so that the four rows of a block fall on disjoint banks; a tap is still one compile-time immediate on a per-lane base.
// The workgroup's 16th column block does not exist: the wave that would own it runs the chunk loop with one block (ni1_live below).
template <int KT_, int KH_, int KW_, int CK_, int MI_, int NI_, int WM_, int WN_, int COLS_, bool PIPE_ = false, int BF_ = 0, bool DB_ = false,
          int PMAX_ = 0, bool GL_ = false, int BLK_ = 0>
struct ConvCfg {
    // X6 = the split-staged path (operands split once, when a chunk is staged): BF_ 2 = bf16x6 (three bf16 planes, six products),
    // BF_ 3 = f16x3 (two fp16 planes of the SCALED operand, three products)
    static constexpr bool PIPE = PIPE_, BF = BF_ != 0, X6 = BF_ >= 2, F16 = BF_ == 3, DB = DB_, FLAT = PMAX_ > 0, GL = GL_;
    static_assert(BF_ == 0 || BF_ == 2 || BF_ == 3, "precision: 0 fp32-input MFMA, 2 bf16x6, 3 f16x3");
    static constexpr int NPL = BF_ == 2 ? 3 : (BF_ == 3 ? SS_F16_WPLANES : 2);   // 16-bit planes of the staged weights (f16x3: hi, lo [, hi * 2^-11])
    static constexpr int NPA = 3;                                       // A operands of a k-group step
    static constexpr int NPX = BF_ == 2 ? 3 : 2;                        // 16-bit planes of the staged input tile (f16x3: hi, lo * 2^11)
    static constexpr int NPROD = BF_ == 2 ? 6 : 3;                      // MFMAs per (A fragment, B fragment) pair
    static constexpr int PMAX = PMAX_;
    static constexpr int KT = KT_, KH = KH_, KW = KW_, CK = CK_, MI = MI_, NI = NI_, WM = WM_, WN = WN_, COLS = COLS_;
    static constexpr int TAPS = KT * KH * KW;
    static constexpr int NTHREADS = 64 * WM * WN;
    static constexpr int MT = WM * MI * 32;
    static constexpr int NSEG = WN * NI;
    static constexpr bool BLK = BLK_ > 0;
    static constexpr int NLIVE = BLK ? BLK_ * COLS_ : NSEG;              // column blocks that exist (BLK: block rows x block columns <= NSEG)
    static constexpr int ROWS = BLK ? 4 * BLK_ : NSEG / COLS;
    static constexpr int TW = BLK ? 8 * COLS_ : COLS_ * 32;              // tile width in columns
    static constexpr int RH = ROWS + KH - 1;
    static constexpr int XL = (TW + KW - 1 + 3) / 4;                     // 16-B pieces of a tile row that are staged
    static constexpr int XP = BLK ? ((4 * XL + 23) / 32) * 32 + 8 : 4 * XL;   // LDS row pitch (BLK: the next value == 8 mod 32)
    static constexpr int NT = NSEG * 32;                                // FLAT: voxels (flat plane positions) per tile
    static constexpr int FL = NT + 2 * PMAX + 8;                        // FLAT: staged run per (channel, dt): tile + one row and 4 either side
    static constexpr int IN_CH_STRIDE = FLAT ? KT * FL : KT * RH * XP;
    static constexpr int IN_PAIR_STRIDE = PMAX_ > 0 ? KT * (WN_ * NI_ * 32 + 2 * PMAX_ + 8) : KT * RH * XP;   // X6: words of one channel pair of one plane (flat: KT runs of FL)
    static constexpr int IN_PLANE_STRIDE = (CK / 2) * IN_PAIR_STRIDE;   // X6: words of one bf16 plane
    static constexpr int IN_FLOATS = X6 ? NPX * IN_PLANE_STRIDE : CK * IN_CH_STRIDE;
    // k-groups of 16 (split-staged modes): taps per group / channels per lane-half, by kernel class (27 taps: 4 x 2, 9 taps: 2 x 4, 1 tap: 1 x 8)
    static constexpr int TPG = TAPS >= 27 ? 4 : ((TAPS >= 9 && CK_ < 16) ? 2 : 1);   // (9 taps in 16-channel chunks: one tap x 16 channels per group -- nine groups, no padded tap slot)
    static constexpr int CPH = 8 / TPG;
    static constexpr int NTG = (TAPS + TPG - 1) / TPG;                  // tap groups
    static constexpr int NCG = BF ? CK / (2 * CPH) : 1;                 // channel groups per chunk
    static constexpr int G = NTG * NCG;                                 // 16-wide k-groups per chunk
    static constexpr int W_FLOATS = BF ? NPL * G * 2 * MT * 4 : CK * TAPS * MT;   // split-staged: [G][plane][half][MT] x 16 B
    static constexpr int GA = (G + 1) / 2;                              // X6: k-groups of weight phase A (phase B: the rest)
    // X6 weight staging of tiles with few k-groups per chunk (1x1 taps; a phase's MFMA stream is shorter than a global load):
    // 0 two phases, registers refilled per phase; 1 one phase, whole slab in registers; 2 two phases, one chunk of lookahead
    static constexpr int WMODE = (BF_ >= 2 && G <= 2) ? (SS_X6_WMODE_SMALLG >= 0 ? SS_X6_WMODE_SMALLG : (BF_ == 3 ? 1 : ((WM * WN >= 8 || MI * NI < 8) ? 2 : 0))) : 0;
    // (bf16x6, 128 co x 256 voxels on four waves: the second register set of mode 2 spills)
    static constexpr bool SP = WMODE == 1, LA = WMODE == 2;
    static constexpr int IN_ALL = IN_FLOATS;
    static constexpr int BUF_FLOATS = IN_ALL + W_FLOATS;
    static constexpr int LDS_FLOATS = BUF_FLOATS * (DB ? 2 : 1);
    static_assert(!BF || CK % (2 * CPH) == 0, "split-staged: a chunk holds whole k-groups");
    static_assert(BLK || NSEG % COLS == 0, "segments must fill whole rows");
    static_assert(!BLK || (BF_ >= 2 && PMAX_ == 0 && NI_ == 2 && NLIVE <= NSEG && NLIVE > NSEG - NI_ && XP % 32 == 8 && XP >= 4 * XL), "block tiles: split-staged, the last wave may lack its second block");
    static_assert(CK % 4 == 0 || (DB && GL_ && CK == 2 && !BF), "channel chunk is a multiple of the packed sub-chunk (4), or one channel pair (GL)");
    static_assert(LDS_FLOATS * 4 <= (X6 ? 160 : 80) * 1024, "two workgroups per CU (x6 eight-wave tiles: one)");
    static_assert(!X6 || (!DB && !GL && G >= 2 && CK % 2 == 0), "x6: 2-D / 3-D tiles, two weight phases");
    static_assert(!FLAT || (KH == 3 && KW == 3 && PMAX % 4 == 0), "flat tiles: 3x3 taps");
    static_assert(DB == GL_, "two LDS buffers are the direct-to-LDS form");
    static_assert(!GL || (DB && !BF), "direct-to-LDS staging is the double-buffered fp32 form");
    // workgroups per CU the register allocator must leave room for: the GL forms carry no staging registers, so the tiles whose
    // two LDS buffers fit three times into the CU's 160 KB are held to 168 VGPRs (3 waves per SIMD instead of 2)
    // (the second __launch_bounds__ argument is waves per SIMD: an eight-wave x6 workgroup alone on its CU is two per SIMD as well)
    static constexpr int NWAVES = WM * WN;
    static constexpr int MIN_WG = X6 ? (MI * NI > 8 ? 1 : NWAVES >= 8 ? 2 : (LDS_FLOATS * 4 * 2 <= 160 * 1024 ? NWAVES / 2 : NWAVES / 4))   // split-staged tiles: two waves per SIMD (256 registers) where LDS allows
                                     : ((GL && LDS_FLOATS * 4 * 4 <= 140 * 1024 && MI * NI <= 4) ? MIN_WG4 : ((GL && LDS_FLOATS * 4 * 3 <= 160 * 1024) ? 3 : 2));
};

// f16x3 split of the same position of a channel pair (x0: channel 2p, x1: channel 2p + 1) into the two words the staged planes hold:
// hw = (hi(x0), hi(x1)), lw = (lo(x0), lo(x1)) with hi = fp16(x / 4), lo = fp16((x / 4 - hi) * 2^11) -- the arithmetic of split3's f16
// branch, bit for bit (x / 4 and the remainder are exact in fp32, so each term is rounded once), as six mixed-precision FMAs that write
// the fp16 halves in place: per value 3 VALU instructions instead of 6.5 (multiply, two conversions, subtract, scale-and-convert, pack).
// The staging of a 1x1 tile is ~150 VALU instructions per 24 MFMAs, two thirds of them this split.
__device__ __forceinline__ void split_pair_f16(const float x0, const float x1, unsigned int& hw, unsigned int& lw) {
    const float qs = STEMSEG_F16X3_ACT_SCALE, ks = 2048.0f;           // (neither is an inline constant: one SGPR each)
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

template <class C>
__global__ __launch_bounds__(C::NTHREADS, C::MIN_WG) void conv_igemm_kernel(const ConvKParams p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS];
    float* const in_lds = smem;
    float* const w_lds = smem + C::IN_ALL;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int half = lane >> 5, l31 = lane & 31;

    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch rule, speed only), so give every XCD a
    // contiguous run of tiles -- neighbouring tiles share halo rows / t-planes through that XCD's L2.
    int bx;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    // the output-channel tile runs fastest: the Cout / MT workgroups that read the SAME input tile are neighbours in launch
    // order on one XCD, so the tile comes from HBM once and from that XCD's L2 for the others (1x1 expansions have up to 16)
    const int co_tile = bx % p.n_co;
    bx /= p.n_co;
    // 3-D taps: t runs fastest, so the workgroups resident on one XCD at a time cover all t-planes of a few (x, y) tiles and
    // the t-1 / t+1 planes every tile needs are L2 hits instead of a second and third HBM fetch (STEMSEG_T_FASTEST=0: x fastest)
    int tx, ty, t;
    if (C::KT > 1 && p.t_fastest) {
        t = bx % p.T;
        bx /= p.T;
        tx = bx % p.tiles_x;
        ty = bx / p.tiles_x;
    } else {
        tx = bx % p.tiles_x;
        bx /= p.tiles_x;
        ty = bx % p.tiles_y;
        t = bx / p.tiles_y;
    }
    const int x0 = C::FLAT ? 0 : tx * C::TW, y0 = C::FLAT ? 0 : ty * C::ROWS;
    // position of lane column l (0..31) of column block s inside the tile: (row, column)
    auto seg_row = [](const int sN, const int l) __attribute__((always_inline)) { return C::BLK ? (sN / C::COLS) * 4 + (l >> 3) : sN / C::COLS; };
    auto seg_col = [](const int sN, const int l) __attribute__((always_inline)) { return C::BLK ? (sN % C::COLS) * 8 + (l & 7) : (sN % C::COLS) * 32 + l; };
    const bool ni1_live = !C::BLK || (wn * C::NI + 1 < C::NLIVE);      // (wave-uniform) the wave's second column block exists
    const int pitch = (int)p.in_ys;                           // FLAT: row pitch of the haloed plane
    const int F0 = pitch + tx * C::NT;                        // FLAT: first flat position of this tile (row 1, column 0)
    const int co0 = co_tile * C::MT;

    f32x16 acc[C::MI][C::NI];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const float* a_ptr = w_lds + half * C::MT + wm * (C::MI * 32) + l31;
    const float* b_ptr[C::NI];
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni) {
        const int s = wn * C::NI + ni;
        b_ptr[ni] = C::FLAT ? in_lds + half * C::IN_CH_STRIDE + s * 32 + l31 + 3       // staged run starts at F0 - pitch - 4
                            : in_lds + half * C::IN_CH_STRIDE + seg_row(s, l31) * C::XP + seg_col(s, l31);
    }

    const unsigned int* b_ptr6[C::NI];  // x6: word (channel pair) planes; lane half h owns the pairs [h*CPH/2, (h+1)*CPH/2) of every k-group
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni) {
        const int s = wn * C::NI + ni;
        b_ptr6[ni] = C::FLAT ? reinterpret_cast<const unsigned int*>(in_lds) + half * (C::CPH / 2) * C::IN_PAIR_STRIDE + s * 32 + l31 + 3   // (the staged run starts at F0 - pitch - 4)
                             : reinterpret_cast<const unsigned int*>(in_lds) + half * (C::CPH / 2) * C::IN_PAIR_STRIDE + seg_row(min(s, C::NLIVE - 1), l31) * C::XP + seg_col(min(s, C::NLIVE - 1), l31);
    }
    const float* in_tile = p.in + (int64_t)blockIdx.y * p.in_bs + (int64_t)t * p.in_ts + x0;   // + c*cs + dt*ts + yy*ys
    const int64_t tile_base = (int64_t)t * p.in_ts + x0;

    // ---- staging helpers ------------------------------------------------------------------------------
    constexpr int XQ = C::XP / 4;
    constexpr int NQ = C::FLAT ? C::IN_FLOATS / 4 : C::CK * C::KT * C::RH * XQ;   // 16-B pieces of the input halo tile
    constexpr int MQ = C::MT / 4;
    constexpr int NWQ = C::BF ? C::NPL * C::G * 2 * C::MT : C::CK * C::TAPS * MQ;   // 16-B pieces of the weight slab
    constexpr int IN_PT = (NQ + C::NTHREADS - 1) / C::NTHREADS, W_PT = (NWQ + C::NTHREADS - 1) / C::NTHREADS;
    auto fetch_in = [&](int c0, int q) -> float4 {            // piece q of the input tile for chunk c0 (vec4 layout)
        if constexpr (C::FLAT) {                              // [c][dt][FL]: flat run of the plane from F0 - pitch - 4
            constexpr int FQ = C::FL / 4;
            const int j4 = q % FQ, rr = q / FQ, dt = rr % C::KT, c = rr / C::KT;
            const int f = F0 - pitch - 4 + 4 * j4;
            const int64_t rel = (int64_t)(c0 + c) * p.in_cs + (int64_t)dt * p.in_ts + f;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 + c < p.Cin && f >= 0 && 4 * j4 < C::NT + 2 * pitch + 8 && tile_base + rel + 4 <= p.in_limit)
                v = *reinterpret_cast<const float4*>(in_tile + rel);
            return v;
        }
        const int xq = q % XQ;
        int rr = q / XQ;
        const int r = rr % C::RH;
        rr /= C::RH;
        const int dt = rr % C::KT;
        const int c = rr / C::KT;
        const int yy = min(y0 + r, p.in_H - 1);
        const int64_t rel = (int64_t)(c0 + c) * p.in_cs + (int64_t)dt * p.in_ts + (int64_t)yy * p.in_ys + xq * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + c < p.Cin && tile_base + rel + 4 <= p.in_limit) v = *reinterpret_cast<const float4*>(in_tile + rel);
        return v;
    };
    auto fetch_w = [&](int c0, int q) -> float4 {             // rows (sub, tap, c4) x MT output channels
        const int mq = q % MQ;
        if constexpr (C::CK == 2) {                           // one channel pair: LDS rows (tap, e) <- packed rows (tap, cp*2 + e)
            const int row2 = q / MQ, tap = row2 >> 1, e = row2 & 1;
            const float* wsrc2 = p.wpk + ((int64_t)(c0 / 4) * (C::TAPS * 4) + tap * 4 + ((c0 >> 1) & 1) * 2 + e) * p.Cout + co0;
            float4 v2 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 + e < p.Cin && co0 + mq * 4 < p.Cout) v2 = *reinterpret_cast<const float4*>(wsrc2 + mq * 4);
            return v2;
        }
        const int row = q / MQ;                               // = (sub*TAPS + tap)*4 + c4
        const int ch = c0 + (row / (C::TAPS * 4)) * 4 + (row & 3);
        const float* wsrc = p.wpk + (int64_t)(c0 / 4) * (C::TAPS * 4) * p.Cout + co0;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch < p.Cin && co0 + mq * 4 < p.Cout) v = *reinterpret_cast<const float4*>(wsrc + (int64_t)row * p.Cout + mq * 4);
        return v;
    };
    // GL: source address of piece q (same decomposition as fetch_in / fetch_w), clamped into the volume
    auto addr_in = [&](int c0, int q) -> const float* {
        int64_t rel;
        if constexpr (C::FLAT) {
            constexpr int FQ = C::FL / 4;
            const int j4 = q % FQ, rr = q / FQ, dt = rr % C::KT, c = rr / C::KT;
            const int f = max(F0 - pitch - 4 + 4 * j4, 0);
            rel = (int64_t)(c0 + c) * p.in_cs + (int64_t)dt * p.in_ts + f;
        } else {
            const int xq = q % XQ;
            int rr = q / XQ;
            const int r = rr % C::RH;
            rr /= C::RH;
            const int dt = rr % C::KT;
            const int c = rr / C::KT;
            const int yy = min(y0 + r, p.in_H - 1);
            rel = (int64_t)(c0 + c) * p.in_cs + (int64_t)dt * p.in_ts + (int64_t)yy * p.in_ys + xq * 4;
        }
        const int64_t last = p.in_limit - 4 - tile_base;      // last 16-B piece inside the volume, relative to in_tile
        return in_tile + (rel < last ? rel : last);
    };
    auto addr_w = [&](int c0, int q) -> const float* {
        const int mq = q % MQ;
        if constexpr (C::CK == 2) {
            const int row2 = q / MQ, tap = row2 >> 1, e = row2 & 1;
            return p.wpk + ((int64_t)(c0 / 4) * (C::TAPS * 4) + tap * 4 + ((c0 >> 1) & 1) * 2 + e) * p.Cout + co0 + mq * 4;
        } else {
            const int row = q / MQ;
            return p.wpk + (int64_t)(c0 / 4) * (C::TAPS * 4) * p.Cout + co0 + (int64_t)row * p.Cout + mq * 4;
        }
    };
    auto glds_chunk = [&](int c0, const int off) {            // enqueue the whole chunk c0 into the buffer at float offset `off`
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
        for (int k = 0; k < IN_PT; ++k) {
            const int q = tid + k * C::NTHREADS;
            if (q < NQ) __builtin_amdgcn_global_load_lds((gptr_t)addr_in(c0, q), (lptr_t)(in_lds + off + (q - lane) * 4), 16, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < W_PT; ++k) {
            const int q = tid + k * C::NTHREADS;
            if (q < NWQ) __builtin_amdgcn_global_load_lds((gptr_t)addr_w(c0, q), (lptr_t)(w_lds + off + (q - lane) * 4), 16, 0, 0);
        }
    };
    auto stage_direct = [&](int c0) {                         // global -> LDS, no overlap (scalar fallback for odd strides)
        if (p.vec4) {
            for (int q = tid; q < NQ; q += C::NTHREADS) *reinterpret_cast<float4*>(in_lds + q * 4) = fetch_in(c0, q);
        } else if constexpr (!C::FLAT) {                      // (flat tiles are only launched on 16-B aligned volumes)
            constexpr int NE = C::CK * C::KT * C::RH * C::XP;
            for (int q = tid; q < NE; q += C::NTHREADS) {
                const int xx = q % C::XP;
                int rr = q / C::XP;
                const int r = rr % C::RH;
                rr /= C::RH;
                const int dt = rr % C::KT;
                const int c = rr / C::KT;
                const int yy = min(y0 + r, p.in_H - 1);
                const int64_t rel = (int64_t)(c0 + c) * p.in_cs + (int64_t)dt * p.in_ts + (int64_t)yy * p.in_ys + xx;
                float v = 0.f;
                if (c0 + c < p.Cin && tile_base + rel < p.in_limit) v = in_tile[rel];
                in_lds[q] = v;
            }
        }
        for (int q = tid; q < NWQ; q += C::NTHREADS) *reinterpret_cast<float4*>(w_lds + q * 4) = fetch_w(c0, q);
    };

    // ---- x6 staging --------------------------------------------------------------------------------------
    // exact three-way split of an fp32 value into bf16 terms (both remainders are exact fp32 subtractions)
    auto split3 = [](const float x, unsigned int& h, unsigned int& m, unsigned int& l) {
        if constexpr (C::F16) {
            // f16x3: x * 2^-2 = hi + lo with the low term stored as lo * 2^11 (the weights supply hi_w * 2^-11 for it): hi is a
            // normal fp16 number for 2.5e-4 <= |x| < 2.6e5 and the pair keeps 22 significand bits there (2^-36 absolute when the low
            // term is tiny, and below the range), inf above (and the result says so).  The accumulators are scaled back once, after the chunk loop.
            const float xs = x * STEMSEG_F16X3_ACT_SCALE;
            const _Float16 fh = (_Float16)xs;
            const _Float16 fl = (_Float16)((xs - (float)fh) * 2048.0f);
            h = *reinterpret_cast<const unsigned short*>(&fh);
            m = *reinterpret_cast<const unsigned short*>(&fl);
            l = 0;
        } else {
            const __bf16 bh = (__bf16)x;
            const float r1 = x - (float)bh;
            const __bf16 bm = (__bf16)r1;
            const __bf16 bl = (__bf16)(r1 - (float)bm);
            h = *reinterpret_cast<const unsigned short*>(&bh);
            m = *reinterpret_cast<const unsigned short*>(&bm);
            l = *reinterpret_cast<const unsigned short*>(&bl);
        }
    };
    constexpr int FQ6 = C::FL / 4;                                       // flat: 16-B pieces of one (pair, dt) run
    constexpr int XL6 = C::XL;                                           // staged 16-B pieces per tile row (BLK tiles: fewer than the LDS pitch holds)
    constexpr int NQ6 = C::FLAT ? (C::CK / 2) * C::KT * FQ6 : (C::CK / 2) * C::KT * C::RH * XL6;   // 16-B pieces of a channel PAIR's rows: [pair][dt][row][xq] (flat: [pair][dt][run])
    // word offset of piece q in a plane of the LDS tile (rows of pitch XP; without BLK the pieces ARE the linear image)
    auto in6_lds = [](const int q) __attribute__((always_inline)) { return C::BLK ? (q / XL6) * C::XP + (q % XL6) * 4 : q * 4; };
    constexpr int IN_PT6 = C::X6 ? (NQ6 + C::NTHREADS - 1) / C::NTHREADS : 1;
    constexpr int NWQ_A = C::NPL * C::GA * 2 * C::MT, NWQ6 = C::NPL * C::G * 2 * C::MT;   // 16-B pieces of weight phase A / of the slab
    auto in6_rel = [&](int q, int& c) -> int64_t {                   // piece q -> float offset of its first channel (c0 = 0) from in_tile (< 0: not needed / outside)
        if constexpr (C::FLAT) {                                      // [pair][dt][FL]: flat run of the plane from F0 - pitch - 4
            const int j4 = q % FQ6, rr = q / FQ6, dt = rr % C::KT;
            c = 2 * (rr / C::KT);
            const int f = F0 - pitch - 4 + 4 * j4;
            if (f < 0 || 4 * j4 >= C::NT + 2 * pitch + 8) return -1;  // before the volume (feeds halo-column outputs only) / beyond what this pitch needs
            return (int64_t)c * p.in_cs + (int64_t)dt * p.in_ts + f;
        }
        const int xq = q % XL6;
        int rr = q / XL6;
        const int r = rr % C::RH;
        rr /= C::RH;
        const int dt = rr % C::KT;
        c = 2 * (rr / C::KT);
        const int yy = min(y0 + r, p.in_H - 1);
        return (int64_t)c * p.in_cs + (int64_t)dt * p.in_ts + (int64_t)yy * p.in_ys + xq * 4;
    };
    auto fetch_in6 = [&](int c0, int q, float4& v0, float4& v1) {    // piece q of both channels of its pair
        int c;
        const int64_t rel0 = in6_rel(q, c);
        const int64_t rel = rel0 + (int64_t)c0 * p.in_cs;
        v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rel0 < 0) return;
        if (c0 + c < p.Cin && tile_base + rel + 4 <= p.in_limit) v0 = *reinterpret_cast<const float4*>(in_tile + rel);
        if (c0 + c + 1 < p.Cin && tile_base + rel + p.in_cs + 4 <= p.in_limit) v1 = *reinterpret_cast<const float4*>(in_tile + rel + p.in_cs);
    };
    // the same for the register prefetch inside the chunk loop: the per-thread part of the address is loop invariant (a 32-bit
    // byte offset, computed once), the chunk only moves the uniform base
    unsigned int in6_voff[IN_PT6];
    int in6_clim[IN_PT6];                                             // the piece (channel c0 + its pair's first channel) is inside the volume iff c0 < clim
    const int64_t in6_room = (p.in_limit - 4 - tile_base) * 4;        // last valid 16-B piece, as a byte offset from in_tile
    if constexpr (C::X6) {
#pragma unroll
        for (int k = 0; k < IN_PT6; ++k) {
            const int q = tid + k * C::NTHREADS;
            int c = 0;
            const int64_t rel0 = q < NQ6 ? in6_rel(q, c) : 0;
            const int64_t rel = rel0 < 0 ? 0 : rel0;
            in6_voff[k] = (unsigned int)(rel * 4);
            const int64_t slack = in6_room - rel * 4;                   // c0 * cs * 4 <= slack
            const int64_t by_room = slack < 0 ? 0 : slack / (p.in_cs * 4) + 1;
            in6_clim[k] = (q < NQ6 && rel0 >= 0) ? (int)min((int64_t)(p.Cin - c), by_room) : 0;
        }
    }
    // branch-free (the loads are issued between the MFMAs of the running chunk): a lane whose piece lies outside the volume reads
    // the tile origin instead and in6_mask() zeroes it when it is written to LDS.  (Second channel of the pair: the same test one
    // channel further.)
    typedef float f32x4 __attribute__((ext_vector_type(4)));          // (a native vector: struct copies of float4 stay memcpys)
    auto fetch_in6_fast = [&](int c0, int k, f32x4& v0, f32x4& v1) __attribute__((always_inline)) {
#if SS_PROBE == 1
        return;
#endif
        const char* base = reinterpret_cast<const char*>(in_tile) + (int64_t)c0 * p.in_cs * 4;
        const unsigned int cs4 = (unsigned int)(p.in_cs * 4);          // (a chunk of channels spans < 4 GB: in6_voff already relies on it)
        v0 = *reinterpret_cast<const f32x4*>(base + (c0 < in6_clim[k] ? in6_voff[k] : 0u));
        v1 = *reinterpret_cast<const f32x4*>(base + (c0 + 1 < in6_clim[k] ? in6_voff[k] + cs4 : 0u));
    };
    auto store_in6 = [&](int q, const float4& v0, const float4& v1) { // split both channels, interleave, 16-B stores into the tile
        unsigned int h0, m0, l0, h1, m1, l1;
        uint4 ph, pm, pl;
        if constexpr (C::F16) {                                       // (three mixed-precision FMAs per value, no packing: see split_pair_f16)
            split_pair_f16(v0.x, v1.x, ph.x, pm.x); split_pair_f16(v0.y, v1.y, ph.y, pm.y);
            split_pair_f16(v0.z, v1.z, ph.z, pm.z); split_pair_f16(v0.w, v1.w, ph.w, pm.w);
            unsigned int* d16 = reinterpret_cast<unsigned int*>(in_lds) + in6_lds(q);
            *reinterpret_cast<uint4*>(d16) = ph;
            *reinterpret_cast<uint4*>(d16 + C::IN_PLANE_STRIDE) = pm;
            return;
        }
        split3(v0.x, h0, m0, l0); split3(v1.x, h1, m1, l1); ph.x = h0 | (h1 << 16); pm.x = m0 | (m1 << 16); pl.x = l0 | (l1 << 16);
        split3(v0.y, h0, m0, l0); split3(v1.y, h1, m1, l1); ph.y = h0 | (h1 << 16); pm.y = m0 | (m1 << 16); pl.y = l0 | (l1 << 16);
        split3(v0.z, h0, m0, l0); split3(v1.z, h1, m1, l1); ph.z = h0 | (h1 << 16); pm.z = m0 | (m1 << 16); pl.z = l0 | (l1 << 16);
        split3(v0.w, h0, m0, l0); split3(v1.w, h1, m1, l1); ph.w = h0 | (h1 << 16); pm.w = m0 | (m1 << 16); pl.w = l0 | (l1 << 16);
        unsigned int* d = reinterpret_cast<unsigned int*>(in_lds) + in6_lds(q);
        *reinterpret_cast<uint4*>(d) = ph;
        *reinterpret_cast<uint4*>(d + C::IN_PLANE_STRIDE) = pm;
        if constexpr (C::NPX == 3) *reinterpret_cast<uint4*>(d + 2 * C::IN_PLANE_STRIDE) = pl;
    };
    auto store_in6_masked = [&](int c0, int k, const f32x4& r0, const f32x4& r1) __attribute__((always_inline)) {
        const bool ok0 = c0 < in6_clim[k], ok1 = c0 + 1 < in6_clim[k];
        float4 v0, v1;
        v0.x = ok0 ? r0.x : 0.f; v0.y = ok0 ? r0.y : 0.f; v0.z = ok0 ? r0.z : 0.f; v0.w = ok0 ? r0.w : 0.f;
        v1.x = ok1 ? r1.x : 0.f; v1.y = ok1 ? r1.y : 0.f; v1.z = ok1 ? r1.z : 0.f; v1.w = ok1 ? r1.w : 0.f;
        store_in6(tid + k * C::NTHREADS, v0, v1);
    };
    auto stage_in6_direct = [&](int c0) {        // global -> split -> LDS without overlap (prologue, odd strides)
        if (p.vec4) {
            for (int q = tid; q < NQ6; q += C::NTHREADS) {
                float4 v0, v1;
                fetch_in6(c0, q, v0, v1);
                store_in6(q, v0, v1);
            }
        } else if constexpr (!C::FLAT) {                              // (flat tiles are only launched on 16-B aligned volumes)
            constexpr int NE6 = (C::CK / 2) * C::KT * C::RH * C::XP;     // one word (pair, position) per iteration
            for (int q = tid; q < NE6; q += C::NTHREADS) {
                const int xx = q % C::XP;
                int rr = q / C::XP;
                const int r = rr % C::RH;
                rr /= C::RH;
                const int dt = rr % C::KT;
                const int c = 2 * (rr / C::KT);
                const int yy = min(y0 + r, p.in_H - 1);
                const int64_t rel = (int64_t)(c0 + c) * p.in_cs + (int64_t)dt * p.in_ts + (int64_t)yy * p.in_ys + xx;
                float a0 = 0.f, a1 = 0.f;
                if (c0 + c < p.Cin && tile_base + rel < p.in_limit) a0 = in_tile[rel];
                if (c0 + c + 1 < p.Cin && tile_base + rel + p.in_cs < p.in_limit) a1 = in_tile[rel + p.in_cs];
                unsigned int h0, m0, l0, h1, m1, l1;
                split3(a0, h0, m0, l0);
                split3(a1, h1, m1, l1);
                unsigned int* d = reinterpret_cast<unsigned int*>(in_lds) + q;
                d[0] = h0 | (h1 << 16);
                d[C::IN_PLANE_STRIDE] = m0 | (m1 << 16);
                if constexpr (C::NPX == 3) d[2 * C::IN_PLANE_STRIDE] = l0 | (l1 << 16);
            }
        }
    };
    // weight pieces of chunk c0's slab (packed order [grp][plane][half][co] = the LDS image); columns past Cout are fetched from
    // the row's last valid channel (they feed rows never stored)
    // (two-phase tiles hold one phase at a time -- phase A is the larger; single-phase tiles hold the whole slab)
    constexpr int W_PT6 = C::X6 ? ((C::SP ? NWQ6 : NWQ_A) + C::NTHREADS - 1) / C::NTHREADS : 1;
    // address = uniform base (chunk, first row of the piece run, co0) + ONE loop-invariant 32-bit per-thread offset: NTHREADS and the
    // phase boundaries are multiples of MT, so a thread keeps its column and walks the rows in steps of NTHREADS / MT
    static_assert(C::NTHREADS % C::MT == 0 && NWQ_A % C::MT == 0, "weight pieces: a thread keeps its column");
    const unsigned int w6_voff = C::X6 ? ((unsigned int)(tid / C::MT) * (unsigned int)p.Cout + (unsigned int)min(tid % C::MT, p.Cout - 1 - co0)) * 16u : 0u;
    auto w6_src = [&](int c0, int q_uniform) __attribute__((always_inline)) -> const char* {      // q_uniform = q - tid (a multiple of MT)
        const char* base = reinterpret_cast<const char*>(p.wpk) +
                           (((int64_t)(c0 / C::CK) * (C::NPL * C::G * 2) + q_uniform / C::MT) * p.Cout + co0) * 16;
        return base + w6_voff;
    };
    // MFMA stream over the k-groups [g0, g1) of the staged chunk: 3 x MI b128 (A) + 3 x NI x 4 b32 (B) per 6 x MI x NI MFMAs
    // side(step) is called once per (k-group, mi) step, in front of its MFMAs: the chunk loop hangs the next chunk's global loads
    // there, a few per step (all of them at the top of the chunk back the texture path up and the waves stall AT ISSUE, with the
    // matrix pipe idle behind them: measured 17% of a 1x1 layer)
    auto compute6 = [&](auto g0c, auto g1c, auto&& side, auto live_c) __attribute__((always_inline)) {
        constexpr int g0 = decltype(g0c)::value, g1 = decltype(g1c)::value;
        constexpr bool LIVE1 = decltype(live_c)::value;       // the wave's column blocks beyond the first exist (BLK tiles: not in the last wave)
        constexpr int NPL = C::NPL, NPX = C::NPX, NPA = C::NPA;
        typedef typename std::conditional<C::F16, _Float16, __bf16>::type h16;
        typedef h16 h16x8 __attribute__((ext_vector_type(8)));
        const char* a_base = reinterpret_cast<const char*>(w_lds) + (half * C::MT + wm * (C::MI * 32) + l31) * 16;
        const unsigned int* bdy6[C::NI][3];                   // FLAT: row dy of the taps = + dy * pitch (runtime), everything else immediates
        if constexpr (C::FLAT) {
#pragma unroll
            for (int ni = 0; ni < C::NI; ++ni) { bdy6[ni][0] = b_ptr6[ni]; bdy6[ni][1] = b_ptr6[ni] + pitch; bdy6[ni][2] = b_ptr6[ni] + 2 * pitch; }
        }
        auto ld_b = [&](const int grp, h16x8 (&b)[NPX][C::NI]) __attribute__((always_inline)) {
#if SS_PROBE == 4
            if (grp >= 0) { _Pragma("unroll") for (int pl = 0; pl < NPX; ++pl) _Pragma("unroll") for (int ni = 0; ni < C::NI; ++ni) asm volatile("" : "+v"(b[pl][ni])); return; }
#endif
            const int cg = grp / C::NTG, tg = grp % C::NTG;
#pragma unroll
            for (int ni = 0; ni < (LIVE1 ? C::NI : 1); ++ni)
#pragma unroll
                for (int pl = 0; pl < NPX; ++pl) {
                    uint4 w4;
                    unsigned int* wv = reinterpret_cast<unsigned int*>(&w4);
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        const int j0 = 2 * wd, tapi = j0 / C::CPH, chl = j0 % C::CPH;
                        int tap = tg * C::TPG + tapi;
                        tap = tap < C::TAPS ? tap : C::TAPS - 1;       // padded taps: any valid address (their weights are zero)
                        const int dt = tap / (C::KH * C::KW), dy = (tap / C::KW) % C::KH, dx = tap % C::KW;
                        if constexpr (C::FLAT) {
                            const int off = pl * C::IN_PLANE_STRIDE + (cg * C::CPH + chl / 2) * C::IN_PAIR_STRIDE + dt * C::FL + dx;
                            wv[wd] = bdy6[ni][dy][off];
                        } else {
                            const int off = pl * C::IN_PLANE_STRIDE + (cg * C::CPH + chl / 2) * C::IN_PAIR_STRIDE + (dt * C::RH + dy) * C::XP + dx;
                            wv[wd] = b_ptr6[ni][off];
                        }
                    }
                    b[pl][ni] = __builtin_bit_cast(h16x8, w4);
                }
        };
        auto ld_a = [&](const int grp, const int mi, h16x8 (&a)[NPA]) __attribute__((always_inline)) {
#if SS_PROBE == 4
            if (grp >= 0) { _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) asm volatile("" : "+v"(a[pl])); return; }
#endif
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                a[pl] = *reinterpret_cast<const h16x8*>(a_base + (((grp * NPL + pl) * 2) * C::MT + mi * 32) * 16);
        };
        // smallest products first (planes: 0 hi, 1 mid / lo, 2 lo); consecutive MFMAs alternate between the NI accumulators of this mi
        auto mm = [&](const int mi, h16x8 (&a)[NPA], const h16x8 (&b)[NPX][C::NI]) __attribute__((always_inline)) {
#if SS_PROBE == 3
            _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) asm volatile("" ::"v"(a[pl]));
            _Pragma("unroll") for (int pl = 0; pl < NPX; ++pl) _Pragma("unroll") for (int ni = 0; ni < C::NI; ++ni) asm volatile("" ::"v"(b[pl][ni]));
            return;
#endif
            if constexpr (C::F16) {
                // two staged planes: the weights' hi * 2^-11 operand (it meets the input tile's lo * 2^11 plane) is made here, four packed
                // multiplies in the shadow of the MFMAs (exact: a power of two on a normal number; the same rounding as the packed
                // third plane otherwise)
                if constexpr (NPL == 2) a[2] = a[0] * (h16)(1.0f / 2048.0f);
#define SS_X6_TERM(PA, PB)                                                                                                     \
    _Pragma("unroll") for (int ni = 0; ni < (LIVE1 ? C::NI : 1); ++ni)                                                         \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[PA], b[PB][ni], acc[mi][ni], 0, 0, 0);
                SS_X6_TERM(1, 0) SS_X6_TERM(2, 1) SS_X6_TERM(0, 0)      // lo_w * hi_x, (hi_w 2^-11) * (lo_x 2^11), hi_w * hi_x
#undef SS_X6_TERM
            } else {
#define SS_X6_TERM(PA, PB)                                                                                                     \
    _Pragma("unroll") for (int ni = 0; ni < (LIVE1 ? C::NI : 1); ++ni)                                                         \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], b[PB][ni], acc[mi][ni], 0, 0, 0);
                SS_X6_TERM(2, 0) SS_X6_TERM(0, 2) SS_X6_TERM(1, 1) SS_X6_TERM(1, 0) SS_X6_TERM(0, 1) SS_X6_TERM(0, 0)
#undef SS_X6_TERM
            }
        };
        // software pipeline over the (k-group, mi) steps: the A fragments of step s + 1 are requested from LDS BEFORE the
        // 6 x NI MFMAs of step s are issued (two register sets), a group's B fragments right after the previous group's last
        // MFMAs -- left to itself the compiler reads each fragment right in front of the MFMAs that consume it and waits there
        constexpr int NSTEP = (g1 - g0) * C::MI;
        h16x8 bfr[NPX][C::NI], a0[NPA], a1[NPA];
        side(-1);                                          // (tiles that do not spread issue everything here, ahead of the fragments)
        ld_b(g0, bfr);
        ld_a(g0, 0, a0);
#pragma unroll
        for (int st = 0; st < NSTEP; st += 2) {
            const int grp = g0 + st / C::MI, mi = st % C::MI;
            if (st + 1 < NSTEP) ld_a(g0 + (st + 1) / C::MI, (st + 1) % C::MI, a1);
            side(st);
            __builtin_amdgcn_sched_barrier(0);
            mm(mi, a0, bfr);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < NSTEP) {
                if ((st + 1) % C::MI == 0) ld_b(g0 + (st + 1) / C::MI, bfr);
                if (st + 2 < NSTEP) ld_a(g0 + (st + 2) / C::MI, (st + 2) % C::MI, a0);
                side(st + 1);
                __builtin_amdgcn_sched_barrier(0);
                mm((st + 1) % C::MI, a1, bfr);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 2 < NSTEP && (st + 2) % C::MI == 0) ld_b(g0 + (st + 2) / C::MI, bfr);
            }
            (void)grp;
        }
    };
    // ---- MFMA stream over one staged chunk: every tap is a shifted LDS read ------------------------------
    auto compute = [&](const int buf_off = 0) {
        constexpr int NSUB = C::CK >= 4 ? C::CK / 4 : 1, NCP = C::CK >= 4 ? 2 : 1, WROWS = C::CK >= 4 ? 4 : 2;
        const float* bdy[C::NI][3];                           // FLAT: row dy of the taps = + dy * pitch (runtime), rest immediates
        if constexpr (C::FLAT) {
#pragma unroll
            for (int ni = 0; ni < C::NI; ++ni) {
                bdy[ni][0] = b_ptr[ni]; bdy[ni][1] = b_ptr[ni] + pitch; bdy[ni][2] = b_ptr[ni] + 2 * pitch;
            }
        }
        // software-pipelined k-steps: the operands of step i + 1 are requested from LDS BEFORE the eight MFMAs of step i are
        // issued, so their latency hides under 512 cycles of matrix work (left to itself the compiler reads each A pair right
        // in front of the MFMAs that consume it and waits for it there).  Measured neutral at 2 waves per SIMD -- the other
        // wave already covered those waits -- but it keeps a single resident wave from stalling.
        constexpr int NSTEPS = NSUB * C::TAPS * NCP;
        auto ld = [&](const int i, float (&a)[C::MI], float (&b)[C::NI]) {
            const int cp = i % NCP, tap = (i / NCP) % C::TAPS, sub = i / (NCP * C::TAPS);
            const int dt = tap / (C::KH * C::KW), dy = (tap / C::KW) % C::KH, dx = tap % C::KW;
            const int wrow = (sub * C::TAPS + tap) * WROWS + cp * 2;
            const int boff = C::FLAT ? (sub * 4 + cp * 2) * C::IN_CH_STRIDE + dt * C::FL + dx
                                     : (sub * 4 + cp * 2) * C::IN_CH_STRIDE + (dt * C::RH + dy) * C::XP + dx;
#pragma unroll
            for (int mi = 0; mi < C::MI; ++mi) a[mi] = a_ptr[buf_off + wrow * C::MT + mi * 32];
#pragma unroll
            for (int ni = 0; ni < C::NI; ++ni) {
                if constexpr (C::FLAT) b[ni] = bdy[ni][dy][buf_off + boff];
                else b[ni] = b_ptr[ni][buf_off + boff];
            }
        };
        auto mm = [&](const float (&a)[C::MI], const float (&b)[C::NI]) {
#pragma unroll
            for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < C::NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        };
        float a0[C::MI], b0[C::NI], a1[C::MI], b1[C::NI];
        ld(0, a0, b0);
#pragma unroll
        for (int i = 0; i < NSTEPS; i += 2) {
            if (i + 1 < NSTEPS) ld(i + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);             // (the scheduler would otherwise sink the reads back to their uses)
            mm(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < NSTEPS) {
                if (i + 2 < NSTEPS) ld(i + 2, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mm(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    const int c_begin = blockIdx.z * p.chunks_per_split * C::CK;
    const int c_end = min(p.Cin, c_begin + p.chunks_per_split * C::CK);
    if constexpr (C::X6) {
        // Register-staged software pipeline: the next chunk's 16-B pieces are loaded into registers between the MFMAs of the running
        // chunk and written to LDS when the slots they replace fall idle (three schedules below, picked per tile by C::WMODE).
        typedef std::integral_constant<int, 0> Q0;
        typedef std::integral_constant<int, NWQ_A> QA;
        typedef std::integral_constant<int, NWQ6> QE;
        f32x4 rin[2 * IN_PT6];
        f32x4 rw6[W_PT6];
        f32x4 rw6b[W_PT6];                                            // (lookahead tiles: phase B's own register set)
        auto fetch_w6_k = [&](int c0, auto q0c, auto q1c, f32x4 (&r)[W_PT6], const int k) __attribute__((always_inline)) {      // piece k of [q0, q1)
            constexpr int q0 = decltype(q0c)::value, q1 = decltype(q1c)::value;
#if SS_PROBE == 1
            return;
#endif
            if ((k + 1) * C::NTHREADS <= q1 - q0 || q0 + tid + k * C::NTHREADS < q1) r[k] = *reinterpret_cast<const f32x4*>(w6_src(c0, q0 + k * C::NTHREADS));
        };
        auto store_w6 = [&](auto q0c, auto q1c, const f32x4 (&r)[W_PT6]) __attribute__((always_inline)) {
            constexpr int q0 = decltype(q0c)::value, q1 = decltype(q1c)::value;
#if SS_PROBE == 2
            _Pragma("unroll") for (int k = 0; k < W_PT6; ++k) asm volatile("" ::"v"(r[k]));
            return;
#endif
    #pragma unroll
            for (int k = 0; k < W_PT6; ++k) { const int q = q0 + tid + k * C::NTHREADS; if ((k + 1) * C::NTHREADS <= q1 - q0 || (k * C::NTHREADS < q1 - q0 && q < q1)) *reinterpret_cast<f32x4*>(w_lds + q * 4) = r[k]; }
        };
        // the next chunk's input tile, written between barriers (the one tile is what the MFMA stream reads)
        auto store_in6_all = [&](int c0) __attribute__((always_inline)) {
#if SS_PROBE == 2
            _Pragma("unroll") for (int k = 0; k < 2 * IN_PT6; ++k) asm volatile("" ::"v"(rin[k]));
            return;
#endif
            if (p.vec4) {
#pragma unroll
                for (int k = 0; k < IN_PT6; ++k) {
                    const int q = tid + k * C::NTHREADS;
                    if ((k + 1) * C::NTHREADS <= NQ6 || q < NQ6) store_in6_masked(c0, k, rin[2 * k], rin[2 * k + 1]);
                }
            } else stage_in6_direct(c0);
        };
        // side work of one phase, spread over its (k-group, mi) steps: global loads of the next chunk (input-tile piece pairs, then
        // weight pieces) over the steps [0, NL)
        auto side_items = [&](const int st, auto nspread_c, auto n_in_c, auto n_w_c, auto&& f_in, auto&& f_w) __attribute__((always_inline)) {
            constexpr int nspread = decltype(nspread_c)::value, n_in = decltype(n_in_c)::value, n_w = decltype(n_w_c)::value, n = n_in + n_w;
            if (st >= nspread || st < 0) return;
#pragma unroll
            for (int i = 0; i < n; ++i)
                if (i >= st * n / nspread && i < (st + 1) * n / nspread) { if (i < n_in) f_in(i); else f_w(i - n_in); }
        };
        constexpr int NSA = C::GA * C::MI, NSB = (C::G - C::GA) * C::MI, NSALL = C::G * C::MI;
        constexpr int NPA = (NWQ_A + C::NTHREADS - 1) / C::NTHREADS, NPB = (NWQ6 - NWQ_A + C::NTHREADS - 1) / C::NTHREADS;
        constexpr int NPALL = (NWQ6 + C::NTHREADS - 1) / C::NTHREADS;
        constexpr bool SPRD = SS_X6_SPREAD && (C::F16 || SS_X6_SPREAD > 1);     // (bf16x6: no gain on the 1x1 tiles, no registers to spare on the many-k-group ones)
        typedef std::integral_constant<int, C::GA> GAc;
        typedef std::integral_constant<int, C::G> Gc;
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, IN_PT6> INc;
        if (c_begin < c_end) {
            for (int q = tid; q < NWQ6; q += C::NTHREADS) *reinterpret_cast<float4*>(w_lds + q * 4) = *reinterpret_cast<const float4*>(w6_src(c_begin, q - tid));
            stage_in6_direct(c_begin);
        }
        __syncthreads();
        // The chunk bodies are instantiated per (another chunk follows, two more follow) instead of testing it at run time: with the
        // fetches and the stores under separate run-time `if`s the waitcnt pass cannot pair them up, believes loads into the staging
        // registers may still be in flight at the loop head, and guards the first address computation that reuses one with an
        // in-order vmcnt wait -- which then waits for the loads just issued in front of it.
        typedef std::true_type Yes;
        typedef std::false_type No;
        auto f_in_at = [&](const int cn) { return [&, cn](const int k) __attribute__((always_inline)) { if (p.vec4) fetch_in6_fast(cn, k, rin[2 * k], rin[2 * k + 1]); }; };
        if constexpr (C::SP) {
            // few k-groups per chunk (1x1 taps): one phase; the whole next chunk (weights + input tile) gathers in registers under
            // the chunk's MFMA stream.  Two barriers per chunk.
            auto chunk = [&](const int c0, auto more_c) __attribute__((always_inline)) {
                constexpr bool more = decltype(more_c)::value;
                const int cn = c0 + C::CK;
                compute6(Q0{}, Gc{}, [&](const int st) __attribute__((always_inline)) {
                    if constexpr (more) {
                        side_items(SPRD ? st : -1 - st, std::integral_constant<int, SPRD ? (NSALL + SS_X6_SPREAD_DIV - 1) / SS_X6_SPREAD_DIV : 1>{}, INc{}, std::integral_constant<int, NPALL>{}, f_in_at(cn),
                                   [&](const int k) __attribute__((always_inline)) { fetch_w6_k(cn, Q0{}, QE{}, rw6, k); });
                    }
                }, Yes{});
                SS_CHUNK_SYNC();
                if constexpr (more) {
                    store_w6(Q0{}, QE{}, rw6);
                    store_in6_all(cn);
                    SS_CHUNK_SYNC();
                }
            };
            int c0 = c_begin;
            for (; c0 + C::CK < c_end; c0 += C::CK) chunk(c0, Yes{});
            if (c0 < c_end) chunk(c0, No{});
        } else if constexpr (C::LA) {
            // two phases as below, with the weights one more chunk ahead and a register set per phase: phase A's set is refilled
            // (chunk + 2) under phase B, phase B's set and the input tile under phase A -- every global load has at least half a
            // chunk of MFMA stream to land, and phase A's LDS writes still hide under phase B.
            if (c_begin + C::CK < c_end) {
#pragma unroll
                for (int k = 0; k < NPA; ++k) fetch_w6_k(c_begin + C::CK, Q0{}, QA{}, rw6, k);
            }
            auto chunk = [&](const int c0, auto more_c, auto more2_c) __attribute__((always_inline)) {
                constexpr bool more = decltype(more_c)::value, more2 = decltype(more2_c)::value;
                const int cn = c0 + C::CK, cnn = c0 + 2 * C::CK;
                compute6(Q0{}, GAc{}, [&](const int st) __attribute__((always_inline)) {
                    if constexpr (more) side_items(SPRD ? st : -1 - st, std::integral_constant<int, SPRD ? NSA : 1>{}, INc{}, std::integral_constant<int, NPB>{}, f_in_at(cn),
                                         [&](const int k) __attribute__((always_inline)) { fetch_w6_k(cn, QA{}, QE{}, rw6b, k); });
                }, Yes{});
                SS_CHUNK_SYNC();
                if constexpr (more) store_w6(Q0{}, QA{}, rw6);
                compute6(GAc{}, Gc{}, [&](const int st) __attribute__((always_inline)) {
                    if constexpr (more2) side_items(SPRD ? st : -1 - st, std::integral_constant<int, SPRD ? (NSB + SS_X6_SPREAD_DIV - 1) / SS_X6_SPREAD_DIV : 1>{}, I0{}, std::integral_constant<int, NPA>{}, [](const int) {},
                                          [&](const int k) __attribute__((always_inline)) { fetch_w6_k(cnn, Q0{}, QA{}, rw6, k); });
                }, Yes{});
                SS_CHUNK_SYNC();
                if constexpr (more) {
                    store_w6(QA{}, QE{}, rw6b);
                    store_in6_all(cn);
                    SS_CHUNK_SYNC();
                }
            };
            int c0 = c_begin;
            for (; c0 + 2 * C::CK < c_end; c0 += C::CK) chunk(c0, Yes{}, Yes{});
            if (c0 + C::CK < c_end) { chunk(c0, Yes{}, No{}); c0 += C::CK; }
            if (c0 < c_end) chunk(c0, No{}, No{});
        } else {
            // The weight slab is staged in two k-group phases that ping-pong with the MFMA stream: under phase A the next chunk's
            // phase-A weights and input tile gather in registers; they are written when phase A's slots fall idle, the registers
            // then collect the next chunk's phase-B weights under phase B's MFMA stream.  Three barriers per chunk.
            auto chunk = [&](const int c0, auto more_c, auto live_c) __attribute__((always_inline)) {
                constexpr bool more = decltype(more_c)::value;
                const int cn = c0 + C::CK;
                compute6(Q0{}, GAc{}, [&](const int st) __attribute__((always_inline)) {
                    if constexpr (more) side_items(SPRD ? st : -1 - st, std::integral_constant<int, SPRD ? (NSA + SS_X6_SPREAD_DIV - 1) / SS_X6_SPREAD_DIV : 1>{}, INc{}, std::integral_constant<int, NPA>{}, f_in_at(cn),
                                         [&](const int k) __attribute__((always_inline)) { fetch_w6_k(cn, Q0{}, QA{}, rw6, k); });
                }, live_c);
                SS_CHUNK_SYNC();                                   // phase A's slots are idle
                if constexpr (more) store_w6(Q0{}, QA{}, rw6);
                compute6(GAc{}, Gc{}, [&](const int st) __attribute__((always_inline)) {
                    if constexpr (more) {
                        side_items(SPRD ? st : -1 - st, std::integral_constant<int, SPRD ? (NSB + SS_X6_SPREAD_DIV - 1) / SS_X6_SPREAD_DIV : 1>{}, I0{}, std::integral_constant<int, NPB>{}, [](const int) {},
                                   [&](const int k) __attribute__((always_inline)) { fetch_w6_k(cn, QA{}, QE{}, rw6, k); });
                    }
                }, live_c);
                SS_CHUNK_SYNC();                                   // everyone is done with phase B's slots and this chunk's input tile
                if constexpr (more) {
                    store_w6(QA{}, QE{}, rw6);
                    store_in6_all(cn);
                    SS_CHUNK_SYNC();
                }
            };
            auto run = [&](auto live_c) __attribute__((always_inline)) {
                int c0 = c_begin;
                for (; c0 + C::CK < c_end; c0 += C::CK) chunk(c0, Yes{}, live_c);
                if (c0 < c_end) chunk(c0, No{}, live_c);
            };
            // BLK tiles: the last wave owns ONE column block -- its own instance of the chunk loop, without the second block's fragment reads and MFMAs
            if constexpr (C::BLK && C::NLIVE < C::NSEG) {
                if (ni1_live) run(Yes{});
                else run(No{});
            } else run(Yes{});
        }
        if constexpr (C::F16) {
            // undo the operand scales (powers of two: exact): 1 / (weight scale of the output channel x activation scale), one float per
            // output channel behind the last weight slab (pack_conv_weight_f16x3_kernel).  C/D layout: register r of lane (half, l31)
            // is row (r & 3) + 8 * (r >> 2) + 4 * half -> four consecutive channels per (mi, r >> 2).
            const float* invp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.wpk) +
                                                               (int64_t)((p.Cin + C::CK - 1) / C::CK) * (C::NPL * C::G * 2) * p.Cout * 16);
            const int co_w = co0 + wm * (C::MI * 32) + 4 * half;
#pragma unroll
            for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co4 = co_w + mi * 32 + 8 * j;                      // (Cout % 32 == 0: the four rows are valid together)
                    const float4 sc = co4 < p.Cout ? *reinterpret_cast<const float4*>(invp + co4) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
                    for (int ni = 0; ni < C::NI; ++ni) {
                        acc[mi][ni][4 * j + 0] *= sc.x; acc[mi][ni][4 * j + 1] *= sc.y;
                        acc[mi][ni][4 * j + 2] *= sc.z; acc[mi][ni][4 * j + 3] *= sc.w;
                    }
                }
        }
    } else if constexpr (C::DB && C::GL) {
        // double-buffered LDS filled by LDS-DMA: chunk i+1 is enqueued into the idle buffer, chunk i's MFMA stream runs, and
        // the __syncthreads() that ends the chunk carries the vmcnt(0) that retires the DMA (the compiler puts no wait in
        // front of the ds_reads: checked in the ISA) -- one barrier per chunk, nothing between the last MFMA and the barrier
        if (c_begin < c_end) glds_chunk(c_begin, 0);
        __syncthreads();
        int cur = 0;
        for (int c0 = c_begin; c0 < c_end; c0 += C::CK) {
            if (c0 + C::CK < c_end) glds_chunk(c0 + C::CK, C::BUF_FLOATS - cur);
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            cur = C::BUF_FLOATS - cur;
            __syncthreads();
        }
    } else if (C::PIPE && p.vec4) {
        // software pipeline: the next chunk's global loads are issued into registers BEFORE the MFMA stream of the
        // current chunk and written to LDS after it, so HBM/L2 latency hides under the matrix pipe.
        float4 rin[IN_PT], rw[W_PT];
        auto fetch_regs = [&](int c0) {
#pragma unroll
            for (int k = 0; k < IN_PT; ++k) { const int q = tid + k * C::NTHREADS; if (q < NQ) rin[k] = fetch_in(c0, q); }
#pragma unroll
            for (int k = 0; k < W_PT; ++k) { const int q = tid + k * C::NTHREADS; if (q < NWQ) rw[k] = fetch_w(c0, q); }
        };
        auto regs_to_lds = [&]() {
#pragma unroll
            for (int k = 0; k < IN_PT; ++k) { const int q = tid + k * C::NTHREADS; if (q < NQ) *reinterpret_cast<float4*>(in_lds + q * 4) = rin[k]; }
#pragma unroll
            for (int k = 0; k < W_PT; ++k) { const int q = tid + k * C::NTHREADS; if (q < NWQ) *reinterpret_cast<float4*>(w_lds + q * 4) = rw[k]; }
        };
        if (c_begin < c_end) {
            fetch_regs(c_begin);
            regs_to_lds();
        }
        __syncthreads();
        for (int c0 = c_begin; c0 < c_end; c0 += C::CK) {
            const bool more = c0 + C::CK < c_end;
            if (more) fetch_regs(c0 + C::CK);
            __builtin_amdgcn_sched_barrier(0);      // keep the loads ahead of the MFMA stream
            compute();
            __syncthreads();                        // everyone is done reading this chunk
            if (more) {
                regs_to_lds();
                __syncthreads();
            }
        }
    } else {
        for (int c0 = c_begin; c0 < c_end; c0 += C::CK) {
            __syncthreads();   // everyone is done reading the previous chunk
            stage_direct(c0);
            __syncthreads();
            compute();
        }
    }

    // ---- epilogue: C/D layout col = lane&31 (voxel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel) ----
    const int co_base = co0 + wm * (C::MI * 32);
    if (p.gn_part) {
        // GroupNorm statistics of this tile's (acc + bias): a group of 8 channels is the rows (r>>2 fixed) of both lane halves,
        // a group of 4 channels the rows of one half -- fp32 over a lane's <= 4 * NI values, fp64 from there on: lanes (xor
        // butterfly, fixed order), the WN waves that share the channels (LDS, fixed order), then ONE slot of the global
        // [group][slot] table per tile.  No atomics anywhere: bit-identical run to run.
        __syncthreads();                                       // the staged tiles are dead: their LDS is reused below
        constexpr int GPW8 = C::MI * 4;                        // 8-channel groups per wave (x2 for 4-channel groups)
        double* red = reinterpret_cast<double*>(smem + C::NWAVES * 32 * 36);  // behind the epilogue's transpose buffers
        static_assert(C::NWAVES * 32 * 36 + 2 * C::WM * C::WN * GPW8 * 2 * 2 <= C::LDS_FLOATS, "GN partial sums must fit the staging LDS");
        const bool g4 = p.gn_cpg == 4;
        bool ok[C::NI];
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) {
            const int sN = wn * C::NI + ni;
            int y = y0 + seg_row(sN, l31), x = x0 + seg_col(sN, l31);
            if constexpr (C::FLAT) {                           // (statistics are only fused into per-plane flat launches: flat_t == 0)
                const int f = F0 + sN * 32 + l31, y1 = f / pitch, x1 = f - y1 * pitch;
                y = y1 - 1;
                x = (x1 >= 1) ? x1 - 1 : p.W;
            }
            ok[ni] = y < p.H && x < p.W && sN < C::NLIVE;
        }
#pragma unroll
        for (int mi = 0; mi < C::MI; ++mi) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float fs = 0.f, fss = 0.f;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int co = co_base + mi * 32 + rr + 8 * j + 4 * half;
                    const float bv = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
                    for (int ni = 0; ni < C::NI; ++ni)
                        if (ok[ni]) {
                            const float v = acc[mi][ni][4 * j + rr] + bv;
                            fs += v;
                            fss += v * v;
                        }
                }
                double ds = (double)fs, dss = (double)fss;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { ds += __shfl_xor(ds, o, 64); dss += __shfl_xor(dss, o, 64); }
                if (!g4) { ds += __shfl_xor(ds, 32, 64); dss += __shfl_xor(dss, 32, 64); }
                if (l31 == 0 && (g4 || half == 0)) {
                    const int gl = g4 ? (mi * 4 + j) * 2 + half : mi * 4 + j;          // group within this wave's channel block
                    const int gpw = g4 ? 2 * GPW8 : GPW8;
                    double* d = red + ((size_t)(wm * gpw + gl) * C::WN + wn) * 2;
                    d[0] = ds; d[1] = dss;
                }
            }
        }
        __syncthreads();
        {
            const int gpw = g4 ? 2 * GPW8 : GPW8, ngw = C::WM * gpw;           // groups of this workgroup's MT channels
            if (tid < ngw) {
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int k = 0; k < C::WN; ++k) { a += red[((size_t)tid * C::WN + k) * 2]; b += red[((size_t)tid * C::WN + k) * 2 + 1]; }
                const int gg = co0 / p.gn_cpg + tid;
                if (gg * p.gn_cpg < p.Cout) {
                    const int slot = p.gn_slot0 + (t * p.tiles_y + ty) * p.tiles_x + tx;
                    double* o = p.gn_part + (int64_t)blockIdx.y * p.gn_bs + ((size_t)gg * p.gn_cap + slot) * 2;
                    o[0] = a; o[1] = b;
                }
            }
        }
        __syncthreads();
    }
    if (p.vec_epi && !C::FLAT) {
        // 16-B stores: each wave transposes its 32x32 accumulator tiles through LDS so that a lane owns 4 consecutive
        // voxels of one channel (the MFMA layout gives it 16 channels of ONE voxel -> 4-B stores, 4x the instructions)
        __syncthreads();                                       // all waves are done with the staged tiles
        constexpr int TP = 36;                                 // padded row pitch (floats), keeps float4 reads aligned
        float* tl = smem + wave * (32 * TP);
        static_assert(C::NWAVES * 32 * 36 <= C::LDS_FLOATS, "epilogue transpose buffer must fit the staging LDS");
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) {
            const int s = wn * C::NI + ni;
            if (C::BLK && s >= C::NLIVE) continue;             // (the last wave's missing block)
            // a lane of the transposed tile owns positions c4 .. c4 + 3 of the column block, c4 = (lane & 7) * 4: four columns of one row in either block shape
            const int y = y0 + seg_row(s, (lane & 7) * 4);
            const int xs = x0 + seg_col(s, (lane & 7) * 4) - (lane & 7) * 4;            // so that xs + c4 is the lane's first column
            // residual and bias of two 32-channel sub-tiles at a time are requested up front: otherwise each of the loads
            // below is waited for on its own, right where it is used, and the epilogue of a short-K conv becomes a chain of
            // HBM latencies
            constexpr int MG = (C::MI >= 2 && C::MIN_WG < 3) ? 2 : 1;      // (3 workgroups per CU: 168 VGPRs, one sub-tile at a time)
#pragma unroll
            for (int m0 = 0; m0 < C::MI; m0 += MG) {
                float4 rres[MG][4];
                float rbias[MG][4];
#pragma unroll
                for (int mg = 0; mg < MG; ++mg)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = (lane >> 3) + 8 * j, c4 = (lane & 7) * 4;
                        const int co = co_base + (m0 + mg) * 32 + row;
                        const int x = xs + c4;
                        rres[mg][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                        rbias[mg][j] = 0.f;
                        if (y < p.H && x < p.W && co < p.Cout) {
                            if (p.res) rres[mg][j] = *reinterpret_cast<const float4*>(p.res + (int64_t)co * p.res_cs + (int64_t)t * p.res_ts + (int64_t)y * p.res_ys + x);
                            if (p.bias) rbias[mg][j] = p.bias[co];
                        }
                    }
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    const int mi = m0 + mg;
#pragma unroll
                    for (int r = 0; r < 16; ++r) tl[((r & 3) + 8 * (r >> 2) + 4 * half) * TP + l31] = acc[mi][ni][r];
                    // no barrier needed: the tile buffer is private to this wave (wave-synchronous LDS traffic)
                    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the ds_writes above have landed
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = (lane >> 3) + 8 * j, c4 = (lane & 7) * 4;
                        const int co = co_base + mi * 32 + row;
                        const int x = xs + c4;
                        if (y < p.H && x < p.W && co < p.Cout) {    // W % 4 == 0 is guaranteed by the launcher for this path
                            float4 v = *reinterpret_cast<const float4*>(tl + row * TP + c4);
                            const float bv = rbias[mg][j];
                            v.x += bv; v.y += bv; v.z += bv; v.w += bv;
                            const int64_t off = (int64_t)t * p.out_ts + (int64_t)y * p.out_ys + x;
                            const float4 rv = rres[mg][j];         // (acc + bias) + residual, as before
                            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                            if (p.relu) { v.x = relu_keep_nan(v.x); v.y = relu_keep_nan(v.y); v.z = relu_keep_nan(v.z); v.w = relu_keep_nan(v.w); }
                            *reinterpret_cast<float4*>(p.out + (int64_t)blockIdx.y * p.out_bs + (int64_t)blockIdx.z * p.out_split_stride + (int64_t)co * p.out_cs + off) = v;
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);            // reads done before the next tile overwrites the buffer
                }
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni) {
        const int s = wn * C::NI + ni;
        if (C::BLK && s >= C::NLIVE) continue;
        int y = y0 + seg_row(s, l31);
        int x = x0 + seg_col(s, l31);
        int te = t;                                            // frame of this position (flat_t: decoded from the flat index)
        if constexpr (C::FLAT) {                               // flat position -> (row, column) of the haloed plane -> output (y, x)
            int f = F0 + s * 32 + l31;
            if (p.flat_t) {                                    // the run crosses the frames: [T][H + 2][pitch]
                te = f / (int)p.in_ts;
                f -= te * (int)p.in_ts;
            }
            const int y1 = f / pitch, x1 = f - y1 * pitch;
            y = (y1 >= 1 && te < p.T_all) ? y1 - 1 : p.H;      // halo rows (and frames past the end): nothing to store
            x = (x1 >= 1) ? x1 - 1 : p.W;                      // halo columns: nothing to store
        }
        if (y < p.H && x < p.W) {
            int64_t off, roff = 0;
            if (p.dec_W > 0) {      // flat [C][V] launch: x is the voxel index, decode it for the destination layout
                const int hw = p.dec_H * p.dec_W;
                const int t2 = x / hw, r2 = x - t2 * hw, y2 = r2 / p.dec_W, x2 = r2 - y2 * p.dec_W;
                off = (int64_t)t2 * p.out_ts + (int64_t)y2 * p.out_ys + x2;
                roff = (int64_t)t2 * p.res_ts + (int64_t)y2 * p.res_ys + x2;
            } else {
                off = (int64_t)te * p.out_ts + (int64_t)y * p.out_ys + x;
                roff = (int64_t)te * p.res_ts + (int64_t)y * p.res_ys + x;
            }
            if constexpr (C::F16) {
                if (p.out_p16) {                                   // (uniform) pair planes in octets: lane half h holds channels 8 q + 4 h + 0..3 of a 32-row tile
                    const int64_t plane_words = (int64_t)(p.Cout / 8) * p.out_cs * 4;
#pragma unroll
                    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int co4 = co_base + mi * 32 + 8 * q + 4 * half;
                            if (co4 < p.Cout) {
                                float v[4];
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    v[k] = acc[mi][ni][4 * q + k] + (p.bias ? p.bias[co4 + k] : 0.f);
                                    if (p.relu) v[k] = relu_keep_nan(v[k]);
                                }
                                uint2 hw, lw;
                                split_pair_f16(v[0], v[1], hw.x, lw.x);
                                split_pair_f16(v[2], v[3], hw.y, lw.y);
                                unsigned int* d = p.out_p16 + ((int64_t)(co4 / 8) * p.out_cs + off) * 4 + 2 * half;
                                *reinterpret_cast<uint2*>(d) = hw;
                                *reinterpret_cast<uint2*>(d + plane_words) = lw;
                            }
                        }
                    continue;
                }
            }
            float* o = p.out + (int64_t)blockIdx.y * p.out_bs + (int64_t)blockIdx.z * p.out_split_stride + off;
#pragma unroll
            for (int mi = 0; mi < C::MI; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co_base + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < p.Cout) {
                        float v = acc[mi][ni][r] + (p.bias ? p.bias[co] : 0.f);
                        if (p.res) v += p.res[(int64_t)co * p.res_cs + roff];
                        if (p.relu) v = relu_keep_nan(v);
                        o[(int64_t)co * p.out_cs] = v;
                    }
                }
            }
        }
    }
}

// split-K epilogue: out[c,t,y,x] = act(bias[c] + res + sum_z partial[z][c][v])   (fixed order -> deterministic)
struct SplitReduceParams {
    const float* partial;
    const float* bias;
    const float* res;
    float* out;
    int64_t out_cs, out_ts, out_ys, res_cs, res_ts, res_ys, slab;
    int C, H, W, ksplit, relu;     // (H, W): how to split the flat voxel index into (t, y, x)
    int64_t V;
    unsigned per_c;                // threads needed per channel: V / (4 or 1)
    double* gn_part;               // GroupNorm partial sums of the reduced output (see ConvKParams), or NULL
    int gn_cpg, gn_cap, gn_slot0;
    int64_t part_bs, out_bs, gn_bs; // clip batch (blockIdx.z = clip): strides of the partial slabs, the output and the partial-sum table
};
// grid = (blocks per channel, channels).  One thread per output float4 (VEC: W % 4 == 0 and 16-B aligned output / residual
// rows) or per output float; 32-bit index math.  Partial slabs are dense [C][T][H][W], so their reads are coalesced 16-B
// loads in the VEC form.  With gn_part every block also leaves the (sum, sum of squares) of the values it wrote in its own
// slot (gn_slot0 + (c % cpg) * gridDim.x + blockIdx.x) of the channel's group -- fixed-order tree, no atomics.
template <bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const SplitReduceParams p) {
    constexpr int VW = VEC ? 4 : 1;
    const unsigned wq = (unsigned)p.W / VW;
    const unsigned c = blockIdx.y;
    const unsigned j = blockIdx.x * 256u + threadIdx.x;        // item within the channel
    const bool active = j < p.per_c;
    const float* const partial = p.partial + (int64_t)blockIdx.z * p.part_bs;
    float* const outp = p.out + (int64_t)blockIdx.z * p.out_bs;
    float s1 = 0.f, s2 = 0.f;
    if (active) {
        const unsigned r2 = j / wq, x = (j - r2 * wq) * VW;    // r2 = t * H + y
        const unsigned t = r2 / (unsigned)p.H, y = r2 - t * (unsigned)p.H;
        const int64_t i = (int64_t)c * p.V + (int64_t)r2 * p.W + x;
        const int64_t o = (int64_t)c * p.out_cs + (int64_t)t * p.out_ts + (int64_t)y * p.out_ys + x;
        if constexpr (VEC) {
            float4 acc = *reinterpret_cast<const float4*>(partial + i);
            for (int z = 1; z < p.ksplit; ++z) {
                const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)z * p.slab + i);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            if (p.bias) { const float b = p.bias[c]; acc.x += b; acc.y += b; acc.z += b; acc.w += b; }
            if (p.res) {
                const float4 r = *reinterpret_cast<const float4*>(p.res + (int64_t)c * p.res_cs + (int64_t)t * p.res_ts + (int64_t)y * p.res_ys + x);
                acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
            }
            if (p.relu) { acc.x = relu_keep_nan(acc.x); acc.y = relu_keep_nan(acc.y); acc.z = relu_keep_nan(acc.z); acc.w = relu_keep_nan(acc.w); }
            *reinterpret_cast<float4*>(outp + o) = acc;
            s1 = (acc.x + acc.y) + (acc.z + acc.w);
            s2 = (acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w);
        } else {
            float acc = partial[i];
            for (int z = 1; z < p.ksplit; ++z) acc += partial[(int64_t)z * p.slab + i];
            if (p.bias) acc += p.bias[c];
            if (p.res) acc += p.res[(int64_t)c * p.res_cs + (int64_t)t * p.res_ts + (int64_t)y * p.res_ys + x];
            if (p.relu) acc = relu_keep_nan(acc);
            outp[o] = acc;
            s1 = acc;
            s2 = acc * acc;
        }
    }
    if (p.gn_part) {                                           // (uniform)
        __shared__ double red[2][4];
        double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d1; red[1][threadIdx.x >> 6] = d2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int g = (int)c / p.gn_cpg, slot = p.gn_slot0 + ((int)c % p.gn_cpg) * (int)gridDim.x + (int)blockIdx.x;
            double* o = p.gn_part + (int64_t)blockIdx.z * p.gn_bs + ((size_t)g * p.gn_cap + slot) * 2;
            o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
}

// [Cout][Cin][taps] -> [Cin/4][taps][4][Cout]
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int taps) {
    const int64_t n = (int64_t)Cout * Cin * taps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        int64_t r = i / Cout;
        const int c4 = (int)(r & 3);
        r >>= 2;
        const int tap = (int)(r % taps);
        const int chunk = (int)(r / taps);
        const int ci = chunk * 4 + c4;
        packed[i] = w[((int64_t)co * Cin + ci) * taps + tap];
    }
}

// bf16x6 packing: [Cout][Cin][taps] fp32 -> per chunk [G][hi|mid|lo][half][Cout][8 bf16]; element j of (grp, half) is channel
// cg*2*CPH + half*CPH + j % CPH of the chunk, tap tg*TPG + j / CPH (zero beyond the last tap / channel); group-major so that a
// weight PHASE -- a run of k-groups -- is one contiguous LDS image
__global__ void pack_conv_weight_bf16x6_kernel(const float* __restrict__ w, uint4* __restrict__ packed, int Cout, int Cin, int taps,
                                                int CK, int TPG) {
    const int CPH = 8 / TPG, NTG = (taps + TPG - 1) / TPG, NCG = CK / (2 * CPH), G = NTG * NCG;
    const int nchunks = (Cin + CK - 1) / CK;
    const int64_t n = (int64_t)nchunks * G * 3 * 2 * Cout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        int64_t r = i / Cout;
        const int h = (int)(r & 1);
        r >>= 1;
        const int pl = (int)(r % 3);
        r /= 3;
        const int grp = (int)(r % G);
        const int chunk = (int)(r / G);
        const int cg = grp / NTG, tg = grp % NTG;
        unsigned short v[8];
        for (int j = 0; j < 8; ++j) {
            const int tapi = j / CPH, chl = j % CPH;
            const int tap = tg * TPG + tapi, ci = chunk * CK + cg * 2 * CPH + h * CPH + chl;
            float x = 0.f;
            if (tap < taps && ci < Cin) x = w[((int64_t)co * Cin + ci) * taps + tap];
            const __bf16 hi = (__bf16)x;
            const float r1 = x - (float)hi;
            const __bf16 mid = (__bf16)r1;
            const __bf16 lo = (__bf16)(r1 - (float)mid);
            const __bf16 pick = pl == 0 ? hi : (pl == 1 ? mid : lo);
            v[j] = *reinterpret_cast<const unsigned short*>(&pick);
        }
        uint4 o;
        o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
        o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
        packed[i] = o;
    }
}

// f16x3 packing: the bf16x6 slab order with fp16 planes [G][hi|lo|hi * 2^-11][half][Cout][8 fp16] of w * S[co], S[co] = 2^(13 -
// floor(log2(max|w[co]|))) PER OUTPUT CHANNEL -- each channel's largest weight lands in [2^13, 2^14), so all three terms of every
// weight within 2^-16 of its channel's largest are normal fp16 numbers (the third plane multiplies the input tile's lo * 2^11 term).
// (A scale per LAYER, as in round 3, loses bits on every channel whose weights sit far below the layer's largest: FrozenBN folded
// with eps = 0 multiplies each output channel by gamma / sqrt(var), make_layers.py:51-63, which spans orders of magnitude in
// trained checkpoints.)  Behind the last slab: float inv[Cout] = 1 / (S[co] * activation scale) -- the conv kernel multiplies its
// accumulator rows by it -- then uint32 max_bits[Cout] (bits of max|w[co]|, pack-time scratch).
__global__ __launch_bounds__(256) void absmax_rows_kernel(const float* __restrict__ w, int64_t row_len, unsigned int* __restrict__ out) {
    __shared__ unsigned int red[4];
    const float* r = w + (int64_t)blockIdx.x * row_len;
    unsigned int m = 0;
    for (int64_t i = threadIdx.x; i < row_len; i += 256) m = max(m, __float_as_uint(fabsf(r[i])));   // (non-negative floats order like their bit patterns; NaN / inf end up on top)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}

__device__ __forceinline__ float f16x3_weight_scale(unsigned int max_bits) {
    const int e = (int)((max_bits >> 23) & 0xff);               // biased exponent of max|w| (0: zero / subnormal weights only -> scale 1)
    if (e == 0 || e == 0xff) return 1.0f;
    return __uint_as_float((unsigned int)min(max(127 + 13 - (e - 127), 1), 254) << 23);
}

__global__ void pack_conv_weight_f16x3_kernel(const float* __restrict__ w, uint4* __restrict__ packed, int Cout, int Cin, int taps,
                                               int CK, int TPG) {
    const int CPH = 8 / TPG, NTG = (taps + TPG - 1) / TPG, NCG = CK / (2 * CPH), G = NTG * NCG;
    const int nchunks = (Cin + CK - 1) / CK;
    constexpr int NPLW = SS_F16_WPLANES;
    const int64_t n = (int64_t)nchunks * G * NPLW * 2 * Cout;
    float* inv = reinterpret_cast<float*>(packed + n);
    const unsigned int* max_bits = reinterpret_cast<const unsigned int*>(inv + Cout);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const float S = f16x3_weight_scale(max_bits[co]);
        if (i < Cout) inv[co] = 1.0f / (S * STEMSEG_F16X3_ACT_SCALE);
        int64_t r = i / Cout;
        const int h = (int)(r & 1);
        r >>= 1;
        const int pl = (int)(r % NPLW);
        r /= NPLW;
        const int grp = (int)(r % G);
        const int chunk = (int)(r / G);
        const int cg = grp / NTG, tg = grp % NTG;
        unsigned short v[8];
        for (int j = 0; j < 8; ++j) {
            const int tapi = j / CPH, chl = j % CPH;
            const int tap = tg * TPG + tapi, ci = chunk * CK + cg * 2 * CPH + h * CPH + chl;
            float x = 0.f;
            if (tap < taps && ci < Cin) x = w[((int64_t)co * Cin + ci) * taps + tap] * S;
            const _Float16 hi = (_Float16)x;
            const _Float16 lo = (_Float16)(x - (float)hi);
            const _Float16 his = (_Float16)((float)hi * (1.0f / 2048.0f));
            const _Float16 pick = pl == 0 ? hi : (pl == 1 ? lo : his);
            v[j] = *reinterpret_cast<const unsigned short*>(&pick);
        }
        uint4 o;
        o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
        o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
        packed[i] = o;
    }
}

// tile shapes --------------------------------------------------------------------------------------
//                       KT KH KW  CK  MI NI WM WN COLS
// fp32-input MFMA mode (STEMSEG_PRECISION_F32)
using K3Big = ConvCfg<3, 3, 3, 4, 4, 2, 1, 4, 1>;          // 128 co x (8 rows x 32 cols): the big tile's generic form (unaligned volumes, Cout % 128 != 0)
using K3BigGL = ConvCfg<3, 3, 3, 2, 4, 2, 1, 4, 1, false, 0, true, 0, true>;   // same tile, 2-channel chunks, two LDS buffers filled by LDS-DMA
using K3Med = ConvCfg<3, 3, 3, 4, 2, 2, 2, 2, 1, true>;    // 128 co x (4 rows x 32 cols)
using K3Small = ConvCfg<3, 3, 3, 4, 2, 1, 2, 2, 1, true>;  // 128 co x (2 rows x 32 cols)
using K1Big = ConvCfg<1, 1, 1, 32, 4, 2, 1, 4, 8, true>;   // 128 co x 256 voxels
using K1Small = ConvCfg<1, 1, 1, 32, 2, 2, 2, 2, 4, true>; // 128 co x 128 voxels
using K1M64 = ConvCfg<1, 1, 1, 32, 2, 2, 1, 4, 8, true>;   //  64 co x 256 voxels
using K1BigGL = ConvCfg<1, 1, 1, 16, 4, 2, 1, 4, 8, false, 0, true, 0, true>;     // direct-to-LDS twins of the 1x1 tiles (half the chunk: two buffers in the same LDS)
using K1SmallGL = ConvCfg<1, 1, 1, 16, 2, 2, 2, 2, 4, false, 0, true, 0, true>;
using K1M64GL = ConvCfg<1, 1, 1, 16, 2, 2, 1, 4, 8, false, 0, true, 0, true>;
// 2-D convolutions of the encoder: the frames of a clip are the T axis, KT = 1
using K2Big = ConvCfg<1, 3, 3, 8, 4, 2, 1, 4, 1, true>;    // 128 co x (8 rows x 32 cols)
using K2Med = ConvCfg<1, 3, 3, 8, 2, 2, 2, 2, 1, true>;    // 128 co x (4 rows x 32 cols)
using K2Small = ConvCfg<1, 3, 3, 8, 2, 1, 2, 2, 1, true>;  // 128 co x (2 rows x 32 cols)
using K2M64 = ConvCfg<1, 3, 3, 8, 2, 2, 1, 4, 1, true>;    //  64 co x (8 rows x 32 cols)
// flat-tile forms for maps whose width wastes >= 10 % of a 32-column tile (8x / 16x maps: pitch 112 / 56; the staged run is
// tile + 2 * PMAX + 8 floats per channel plane, so a tighter PMAX stages less)
using K2FlatBig = ConvCfg<1, 3, 3, 8, 4, 2, 1, 4, 8, true, 0, false, 112>;     // 128 co x 256 flat positions
using K2FlatMed = ConvCfg<1, 3, 3, 8, 2, 2, 2, 2, 4, true, 0, false, 112>;     // 128 co x 128
using K3FlatMed = ConvCfg<3, 3, 3, 4, 2, 2, 2, 2, 4, true, 0, false, 112>;     // 128 co x 128
using K3FlatSmall = ConvCfg<3, 3, 3, 4, 2, 1, 2, 2, 2, true, 0, false, 112>;   // 128 co x 64
using K2FlatBig56 = ConvCfg<1, 3, 3, 8, 4, 2, 1, 4, 8, true, 0, false, 56>;
using K2FlatMed56 = ConvCfg<1, 3, 3, 8, 2, 2, 2, 2, 4, true, 0, false, 56>;
using K3FlatMed56 = ConvCfg<3, 3, 3, 4, 2, 2, 2, 2, 4, true, 0, false, 56>;
using K3FlatSmall56 = ConvCfg<3, 3, 3, 4, 2, 1, 2, 2, 2, true, 0, false, 56>;

// Split-staged tiles (BFV 2 = bf16x6, 3 = f16x3).  3x3x3: eight waves share one weight slab (86 KB in bf16x6, 57 KB in f16x3; one
// workgroup per CU, two waves per SIMD); 2-D and 1x1 tiles keep four-wave shapes (two workgroups per CU) except the big ones
template <int BFV>
struct SplitTiles {
    // 1x3x3 chunks: f16x3 takes 16 channels -- a k-group is then ONE tap x 16 channels and the nine taps fill nine groups exactly; with 8
    // channels a group is two taps x 8 channels and the ninth tap drags a zero tap along: 10 % of the class's MFMAs multiplied zeros
    // (the chip is power-bound under these streams, so MFMAs not issued are time).  bf16x6's three planes of 16 channels do not fit the LDS.
    static constexpr int CK2 = BFV == 3 ? 16 : 8;
    using Y3Big = ConvCfg<3, 3, 3, 4, 4, 2, 1, 8, 1, false, BFV>;   // 128 co x (16 rows x 32 cols), 512 threads
    using Y3Med = ConvCfg<3, 3, 3, 4, 2, 2, 2, 4, 1, false, BFV>;   // 128 co x ( 8 rows x 32 cols), 512 threads
    using Y3Small = ConvCfg<3, 3, 3, 4, 2, 1, 2, 4, 1, false, BFV>; // 128 co x ( 4 rows x 32 cols), 512 threads
    using Y2Big = ConvCfg<1, 3, 3, CK2, 4, 2, 1, 8, 1, false, BFV>;   // 128 co x (16 rows x 32 cols), 512 threads (the weight prefetch of a four-wave tile spills)
    using Y2Med = ConvCfg<1, 3, 3, CK2, 2, 2, 2, 2, 1, false, BFV>;   // 128 co x (4 rows x 32 cols)
    using Y2Small = ConvCfg<1, 3, 3, CK2, 2, 1, 2, 2, 1, false, BFV>; // 128 co x (2 rows x 32 cols)
    using Y2M64 = ConvCfg<1, 3, 3, CK2, 2, 2, 1, 4, 1, false, BFV>;   //  64 co x (8 rows x 32 cols)
    using Y1Big = ConvCfg<1, 1, 1, 32, 4, 2, 1, 4, 8, false, BFV>;  // 128 co x 256 voxels
    using Y1Small = ConvCfg<1, 1, 1, 32, 2, 2, 2, 2, 4, false, BFV>; // 128 co x 128 voxels
    using Y1M64 = ConvCfg<1, 1, 1, 32, 2, 2, 1, 4, 8, false, BFV>;  //  64 co x 256 voxels
    using Y1Wide = ConvCfg<1, 1, 1, 32, 4, 2, 2, 4, 8, false, BFV>; // 256 co x 256 voxels, 512 threads: an input element is split once per 256 output channels
    // flat (ragged-width) form of the big 2-D tile: 128 co x 512 flat positions of the whole [T][H + 2][pitch] run (across the
    // frames); PMAX = largest row pitch served (the staged run is tile + 2 PMAX + 8 words)
    template <int PMAX> using Y2Flat = ConvCfg<1, 3, 3, CK2, 4, 2, 1, 8, 16, false, BFV, false, PMAX>;
    // 1x4x4 taps on 16-channel chunks (16 k-groups of one tap x 16 channels): the 7x7 stride-2 stem as a stride-1 4x4 convolution over the
    // space-to-depth image (encoder.hip); 64 co x (16 rows x 32 cols), 512 threads
    using Y4Stem = ConvCfg<1, 4, 4, 16, 2, 2, 1, 8, 1, false, BFV>;
    // block tiles (ConvCfg::BLK): 128 co x (20 rows x 24 columns) as 5 x 3 column blocks of 4 rows x 8 columns, 512 threads -- no junk position on
    // 120 x 216 maps (30 x 27 blocks), where the 16 x 32 tiles compute 128 x 224
    using Y3Blk = ConvCfg<3, 3, 3, 4, 4, 2, 1, 8, 3, false, BFV, false, 0, false, 5>;
    using Y2Blk = ConvCfg<1, 3, 3, CK2, 4, 2, 1, 8, 3, false, BFV, false, 0, false, 5>;
    // the medium 3x3x3 tile's block form: 128 co x (4 rows x 56 columns) as 1 x 7 column blocks (the wave pair that would own the eighth block runs
    // one) -- the decoders' 60 x 108 and 30 x 54 maps compute 60 x 112 / 32 x 56 positions instead of the 8-row x 32-column tile's 64 x 128 / 32 x 64
    // (21 % junk -> 3.6 % / 9.6 %): under the power limit MFMAs not issued are time even where the round count stays (DESIGN.md sections 5f, 5h)
    using Y3Blk7 = ConvCfg<3, 3, 3, 4, 2, 2, 2, 4, 7, false, BFV, false, 0, false, 1>;
    // Measured in round 5 and not kept (profiles/r05b_conv_sweep_f16x3_T32.txt, tile_cfg 6 / 7 / 8 of that build): four-wave halves of the
    // eight-wave tiles, two per CU so that one's staging phases run under the other's MFMA stream -- block_4x 1 076 vs 1 006 us, layer-3 3x3
    // 214 vs 203, 1024 -> 256 110 vs 107, 256 -> 1024 + residual 170 vs 154; and 64-channel chunks for the 1x1 tiles (half the chunk
    // boundaries) -- 1024 -> 256 106 vs 107 (256-co tile), 134 vs 125 (128-co tile), 256 -> 1024 190 vs 154.  Neither barrier overlap
    // nor chunk length is what these kernels wait for.  Likewise the input tile fetched TWO chunks ahead through a second register set
    // (profiles/r05h_sweep{,_ina}.txt): the 128 x 128 tile drops from three to two waves per SIMD (174 VGPRs) and loses 18 % (256 -> 1024
    // 182 vs 154 us), the eight-wave 256 x 256 tile is unchanged (111 vs 108): the step 96.4 vs 98.6 clips/s.  Removed.
    // And PING-PONG forms of the eight-wave tiles (two LDS buffers; waves 0-3 run the MFMA stream of chunk i while waves 4-7 -- their SIMD
    // partners -- split their share of chunk i + 1 into the other buffer, roles swapped at every barrier; B fragments a k-group ahead):
    // bit-identical results, and the same times -- 3x3 tiles within +-1 % (fpn_layer1 2 790 vs 2 796 us, layer-3 3x3 246 vs 239), the 256-co
    // 1x1 tile 5-9 % ahead on the long-K reductions (1024 -> 256 112 vs 119 us) = 0.7 % of the step; the 3x3x3 tile does not fit twice into
    // 160 KB (profiles/r05i_pingpong_tiles.txt).  Why no schedule moves these kernels (profiles/r05i_dvfs_zero_inputs.txt): the SAME launches
    // on all-zero activations -- same instructions, less switching -- run 20-25 % faster (block_4x 995 -> 832 us = 441 TF-eq, fpn_layer1
    // 2 790 -> 2 247): on real data the chip sits at its POWER limit (1.9-2.0 GHz effective under these MFMA streams, GRBM_GUI_ACTIVE / time),
    // and cycles saved by a better schedule come back as a lower clock.  What is left to win is work not done: MFMAs on tap padding (10 tap
    // slots for 9 taps in the 1x3x3 class), junk positions of ragged maps, re-split inputs.
};

// sustained per-CU rate while the chip is full, for the row planner's cost model (measured: ~0.75 of the 157.3 / 256 TFLOP/s fp32
// MFMA peak)
constexpr double CU_FLOPS_F32 = 0.46e12;

// A launch DECIDES (tile shape, split-K factor, row cut) on its planning shape and RUNS on its real one.  The two are the same
// for a caller that gives no planning frames (the decoders: a clip is a clip).  The encoder's launches hold as many frames as the
// caller batched -- one clip, four clips of a step, the union of eight overlapping windows -- and the K-partition of a split-K
// launch is a summation order: decided on the real shape, the last bits of every embedding would depend on the batch, and a
// world-N job would not reproduce the world-1 labels (clusterers.py:106-146 is a chain of hard thresholds).  So the encoder
// plans every layer for `plan_frames` frames of its per-frame shape, whatever the launch holds.
struct PlanCtx {
    ConvKParams shape;           // the planning twin of the launch's parameters (T, or the flat voxel count, scaled to plan_frames)
    int64_t scratch_floats;      // split-K scratch the plan may count on (the real scratch must then hold real / plan times that)
};

// workgroups tile shape C makes of shape d (flat_t: the flat run crosses the frames)
template <class C>
static int64_t cfg_workgroups(const ConvKParams& d, int flat_t) {
    int64_t tx = C::FLAT ? ceil_div((int64_t)d.H * d.in_ys, C::NT) : ceil_div(d.W, C::TW);
    const int64_t ty = C::FLAT ? 1 : ceil_div(d.H, C::ROWS);
    int64_t T = d.T;
    if (C::FLAT && flat_t) { tx = ceil_div((int64_t)d.T * d.in_ts, C::NT); T = 1; }
    return tx * ty * T * ceil_div(d.Cout, C::MT);
}

template <class C>
static int launch_cfg(ConvKParams p, hipStream_t s, float* scratch, int64_t scratch_floats, int force_ksplit = 0, const PlanCtx* plan = nullptr) {
    p.tiles_x = C::FLAT ? (int)ceil_div((int64_t)p.H * p.in_ys, C::NT) : (int)ceil_div(p.W, C::TW);
    p.tiles_y = C::FLAT ? 1 : (int)ceil_div(p.H, C::ROWS);
    p.T_all = p.T;
    const int flat_t = (C::FLAT && p.flat_t) ? 1 : 0;
    if (flat_t) {                                        // one run of flat positions over all frames: [T][H + 2][pitch]
        p.tiles_x = (int)ceil_div((int64_t)p.T * p.in_ts, C::NT);
        p.T = 1;
    }
    p.flat_t = flat_t;
    const int out_vec = p.vec_epi;                       // true output rows are 16-B aligned (decides the reduce kernel's form)
    if (C::FLAT) p.vec_epi = 0;
    // split-K over the input-channel chunks when the layer alone cannot fill the chip: split until the PLANNED launch holds ~2.5
    // workgroups per CU for the tiles that share a CU, ~1.25 for the eight-wave split-staged tiles that own one (256 workgroups of
    // those already fill the chip: splitting them only adds slab traffic -- measured, L3 conv2)
    const int nchunks = (int)ceil_div(p.Cin, C::CK);
    const int64_t slab = (int64_t)p.Cout * p.T_all * p.H * p.W;
    int ksplit = 1;
    {
        const ConvKParams& d = plan ? plan->shape : p;
        const int64_t d_T = plan ? d.T : p.T_all;
        ConvKParams dd = d;
        dd.T = (int)d_T;
        const int64_t wgs = cfg_workgroups<C>(dd, flat_t);
        const int64_t slab_plan = (int64_t)d.Cout * d_T * d.H * d.W;
        const int64_t plan_scratch = plan ? plan->scratch_floats : scratch_floats / p.nb;      // (a clip batch decides on one clip's share)
        const int64_t wg_target = (C::X6 && C::NWAVES >= 8) ? 320 : 640;
        while (scratch && ksplit * 2 <= 16 && ksplit * 2 <= nchunks && wgs * ksplit * 2 <= wg_target && slab_plan * ksplit * 2 <= plan_scratch) ksplit *= 2;
        // A launch asked for pair-plane output feeds a fused bottleneck tail, which only an UN-split launch can (bottleneck_fused.hip).  From half a
        // round of workgroups on, splitting buys this launch nothing a one-round launch does not have (144 workgroups over the whole K take what
        // 288 take over half of it on 256 CUs, minus the slab reduce), and costs the block its fused tail: 384 x 640 maps, layer 3 -- 167 us of
        // conv3 + conv1 launches where the fused kernel runs one 94 %-full round.  Decided, like the split, on the planning shape.
        if (p.out_p16 && C::F16 && wgs >= 128) ksplit = 1;
    }
    if (force_ksplit > 0) ksplit = (scratch && (int64_t)p.nb * force_ksplit * slab <= scratch_floats) ? std::min(force_ksplit, nchunks) : 1;
    p.chunks_per_split = (int)ceil_div(nchunks, ksplit);
    ksplit = (int)ceil_div(nchunks, p.chunks_per_split);
    // (a planned launch never shrinks its K-partition to fit: that would be a batch-dependent summation order again)
    SS_CHECK_ARG(ksplit == 1 || (int64_t)p.nb * ksplit * slab <= scratch_floats, "conv3d: split-K scratch too small (%lld floats needed, %lld given)",
                 (long long)((int64_t)p.nb * ksplit * slab), (long long)scratch_floats);
    if (p.out_p16) {                                     // pair-plane output: by-element epilogue of an un-split launch, else plain fp32
        if (ksplit > 1 || !C::F16) p.out_p16 = nullptr;
        else { p.vec_epi = 0; if (p.gn_used_host) *p.gn_used_host = 1; }
    }
    SplitReduceParams rp;
    // 16-B reduce: the true output (and residual) rows are aligned (vec_epi as computed by the caller) and W % 4 == 0
    const bool rp_vec = out_vec && p.dec_W == 0 && (reinterpret_cast<uintptr_t>(scratch) % 16 == 0) && (slab % 4 == 0);
    if (ksplit > 1) {
        rp.partial = scratch; rp.bias = p.bias; rp.res = p.res; rp.out = p.out;
        rp.out_cs = p.out_cs; rp.out_ts = p.out_ts; rp.out_ys = p.out_ys; rp.res_cs = p.res_cs; rp.res_ts = p.res_ts; rp.res_ys = p.res_ys;
        rp.slab = slab; rp.C = p.Cout; rp.V = (int64_t)p.T_all * p.H * p.W; rp.ksplit = ksplit; rp.relu = p.relu;
        rp.H = p.dec_W > 0 ? p.dec_H : p.H; rp.W = p.dec_W > 0 ? p.dec_W : p.W;
        rp.gn_part = nullptr; rp.gn_cpg = rp.gn_cap = rp.gn_slot0 = 0;
        rp.part_bs = (int64_t)ksplit * slab; rp.out_bs = p.out_bs; rp.gn_bs = p.gn_bs;
        p.out_bs = (int64_t)ksplit * slab;                  // (each clip's slabs behind the previous clip's)
        // partial slabs are dense in the launch's own tile coordinates; the whole epilogue moves to the reduce kernel
        p.out = scratch; p.bias = nullptr; p.res = nullptr; p.relu = 0; p.dec_H = p.dec_W = 0;
        p.out_cs = (int64_t)p.T_all * p.H * p.W; p.out_ts = (int64_t)p.H * p.W; p.out_ys = p.W;
        p.out_split_stride = slab;
        p.vec_epi = !C::FLAT && (p.W % 4 == 0) && (reinterpret_cast<uintptr_t>(scratch) % 16 == 0);
    } else p.out_split_stride = 0;
    p.n_co = (int)ceil_div(p.Cout, C::MT);
    const int64_t red_items = slab / p.Cout / (rp_vec ? 4 : 1);      // reduce threads per channel
    const unsigned red_bx = (unsigned)ceil_div(red_items, 256);
    if (p.gn_part) {                                     // slots of this launch in the [group][slot] table (see ConvKParams)
        const int slot0 = *p.gn_used_host;
        const int64_t slots = ksplit > 1 ? (int64_t)p.gn_cpg * red_bx : (int64_t)p.tiles_x * p.tiles_y * p.T;
        SS_CHECK_ARG(slot0 + slots <= p.gn_cap, "conv3d: GroupNorm partial table too small (%lld slots needed, %d available)",
                     (long long)(slot0 + slots), p.gn_cap);
        if (ksplit > 1) { rp.gn_part = p.gn_part; rp.gn_cpg = p.gn_cpg; rp.gn_cap = p.gn_cap; rp.gn_slot0 = slot0; p.gn_part = nullptr; }
        else p.gn_slot0 = slot0;
        *p.gn_used_host = slot0 + (int)slots;
    }
    dim3 grid((unsigned)((int64_t)p.tiles_x * p.tiles_y * p.T * p.n_co), (unsigned)p.nb, (unsigned)ksplit);
    const double flops = 2.0 * p.Cin * C::TAPS * (double)p.Cout * p.T_all * p.H * p.W * p.nb;
    constexpr int tile_rows = (C::FLAT || C::BLK) ? C::NSEG : C::ROWS;      // flat / block tiles count under the 2-D tile of the same size
    const int tag = C::TAPS == 16 ? -1 : C::TAPS == 1 ? 10 + C::NSEG : (C::KT == 1 ? 20 + tile_rows : (tile_rows == 16 ? 9 : tile_rows));   // (4x4 taps = the stem: its caller's own tag)   // 9/8/4/2: 3x3x3, 18/14: 1x1x1, 28/24/22: 1x3x3
    void* ev = profile_begin(tag, flops, s);
    hipLaunchKernelGGL(conv_igemm_kernel<C>, grid, dim3(C::NTHREADS), 0, s, p);
    if (ksplit > 1) {
        SS_CHECK_ARG(red_items < (1ll << 32) - 256 && p.Cout <= 65535, "conv3d: split-K output too large (%lld elements)", (long long)slab);
        rp.per_c = (unsigned)red_items;
        if (rp_vec) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(red_bx, (unsigned)p.Cout, (unsigned)p.nb), dim3(256), 0, s, rp);
        else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(red_bx, (unsigned)p.Cout, (unsigned)p.nb), dim3(256), 0, s, rp);
    }
    profile_end(ev, s);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

// Tail balancing for launches of a few big workgroups per CU (fp32-input mode).  An MFMA-bound CU works through its workgroups
// at a fixed rate, so a launch costs ceil(workgroups / 256) "units"; 840 workgroups (block_4x at 480p) cost 4 units for 3.28
// units of work.  Here the output rows are cut in two: the first nA row-tiles run whole (a multiple of 256 workgroups, or
// close), the remaining rows run split-K by k with the deterministic slab reduce, which cuts their units into k-ths:
// 504 + 3 x 336 workgroups cost 2 + 4/3 = 3.33 units.  The split is chosen by a cost model (units x time per unit +
// slab traffic) and only taken when it beats the plain launch by > 3 %.  Sub-launches are ordinary launches on row
// sub-volumes (pointer offsets), so results are bit-identical to the plain launch wherever k = 1 and equal to a plain
// split-K launch elsewhere (fixed summation order).  The cut is in rows of the FRAME, so it does not depend on the frame count
// of the launch once it is decided on the planning shape.
struct RowPlan { double cost; int nA, k; };

// Cost of the best (whole rows, split-K rows) cut for tile shape C on shape d: rounds x time per round + slab traffic.  `occ`
// workgroups share a CU (LDS-limited), so the chip has 256 * occ slots and one round of co-resident workgroups takes
// occ x (flops per workgroup / sustained per-CU rate x eff); eff = relative MFMA efficiency of the tile shape
// (measured with tools/conv_sweep.py: 4-row tiles reach 0.92 of the 8-row tile's rate).
template <class C>
static RowPlan plan_rows(const ConvKParams& d, bool have_scratch, int64_t scratch_floats, double cu_flops, int occ, double eff) {
    const int64_t slots = 256 * occ;
    const int tiles_x = (int)ceil_div(d.W, C::TW);
    const int n = (int)ceil_div(d.H, C::ROWS);
    const int64_t c = (int64_t)tiles_x * d.T * ceil_div(d.Cout, C::MT);
    const int nchunks = (int)ceil_div(d.Cin, C::CK);
    const double t_round = occ * 2.0 * C::MT * ((double)C::ROWS * C::TW) * d.Cin * C::TAPS / (cu_flops * eff);
    RowPlan best{(double)ceil_div(n * c, slots) * t_round, n, 1};
    const double t_plain = best.cost;
    static const int ks[] = {2, 3, 4, 6, 8};
    for (int nA = 0; nA < n && have_scratch; ++nA) {
        const int64_t slabB = (int64_t)d.Cout * d.T * (d.H - (int64_t)nA * C::ROWS) * d.W;
        for (int k : ks) {
            if (k > nchunks || k * slabB > scratch_floats) continue;
            const double rounds = (double)ceil_div(nA * c, slots) + (double)ceil_div((n - nA) * c * k, slots) / k;
            const double t = rounds * t_round + 2.0 * k * slabB * 4.0 / 3.0e12 + 8e-6;
            if (t < best.cost) best = RowPlan{t, nA, k};
        }
    }
    if (best.cost > 0.97 * t_plain) best = RowPlan{t_plain, n, 1};      // not worth the extra launches
    return best;
}

template <class C>
static int launch_rows(const ConvKParams& p0, hipStream_t s, float* scratch, int64_t scratch_floats, const RowPlan& plan, const PlanCtx* pc) {
    if (plan.k <= 1) return launch_cfg<C>(p0, s, scratch, scratch_floats, 0, pc);
    auto rows = [&](int r0, int r1) {
        ConvKParams q = p0;
        q.in += (int64_t)r0 * p0.in_ys; q.in_limit -= (int64_t)r0 * p0.in_ys; q.in_H = (r1 - r0) + C::KH - 1;
        q.out += (int64_t)r0 * p0.out_ys;
        if (q.res) q.res += (int64_t)r0 * p0.res_ys;
        q.H = r1 - r0;
        return q;
    };
    const int rA = plan.nA * C::ROWS;
    if (rA > 0) {
        const int rc = launch_cfg<C>(rows(0, rA), s, nullptr, 0);
        if (rc) return rc;
    }
    const ConvKParams qB = rows(rA, p0.H);
    SS_CHECK_ARG((int64_t)p0.nb * plan.k * qB.Cout * qB.T * qB.H * qB.W <= scratch_floats, "conv3d: split-K scratch too small for the planned row cut");
    return launch_cfg<C>(qB, s, scratch, scratch_floats, plan.k);
}

// big launches (>= 512 workgroups of the 8-row tile): 8-row tile vs 4-row tile, each with its best row cut
template <class Big, class Med>
static int launch_planned(const ConvKParams& p, const ConvKParams& d, const PlanCtx* pc, hipStream_t s, float* scratch, int64_t scratch_floats, int occ_big, int occ_med) {
    const int64_t plan_scratch = pc ? pc->scratch_floats : scratch_floats / p.nb;
    const RowPlan a = plan_rows<Big>(d, scratch != nullptr, plan_scratch, CU_FLOPS_F32, occ_big, 1.0);
    const RowPlan b = plan_rows<Med>(d, scratch != nullptr, plan_scratch, CU_FLOPS_F32, occ_med, 0.92);
    if (b.cost < 0.97 * a.cost) return launch_rows<Med>(p, s, scratch, scratch_floats, b, pc);
    return launch_rows<Big>(p, s, scratch, scratch_floats, a, pc);
}

// fraction of the computed N positions that are real outputs, 2-D tile vs flat tile (per frame: no dependence on T)
template <class C>
static double tile_efficiency(const ConvKParams& p) {
    if (C::FLAT) return (double)p.H * p.W / ((double)C::NT * ceil_div((int64_t)p.H * p.in_ys, C::NT));
    return (double)p.H * p.W / ((double)C::ROWS * ceil_div(p.H, C::ROWS) * (double)C::TW * ceil_div(p.W, C::TW));
}
template <class Flat, class Tile2D>
static bool prefer_flat(const ConvKParams& p, int tile_cfg) {
    if (!p.vec4 || tile_cfg > 0 || p.dec_W > 0 || p.in_ys > Flat::PMAX || p.in_ys % 4 != 0 || p.W + 2 > p.in_ys) return false;
    return tile_efficiency<Flat>(p) > 1.08 * tile_efficiency<Tile2D>(p);
}

// GL twin when the launch meets its contract, else the register-staged form
template <class GLCfg, class Cfg>
static int launch_gl(const ConvKParams& p, hipStream_t s, float* scratch, int64_t scratch_floats, const PlanCtx* pc) {
    if (p.vec4 && p.Cin % GLCfg::CK == 0 && p.Cout % GLCfg::MT == 0) return launch_cfg<GLCfg>(p, s, scratch, scratch_floats, 0, pc);
    return launch_cfg<Cfg>(p, s, scratch, scratch_floats, 0, pc);
}

template <class C>
static int64_t num_workgroups(const ConvKParams& d) {
    return ceil_div(d.W, C::TW) * ceil_div(d.H, C::ROWS) * d.T * ceil_div(d.Cout, C::MT);
}

// split-staged precisions: the weights were packed with stemseg_hip_pack_conv_weight_prec(..., precision).  Tile = the largest whose
// PLANNED launch (with split-K where scratch is given) still covers the chip; tile_cfg 1 / 2 / 3 force big / medium / small, 5 the
// flat form of the big 2-D tile (f16x3; tests and sweeps).  p: the launch, d: its planning shape.
template <int BFV>
static int launch_split_family(ConvKParams& p, const ConvKParams& d, const PlanCtx* pc, hipStream_t s, float* scratch, int64_t scratch_floats, int tile_cfg,
                               bool k3, bool k2) {
    typedef SplitTiles<BFV> F;
    using Y3Big = typename F::Y3Big; using Y3Med = typename F::Y3Med; using Y3Small = typename F::Y3Small;
    using Y2Big = typename F::Y2Big; using Y2Med = typename F::Y2Med; using Y2Small = typename F::Y2Small; using Y2M64 = typename F::Y2M64;
    using Y1Small = typename F::Y1Small; using Y1Big = typename F::Y1Big; using Y1M64 = typename F::Y1M64; using Y1Wide = typename F::Y1Wide;
    const bool auto_cfg = tile_cfg <= 0 || tile_cfg > 3;
    int cfg = (auto_cfg || tile_cfg == 6) ? 0 : tile_cfg;
    if constexpr (BFV == 3) {
        // Flat tiles (f16x3: the default mode): a 2-D tile of 16 rows x 32 columns computes 128 x 224 positions for a 120 x 216 map and
        // 32 x 64 for layer 3's 30 x 54: 13-21 % of the MFMAs feed positions that are never stored; the flat tile computes the halo
        // columns instead (2 of 56).  Measured (tools/conv_sweep.py, T = 32): a flat workgroup is 6 % slower than a 2-D one at pitch
        // 56 -- by-element epilogue -- and 15-25 % slower at pitch 112 / 224, where the staged run is 1.15x / 1.49x the 2-D tile's
        // piece: the 13-21 % fewer workgroups only pay at pitch <= 56 (layer-3 3x3: 256 -> 224 workgroups, 207 -> 191 us).
        const bool flat_ok = p.vec4 && p.dec_W == 0 && p.in_ys % 4 == 0 && p.W + 2 <= p.in_ys && p.in_ys <= 224 && p.Cout % 128 == 0;
        if (flat_ok && k2 && !p.gn_part && p.in_ts == (int64_t)p.in_H * p.in_ys && p.in_H == p.H + 2 &&
            (tile_cfg == 5 || (auto_cfg && num_workgroups<Y2Big>(d) >= (scratch ? 96 : 384)))) {
            const double e2d = (double)p.H * p.W / ((double)Y2Big::ROWS * ceil_div(p.H, Y2Big::ROWS) * 32.0 * ceil_div(p.W, 32));
            const double efl = (double)d.T * p.H * p.W / (512.0 * ceil_div((int64_t)d.T * p.in_ts, 512));
            if (tile_cfg == 5 || (efl > 1.04 * e2d && p.in_ys <= 56)) {
                p.flat_t = 1;
                if (p.in_ys <= 56) return launch_cfg<typename F::template Y2Flat<56>>(p, s, scratch, scratch_floats, 0, pc);
                if (p.in_ys <= 112) return launch_cfg<typename F::template Y2Flat<112>>(p, s, scratch, scratch_floats, 0, pc);
                return launch_cfg<typename F::template Y2Flat<224>>(p, s, scratch, scratch_floats, 0, pc);
            }
        }
    }
    // Block tiles (f16x3): where the big 16 x 32 tile would be chosen and the map wastes > 6 % more of it than of 20 x 24 tiles of 4 x 8 blocks
    // (120 x 216: 10.6 % vs 0), take those; tile_cfg 6 forces them.  Same k order per output: bit-identical to the 16 x 32 tiles.
    static const bool blk_on = [] { const char* e = getenv("STEMSEG_BLK_TILES"); return !(e && e[0] == '0'); }();      // (A/B switch; default on)
    auto blk_gain = [&]() {
        if (!blk_on) return false;
        const double e_blk = (double)p.H * p.W / ((double)ceil_div(p.H, 20) * 20 * ceil_div(p.W, 24) * 24);
        const double e_big = (double)p.H * p.W / ((double)ceil_div(p.H, 16) * 16 * ceil_div(p.W, 32) * 32);
        return e_blk > 1.06 * e_big;
    };
    if constexpr (BFV == 3) {
        if (k3 && p.vec4 && (tile_cfg == 6 || (auto_cfg && num_workgroups<Y3Big>(d) >= 384 && num_workgroups<typename F::Y3Blk>(d) >= 384 && blk_gain())))
            return launch_cfg<typename F::Y3Blk>(p, s, scratch, scratch_floats, 0, pc);
        if (k2 && p.vec4 && p.Cout > 64 && (tile_cfg == 6 || (auto_cfg && num_workgroups<Y2Big>(d) >= (scratch ? 96 : 384) && num_workgroups<typename F::Y2Blk>(d) >= 384 && blk_gain())))
            return launch_cfg<typename F::Y2Blk>(p, s, scratch, scratch_floats, 0, pc);
    }
    if (k3) {
        if (cfg == 0) cfg = num_workgroups<Y3Big>(d) >= 384 ? 1 : (num_workgroups<Y3Med>(d) >= (scratch ? 32 : 256) ? 2 : 3);
        if constexpr (BFV == 3) {
            // where the medium tile is the choice and the map wastes > 6 % more of it than of 4-row x 56-column block tiles, take those (tile_cfg 7
            // forces them; STEMSEG_BLK_TILES=0 switches all block tiles off).  Same k order per output: bit-identical to the 8 x 32 tile.
            const double e_b7 = (double)p.H * p.W / ((double)ceil_div(p.H, 4) * 4 * ceil_div(p.W, 56) * 56);
            const double e_med = (double)p.H * p.W / ((double)ceil_div(p.H, 8) * 8 * ceil_div(p.W, 32) * 32);
            static const bool blk7_on = [] { const char* e = getenv("STEMSEG_BLK7"); return !(e && e[0] == '0'); }();      // (A/B switch; default on)
            if (p.vec4 && (tile_cfg == 7 || (auto_cfg && cfg == 2 && blk_on && blk7_on && e_b7 > 1.06 * e_med)))
                return launch_cfg<typename F::Y3Blk7>(p, s, scratch, scratch_floats, 0, pc);
        }
        if (cfg == 1) return launch_cfg<Y3Big>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg == 2) return launch_cfg<Y3Med>(p, s, scratch, scratch_floats, 0, pc);
        return launch_cfg<Y3Small>(p, s, scratch, scratch_floats, 0, pc);
    }
    if (k2) {
        if (p.Cout <= 64) return launch_cfg<Y2M64>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg == 0) {
            const int64_t need = scratch ? 96 : 384;
            cfg = num_workgroups<Y2Big>(d) >= need ? 1 : (num_workgroups<Y2Med>(d) >= need ? 2 : 3);
        }
        if (cfg == 1) return launch_cfg<Y2Big>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg == 2) return launch_cfg<Y2Med>(p, s, scratch, scratch_floats, 0, pc);
        return launch_cfg<Y2Small>(p, s, scratch, scratch_floats, 0, pc);
    }
    if (p.Cout <= 64) return launch_cfg<Y1M64>(p, s, scratch, scratch_floats, 0, pc);
    if (tile_cfg == 3 && p.Cout % 256 == 0) return launch_cfg<Y1Wide>(p, s, scratch, scratch_floats, 0, pc);
    // reductions / square 1x1 convs onto >= 256 channels: the 256-channel tile splits every input element once per 256 outputs
    // (measured, tools/conv_sweep.py: 1024 -> 256 173 -> 155 us, 256 -> 256 at 4x 806 -> 728 us; short-K expansions lose with it)
    if (auto_cfg && p.Cout % 256 == 0 && p.Cin >= p.Cout && !p.res && num_workgroups<Y1Wide>(d) >= 128)
        return launch_cfg<Y1Wide>(p, s, nullptr, 0, 0, pc);
    if (cfg == 0 || cfg > 2) {
        if (BFV == 3) {
            // f16x3 with the one-phase weight schedule (WMODE 1): the 128-voxel tile wins on every 1x1 shape of the step (tools/conv_sweep.py,
            // T = 32: 64 -> 256 444 -> 403 us, 128 -> 512 263 -> 236, 512 -> 2048 128 -> 115, 512 -> 128 146 -> 133, 2048 -> 512 123 -> 115)
            cfg = 2;
        } else {
            cfg = (num_workgroups<Y1Big>(d) >= (scratch ? 96 : 512)) ? 1 : 2;
            if (p.Cin <= 256 && p.Cout >= 4 * p.Cin && num_workgroups<Y1Big>(d) < 2048) cfg = 2;
        }
    }
    if (cfg == 1) return launch_cfg<Y1Big>(p, s, scratch, scratch_floats, 0, pc);
    return launch_cfg<Y1Small>(p, s, scratch, scratch_floats, 0, pc);
}

int launch_conv3d(const StemsegVolume& in, const float* packed_w, const float* bias, const StemsegVolume& out,
                  int kt, int kh, int kw, int tile_cfg, hipStream_t s, float* scratch, int64_t scratch_floats, const ConvEpilogue* epi) {
    SS_CHECK_ARG(in.ptr && out.ptr && packed_w, "conv3d: null pointer");
    const int prec = epi ? epi->precision : STEMSEG_PRECISION_F32;
    SS_CHECK_ARG(prec == STEMSEG_PRECISION_F32 || prec == STEMSEG_PRECISION_BF16X6 || prec == STEMSEG_PRECISION_F16X3,
                 "conv3d: precision %d (0 f32, 2 bf16x6, 3 f16x3)", prec);
    const bool k3 = (kt == 3 && kh == 3 && kw == 3), k1 = (kt == 1 && kh == 1 && kw == 1), k2 = (kt == 1 && kh == 3 && kw == 3);
    const bool k4 = (kt == 1 && kh == 4 && kw == 4);        // (f16x3 only: the space-to-depth form of the 7x7 stride-2 stem)
    SS_CHECK_ARG(k3 || k1 || k2 || (k4 && prec == STEMSEG_PRECISION_F16X3), "conv3d: kernel %dx%dx%d unsupported (3x3x3, 1x3x3, 1x1x1; 1x4x4 in f16x3 mode)", kt, kh, kw);
    const bool flat = epi && epi->dec_W > 0;
    if (flat) {
        SS_CHECK_ARG(k1 && in.T == 1 && in.H == 1 && epi->dec_H > 0 && (int64_t)out.T * out.H * out.W == in.W &&
                     out.H == epi->dec_H && out.W == epi->dec_W, "conv3d: flat-decode epilogue needs a 1x1x1 conv on a [C][V] input with V == T*H*W of `out`");
    } else {
        SS_CHECK_ARG(in.T == out.T + kt - 1 && in.H == out.H + kh - 1 && in.W == out.W + kw - 1,
                     "conv3d: input extents (%d,%d,%d) must be output (%d,%d,%d) + kernel - 1", in.T, in.H, in.W, out.T, out.H, out.W);
    }
    SS_CHECK_ARG(in.C % 4 == 0 && out.C % 32 == 0, "conv3d: Cin %% 4 == 0 and Cout %% 32 == 0 required (got %d, %d)", in.C, out.C);
    ConvKParams p;
    p.in = in.ptr; p.in_cs = in.c_stride; p.in_ts = in.t_stride; p.in_ys = in.y_stride; p.in_limit = in.limit; p.in_H = in.H;
    p.wpk = packed_w; p.bias = bias;
    p.out = out.ptr; p.out_cs = out.c_stride; p.out_ts = out.t_stride; p.out_ys = out.y_stride;
    p.Cin = in.C; p.Cout = out.C; p.T = out.T; p.H = out.H; p.W = out.W;
    if (flat) { p.T = 1; p.H = 1; p.W = in.W; }
    p.relu = epi ? epi->relu : 0;
    p.res = epi ? epi->res : nullptr;
    p.res_cs = epi ? epi->res_cs : 0; p.res_ts = epi ? epi->res_ts : 0; p.res_ys = epi ? epi->res_ys : 0;
    p.dec_H = flat ? epi->dec_H : 0; p.dec_W = flat ? epi->dec_W : 0;
    p.gn_part = epi ? epi->gn_part : nullptr;
    p.gn_cpg = epi ? epi->gn_cpg : 0; p.gn_cap = epi ? epi->gn_cap : 0; p.gn_slot0 = 0;
    p.gn_used_host = epi ? epi->gn_used : nullptr;
    p.out_p16 = nullptr;
    if (epi && epi->p16_out) {
        const bool dense = out.t_stride == (int64_t)out.H * out.W && out.y_stride == out.W && out.c_stride == (int64_t)out.T * out.H * out.W;
        SS_CHECK_ARG(epi->p16_done && dense && !flat && !epi->res && !epi->gn_part && out.C % 8 == 0 && prec == STEMSEG_PRECISION_F16X3 && (!epi || epi->nb <= 1),
                     "conv3d: the pair-plane output needs f16x3, a dense output volume, no residual / statistics / clip batch");
        *epi->p16_done = 0;
        p.out_p16 = epi->p16_out;
        p.gn_used_host = epi->p16_done;                    // (see ConvKParams::out_p16)
    }
    p.nb = (epi && epi->nb > 1) ? epi->nb : 1;
    p.in_bs = p.nb > 1 ? epi->in_bs : 0; p.out_bs = p.nb > 1 ? epi->out_bs : 0; p.gn_bs = p.nb > 1 ? epi->gn_bs : 0;
    SS_CHECK_ARG(p.nb == 1 || (!p.res && p.nb <= 65535 && p.in_bs % 4 == 0 && p.out_bs % 4 == 0), "conv3d: a clip batch takes no residual and 16-byte aligned clip strides");
    SS_CHECK_ARG(!p.gn_part || ((p.gn_cpg == 4 || p.gn_cpg == 8) && p.gn_used_host && p.Cout % p.gn_cpg == 0 && !(epi->relu || epi->res)),
                 "conv3d: fused GroupNorm statistics need groups of 4 or 8 channels and a plain (bias-only) epilogue");
    auto al16 = [](const void* ptr, int64_t cs, int64_t ts, int64_t ys, int T, int H) {
        return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && (cs % 4 == 0) && (T == 1 || ts % 4 == 0) && (H == 1 || ys % 4 == 0);
    };
    p.vec_epi = (!flat && p.W % 4 == 0 && al16(p.out, p.out_cs, p.out_ts, p.out_ys, p.T, p.H) &&
                 (!p.res || al16(p.res, p.res_cs, p.res_ts, p.res_ys, p.T, p.H))) ? 1 : 0;
    p.tiles_x = p.tiles_y = 0;
    p.t_fastest = 1;
    p.flat_t = 0; p.T_all = p.T;
    const bool aligned = (reinterpret_cast<uintptr_t>(in.ptr) % 16 == 0) && (in.c_stride % 4 == 0) &&
                         (in.T == 1 || in.t_stride % 4 == 0) && (in.H == 1 || in.y_stride % 4 == 0);
    p.vec4 = aligned ? 1 : 0;
    SS_CHECK_ARG(reinterpret_cast<uintptr_t>(packed_w) % 16 == 0, "conv3d: packed weights must be 16-byte aligned");
    // planning shape (see PlanCtx): `frames` frames in this launch, planned as `plan_frames`
    PlanCtx pc_store;
    const PlanCtx* pc = nullptr;
    if (epi && epi->plan_frames > 0) {
        SS_CHECK_ARG(epi->frames > 0 && epi->plan_scratch_floats >= 0, "conv3d: plan_frames needs the launch's own frame count");
        pc_store.shape = p;
        pc_store.scratch_floats = epi->plan_scratch_floats;
        if (p.T == 1 && p.H == 1) {                        // a flat [C][V] launch: V = frames x (voxels per frame)
            SS_CHECK_ARG(p.W % epi->frames == 0, "conv3d: flat volume of %d voxels is not %d whole frames", p.W, epi->frames);
            const int64_t Wp = (int64_t)(p.W / epi->frames) * epi->plan_frames;
            SS_CHECK_ARG(Wp < (1ll << 31), "conv3d: planning volume too large");
            pc_store.shape.W = (int)Wp;
        } else {
            SS_CHECK_ARG(p.T == epi->frames, "conv3d: launch of %d t-planes announced as %d frames", p.T, epi->frames);
            pc_store.shape.T = epi->plan_frames;
        }
        pc_store.shape.T_all = pc_store.shape.T;
        pc = &pc_store;
    }
    const ConvKParams& d = pc ? pc->shape : p;
    if (k4) return launch_cfg<SplitTiles<3>::Y4Stem>(p, s, nullptr, 0, 0, pc);
    if (prec == STEMSEG_PRECISION_BF16X6) return launch_split_family<2>(p, d, pc, s, scratch, scratch_floats, tile_cfg, k3, k2);
    if (prec == STEMSEG_PRECISION_F16X3) return launch_split_family<3>(p, d, pc, s, scratch, scratch_floats, tile_cfg, k3, k2);
    const bool auto_cfg = tile_cfg <= 0 || tile_cfg > 3;
    if (k3) {
        int cfg = tile_cfg;
        if (auto_cfg) {
            // largest tile that still gives every CU two workgroups
            if (num_workgroups<K3Big>(d) >= 512) cfg = 1;
            else if (num_workgroups<K3Med>(d) >= 384) cfg = 2;
            else cfg = 3;
        }
        if (cfg == 2 && prefer_flat<K3FlatMed56, K3Med>(p, tile_cfg)) return launch_cfg<K3FlatMed56>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg == 3 && prefer_flat<K3FlatSmall56, K3Small>(p, tile_cfg)) return launch_cfg<K3FlatSmall56>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg == 2 && prefer_flat<K3FlatMed, K3Med>(p, tile_cfg)) return launch_cfg<K3FlatMed>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg == 3 && prefer_flat<K3FlatSmall, K3Small>(p, tile_cfg)) return launch_cfg<K3FlatSmall>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg == 1 && p.vec4 && p.Cout % 128 == 0) {          // (Cin % 2 == 0 holds: Cin % 4 == 0 above)
            if (auto_cfg) return launch_planned<K3BigGL, K3Med>(p, d, pc, s, scratch, scratch_floats, 2, 2);
            return launch_cfg<K3BigGL>(p, s, scratch, scratch_floats, 0, pc);
        }
        if (cfg == 1 && auto_cfg) return launch_planned<K3Big, K3Med>(p, d, pc, s, scratch, scratch_floats, 2, 2);
        if (cfg == 1) return launch_cfg<K3Big>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg == 2) return launch_cfg<K3Med>(p, s, scratch, scratch_floats, 0, pc);
        return launch_cfg<K3Small>(p, s, scratch, scratch_floats, 0, pc);
    }
    if (k2) {
        if (p.Cout <= 64) return launch_cfg<K2M64>(p, s, scratch, scratch_floats, 0, pc);
        int cfg = tile_cfg;
        if (auto_cfg) {   // biggest tile (best weight reuse) that split-K can still spread over the chip
            const int64_t need = scratch ? 96 : 384;
            if (num_workgroups<K2Big>(d) >= need) cfg = 1;
            else if (num_workgroups<K2Med>(d) >= need) cfg = 2;
            else cfg = 3;
        }
        if (cfg == 1 && prefer_flat<K2FlatBig56, K2Big>(p, tile_cfg)) return launch_cfg<K2FlatBig56>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg >= 2 && prefer_flat<K2FlatMed56, K2Med>(p, tile_cfg)) return launch_cfg<K2FlatMed56>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg == 1 && prefer_flat<K2FlatBig, K2Big>(p, tile_cfg)) return launch_cfg<K2FlatBig>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg >= 2 && prefer_flat<K2FlatMed, K2Med>(p, tile_cfg)) return launch_cfg<K2FlatMed>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg == 1 && auto_cfg && num_workgroups<K2Big>(d) >= 512) return launch_planned<K2Big, K2Med>(p, d, pc, s, scratch, scratch_floats, 3, 3);
        if (cfg == 1) return launch_cfg<K2Big>(p, s, scratch, scratch_floats, 0, pc);
        if (cfg == 2) return launch_cfg<K2Med>(p, s, scratch, scratch_floats, 0, pc);
        return launch_cfg<K2Small>(p, s, scratch, scratch_floats, 0, pc);
    }
    if (p.Cout <= 64) return launch_gl<K1M64GL, K1M64>(p, s, scratch, scratch_floats, pc);
    int cfg = tile_cfg;
    if (cfg <= 0 || cfg > 2) {
        cfg = (num_workgroups<K1Big>(d) >= (scratch ? 96 : 512)) ? 1 : 2;
        // expansion convs with a short K (<= 8 channel chunks) are bound by their output / residual traffic: the 128-voxel
        // tile keeps twice as many workgroups in flight (measured, tools/conv_sweep.py: 64->256 +res 233 -> 167 us)
        // -- up to ~2000 workgroups of the big tile; beyond that (several clips per encoder pass) the big tile wins again
        // (T = 32 sweep: 64->256 +res 532 vs 570 us, 128->512 +res 372 vs 438 us)
        if (p.Cin <= 256 && p.Cout >= 4 * p.Cin && num_workgroups<K1Big>(d) < 2048) cfg = 2;
    }
    if (cfg == 1) return launch_gl<K1BigGL, K1Big>(p, s, scratch, scratch_floats, pc);
    return launch_gl<K1SmallGL, K1Small>(p, s, scratch, scratch_floats, pc);
}

int launch_conv3d_gn(const StemsegVolume& in, const float* packed_w, const float* bias, const StemsegVolume& out, int kt, int kh, int kw,
                     int tile_cfg, hipStream_t s, float* splitk_scratch, int64_t splitk_scratch_floats, const ConvEpilogue* epi, int groups,
                     float eps, float* stats, double* gn_scratch, int64_t stats_bs) {
    SS_CHECK_ARG(stats && gn_scratch && groups > 0 && out.C % groups == 0, "conv3d_gn: bad GroupNorm arguments");
    const int cpg = out.C / groups;
    const int64_t S = (int64_t)out.T * out.H * out.W;
    const bool dense = out.t_stride == (int64_t)out.H * out.W && out.y_stride == out.W && out.c_stride == S;
    // upper bound of the slots any tile choice can need: smallest 2-D tile (2 rows x 32 columns) + a fully split-K launch
    const int64_t slot_bound = ceil_div(out.W, 32) * ceil_div(out.H, 2) * out.T + (int64_t)cpg * ceil_div(S, 256);
    if ((cpg != 4 && cpg != 8) || slot_bound > GN_SLOT_CAP || (epi && (epi->relu || epi->res || epi->dec_W > 0))) {
        SS_CHECK_ARG(dense, "conv3d_gn: the separate statistics pass needs a dense output");
        int rc = launch_conv3d(in, packed_w, bias, out, kt, kh, kw, tile_cfg, s, splitk_scratch, splitk_scratch_floats, epi);
        const int nb = (epi && epi->nb > 1) ? epi->nb : 1;
        for (int b = 0; b < nb && !rc; ++b)
            rc = launch_gn_stats(out.ptr + (nb > 1 ? b * epi->out_bs : 0), out.C, S, groups, eps, stats + (int64_t)b * stats_bs, gn_scratch + (nb > 1 ? b * epi->gn_bs : 0), s);
        return rc;
    }
    ConvEpilogue e = epi ? *epi : ConvEpilogue();
    int used = 0;
    e.gn_part = gn_scratch; e.gn_cpg = cpg; e.gn_cap = GN_SLOT_CAP; e.gn_used = &used;
    const int rc = launch_conv3d(in, packed_w, bias, out, kt, kh, kw, tile_cfg, s, splitk_scratch, splitk_scratch_floats, &e);
    if (rc) return rc;
    return launch_gn_finalize_slots(gn_scratch, groups, GN_SLOT_CAP, used, (double)cpg * (double)S, eps, stats, s, e.nb, e.gn_bs, stats_bs);
}

}  // namespace stemseg

extern "C" int64_t stemseg_hip_conv3d_gn_scratch_doubles(int32_t Cout, int32_t groups) {
    return (Cout > 0 && groups > 0) ? stemseg::gn_scratch_doubles(Cout, groups) : 0;
}

extern "C" int stemseg_hip_conv3d_gn(const StemsegVolume* in, const float* packed_w, const float* bias, const StemsegVolume* out,
                                     int32_t kt, int32_t kh, int32_t kw, int32_t tile_cfg, float* splitk_scratch, int64_t splitk_scratch_floats,
                                     int32_t precision, int32_t groups, float eps, float* stats, double* gn_scratch, void* stream) {
    using namespace stemseg;
    SS_CHECK_ARG(in && out, "conv3d_gn: null volume");
    ConvEpilogue e;
    e.precision = precision;
    return launch_conv3d_gn(*in, packed_w, bias, *out, kt, kh, kw, tile_cfg, as_stream(stream), splitk_scratch, splitk_scratch_floats, &e, groups,
                            eps, stats, gn_scratch);
}

extern "C" int stemseg_hip_pack_conv_weight(const float* w, float* packed, int32_t Cout, int32_t Cin, int32_t taps, void* stream) {
    using namespace stemseg;
    SS_CHECK_ARG(w && packed, "pack_conv_weight: null pointer");
    SS_CHECK_ARG(Cin % 4 == 0 && Cout % 32 == 0 && taps > 0, "pack_conv_weight: Cin %% 4, Cout %% 32 (got %d, %d)", Cin, Cout);
    const int64_t n = (int64_t)Cout * Cin * taps;
    const int blocks = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, packed, Cout, Cin, taps);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

// bytes of the split-staged packing with `planes` 16-bit planes: per channel chunk [k-group][plane][lane half][Cout] x 16 B
// (channel chunk, taps per k-group) of the split-staged packing: must match SplitTiles<precision>
static void split_chunking(int taps, int precision, int& CK, int& TPG) {
    if (taps == 27) { CK = 4; TPG = 4; }
    else if (taps == 9) { CK = precision == STEMSEG_PRECISION_F16X3 ? 16 : 8; TPG = precision == STEMSEG_PRECISION_F16X3 ? 1 : 2; }
    else if (taps == 16) { CK = 16; TPG = 1; }
    else { CK = 32; TPG = 1; }
}
static int64_t split_packed_bytes(int32_t Cout, int32_t Cin, int32_t taps, int planes, int precision) {
    int CK, TPG;
    split_chunking(taps, precision, CK, TPG);
    const int CPH = 8 / TPG, NTG = (taps + TPG - 1) / TPG, NCG = CK / (2 * CPH);
    return (int64_t)((Cin + CK - 1) / CK) * planes * (NTG * NCG) * 2 * Cout * 16;
}

extern "C" int64_t stemseg_hip_packed_weight_bytes_prec(int32_t Cout, int32_t Cin, int32_t taps, int32_t precision) {
    if (Cout <= 0 || Cin <= 0 || !(taps == 27 || taps == 9 || taps == 1 || (taps == 16 && precision == STEMSEG_PRECISION_F16X3))) return 0;
    if (precision == STEMSEG_PRECISION_BF16X6) return split_packed_bytes(Cout, Cin, taps, 3, precision);
    if (precision == STEMSEG_PRECISION_F16X3) return split_packed_bytes(Cout, Cin, taps, SS_F16_WPLANES, precision) + 8 * (int64_t)Cout;      // planes + per output channel: float 1 / scale, uint32 bits of max|w|
    return 0;
}

extern "C" int stemseg_hip_pack_conv_weight_prec(const float* w, void* packed, int32_t Cout, int32_t Cin, int32_t taps, int32_t precision, void* stream) {
    using namespace stemseg;
    SS_CHECK_ARG(precision == STEMSEG_PRECISION_BF16X6 || precision == STEMSEG_PRECISION_F16X3, "pack_conv_weight_prec: precision must be 2 (bf16x6) or 3 (f16x3)");
    SS_CHECK_ARG(w && packed, "pack_conv_weight_prec: null pointer");
    SS_CHECK_ARG(taps == 27 || taps == 9 || taps == 1 || (taps == 16 && precision == STEMSEG_PRECISION_F16X3), "pack_conv_weight_prec: taps must be 27, 9 or 1 (16: f16x3, the stem)");
    int CK, TPG;
    split_chunking(taps, precision, CK, TPG);
    SS_CHECK_ARG(Cin % 4 == 0 && Cout % 32 == 0, "pack_conv_weight_prec: Cin %% 4, Cout %% 32 (got %d, %d)", Cin, Cout);
    if (precision == STEMSEG_PRECISION_BF16X6) {
        const int64_t n = split_packed_bytes(Cout, Cin, taps, 3, precision) / 16;
        const int blocks = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
        hipLaunchKernelGGL(pack_conv_weight_bf16x6_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, reinterpret_cast<uint4*>(packed), Cout, Cin,
                           taps, CK, TPG);
        SS_LAUNCH_CHECK();
        return STEMSEG_OK;
    }
    const int64_t n = split_packed_bytes(Cout, Cin, taps, SS_F16_WPLANES, precision) / 16;
    unsigned int* max_bits = reinterpret_cast<unsigned int*>(reinterpret_cast<float*>(reinterpret_cast<uint4*>(packed) + n) + Cout);
    hipLaunchKernelGGL(absmax_rows_kernel, dim3((unsigned)Cout), dim3(256), 0, as_stream(stream), w, (int64_t)Cin * taps, max_bits);
    SS_LAUNCH_CHECK();
    const int blocks = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
    hipLaunchKernelGGL(pack_conv_weight_f16x3_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, reinterpret_cast<uint4*>(packed), Cout, Cin,
                       taps, CK, TPG);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_conv3d(const StemsegVolume* in, const float* packed_w, const float* bias, const StemsegVolume* out,
                                  int32_t kt, int32_t kh, int32_t kw, int32_t tile_cfg, float* splitk_scratch,
                                  int64_t splitk_scratch_floats, const StemsegConvEpilogue* epilogue, void* stream) {
    using namespace stemseg;
    SS_CHECK_ARG(in && out, "conv3d: null volume");
    ConvEpilogue e;
    if (epilogue) {
        e.relu = epilogue->relu; e.res = epilogue->residual; e.res_cs = epilogue->res_c_stride; e.res_ts = epilogue->res_t_stride;
        e.res_ys = epilogue->res_y_stride; e.dec_H = epilogue->decode_H; e.dec_W = epilogue->decode_W;
        e.precision = epilogue->precision;
        e.frames = epilogue->frames; e.plan_frames = epilogue->plan_frames; e.plan_scratch_floats = epilogue->plan_scratch_floats;
    }
    return launch_conv3d(*in, packed_w, bias, *out, kt, kh, kw, tile_cfg, as_stream(stream), splitk_scratch, splitk_scratch_floats,
                         epilogue ? &e : nullptr);
}

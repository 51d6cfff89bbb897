/*
 * stemseg_hip.h -- C-ABI of libstemseg_hip.so: the MI355X (gfx950) implementation of STEm-Seg's
 * embed+cluster hot path.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * The reference (sabarim/STEm-Seg) has no FFI of its own: its extension surface is Python
 * (registries + nn.Module / callable contracts).  Each entry point below
 * names the reference interface it stands in for (file:line relative to the reference tree); the
 * Python host-side mirror in stem-seg_amd/stemseg_amd/ binds these with ctypes and re-exposes the
 * reference's class / function names.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success or a negative STEMSEG_E_* code; nothing throws;
 *     stemseg_hip_last_error() gives a thread-local message valid until the next call on the thread.
 *   - no allocation and no ownership transfer: all device memory (inputs, outputs, workspaces) is
 *     provided by the caller; pointers are device pointers unless the name ends in _host.
 *   - work is enqueued on the hipStream_t passed as `void* stream` (NULL = default stream); the
 *     only functions that synchronise with the host are the *_read_* ones, and they say so.
 *   - all floating point is IEEE fp32 (the reference asserts fp32, inference/clusterers.py:12),
 *     labels are int64, masks are uint8.
 */
#ifndef STEMSEG_HIP_H
#define STEMSEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STEMSEG_HIP_ABI_VERSION 10

#define STEMSEG_OK              0
#define STEMSEG_E_INVALID      -1   /* bad argument / unsupported shape               */
#define STEMSEG_E_HIP          -2   /* a HIP runtime call failed (see last_error)     */
#define STEMSEG_E_WORKSPACE    -3   /* workspace too small                            */
#define STEMSEG_E_UNSUPPORTED  -4

#define STEMSEG_MAX_HEAD_OUT   10   /* fused heads kernel: embedding + variance + seediness channels (xytff + seediness = 9) */
#define STEMSEG_MAX_INSTANCES  64   /* upper bound for ClusterParams.max_instances    */
#define STEMSEG_MAX_EMB_DIMS    8

int         stemseg_hip_version(void);
const char* stemseg_hip_last_error(void);
/* number of HIP devices visible, or a negative error (used to fail loudly when there is no GPU) */
int         stemseg_hip_device_count(void);

/* Optional in-library profiler (measurement only): when enabled, every convolution launch is bracketed by a
 * hipEvent pair on its own stream.  profile_read SYNCHRONISES the device, then writes per tag t
 * out_host[3t] = summed kernel ms, out_host[3t+1] = summed algorithmic FLOPs, out_host[3t+2] = launches, and clears the
 * log.  Tags: 8 / 4 / 2 = 3x3x3 conv with an 8 / 4 / 2-row tile, 18 / 14 = 1x1x1 conv (256 / 128-voxel tile). */
int stemseg_hip_profile_enable(int32_t on);
int stemseg_hip_profile_read(double* out_host, int32_t n_tags);

/* ------------------------------------------------------------------------------------------------
 * Volumes.  A volume is a [C][T][H][W] fp32 tensor addressed as
 *     ptr + c*c_stride + t*t_stride + y*y_stride + x            (strides in floats, x contiguous)
 * which covers the reference's dense NCDHW tensors (N = 1), channel slices of concat buffers and the
 * zero-haloed ("padded") layout the 3x3x3 convolution consumes.  limit = number of floats that may
 * be read starting at ptr (tile over-reads beyond a row are clamped against it).
 * ---------------------------------------------------------------------------------------------- */
typedef struct StemsegVolume {
    float*  ptr;
    int64_t c_stride, t_stride, y_stride;
    int32_t C, T, H, W;
    int64_t limit;
} StemsegVolume;

/* Geometry of the zero-haloed layout for a logical [C][T][H][W] volume: one zero voxel on every side
 * of T, H, W, row pitch rounded up to 4 floats.  out[0..4] = {row_pitch, t_stride, c_stride,
 * total_floats (incl. tail slack), interior_offset}: element (c,t,y,x) lives at
 * base + interior_offset + c*c_stride + t*t_stride + y*row_pitch + x. */
int stemseg_hip_padded_geometry(int32_t C, int32_t T, int32_t H, int32_t W, int64_t out[5]);

/* ------------------------------------------------------------------------------------------------
 * Decoder building blocks (each is also used on its own by the parity tests).
 * Reference: nn.Conv3d / nn.GroupNorm / nn.ReLU / nn.AvgPool3d / F.interpolate as instantiated in
 * stemseg/modeling/embedding_decoder.py:20-96 and common.py:69-78.
 * ---------------------------------------------------------------------------------------------- */

/* Repack a reference-layout conv weight [Cout][Cin][taps] (taps = kt*kh*kw, row-major over kt,kh,kw)
 * into the MFMA implicit-GEMM layout [Cin/4][taps][4][Cout].  Cin % 4 == 0, Cout % 32 == 0. */
int stemseg_hip_pack_conv_weight(const float* w, float* packed, int32_t Cout, int32_t Cin, int32_t taps, void* stream);

/* out[co,t,y,x] = bias[co] + sum_{ci,dt,dy,dx} W[co,ci,dt,dy,dx] * in[ci, t+dt, y+dy, x+dx]
 * "valid" cross-correlation: `in` is the haloed volume (in->T = out->T + kt-1, likewise H, W), so
 * Conv3d(k=3, padding=1) of embedding_decoder.py:21 is `in` = zero-haloed layout.  (kt,kh,kw) is
 * (3,3,3) or (1,1,1).  bias may be NULL.  tile_cfg: 0 = auto, 1..3 = force a tile shape (tuning).
 * splitk_scratch (may be NULL): device scratch of splitk_scratch_floats floats; when a layer yields too few
 * workgroups to fill the chip the input channels are split over up to 16 workgroups per tile, partial sums go to the
 * scratch (k * Cout*T*H*W floats) and a second kernel reduces them in fixed order (+ bias) into `out`. */
typedef struct StemsegConvEpilogue {
    int32_t      relu;             /* max(v, 0) last                                                                     */
    const float* residual;         /* NULL or a tensor added before the ReLU, addressed at the OUTPUT's (c,t,y,x) with: */
    int64_t      res_c_stride, res_t_stride, res_y_stride;
    int32_t      decode_H, decode_W; /* > 0: the 1x1x1 conv runs on a flat [C][V] input (in->T = in->H = 1, in->W = V) and
                                        voxel v is stored at (t,y,x) = (v/(H*W), (v/W)%H, v%W) of `out` (e.g. dense -> haloed) */
    int32_t      precision;        /* arithmetic of the matrix products; packed_w must be packed for the same mode.
                                      STEMSEG_PRECISION_F32: v_mfma_f32_32x32x2_f32 on the fp32 operands (exact products);
                                      packed_w from stemseg_hip_pack_conv_weight.
                                      STEMSEG_PRECISION_BF16X6: every fp32 operand is split EXACTLY into three bf16 terms
                                      (hi + mid + lo, 24 significand bits) and a*b is the sum of the six products of weight
                                      >= 2^-16 on the bf16 matrix cores, fp32 accumulation: the dropped products are <= 2^-23
                                      |a*b|, below the rounding of the fp32 accumulation itself -- fp32-level results, fp32's
                                      exponent range, 2.7x the fp32-MFMA rate.
                                      STEMSEG_PRECISION_F16X3: both operands are scaled by a power of two (activations by 2^-2,
                                      every OUTPUT CHANNEL's weights so that its largest lands in [2^13, 2^14)) and split into two fp16
                                      terms (hi + lo: 22 significand bits, fp32 has 24); the low activation term is stored as
                                      lo * 2^11 and meets a third weight operand, hi_w * 2^-11, so hi is a normal fp16 number and
                                      the pair keeps 22 bits (or 2^-36 absolute) for 2.5e-4 <= |a| < 2.6e5.  a*b = lo_w*hi_a +
                                      hi_w*lo_a + hi_w*hi_a on the fp16 matrix cores, fp32 accumulation, accumulators scaled back
                                      exactly: the dropped lo*lo product and the split remainder are <= 2^-22 |a*b| -- measured
                                      below the rounding spread of fp32 accumulation orders -- at HALF the matrix work of
                                      bf16x6.  |a| >= 2.6e5 overflows (inf in, non-finite out: see stemseg_hip_nonfinite_flags).
                                      packed_w of both split modes from stemseg_hip_pack_conv_weight_prec. */
    /* Planning shape: the launch holds `frames` frames (its T axis, or -- flat [C][V] input -- V / the per-frame voxel count), and
       the tile shape and split-K factor are decided as if it held `plan_frames`, counting on `plan_scratch_floats` of split-K
       scratch (the real scratch must hold frames / plan_frames times what the plan uses, or the call fails).  The K-partition of a
       split-K launch is a summation order: decided this way, every output bit is a function of the per-frame layer shape, the
       precision and plan_frames -- not of how many frames share the launch.  plan_frames = 0: decide on the real shape. */
    int32_t      frames, plan_frames;
    int64_t      plan_scratch_floats;
} StemsegConvEpilogue;
#define STEMSEG_PRECISION_F32    0
#define STEMSEG_PRECISION_BF16X6 2      /* (1 was the two-term bf16 split of rounds 2-4: ~1e-4 on the maps, retired) */
#define STEMSEG_PRECISION_F16X3  3
/* Split-staged packings, by precision code.  2: per channel chunk [k-group][hi | mid | lo][lane half][Cout] x 8 bf16; 3: the same
 * order with two fp16 planes (hi, lo) of the scaled weights + per output channel a float 1 / (weight scale x activation scale) and
 * the bits of max|w| of the channel (8 * Cout bytes).  bytes = 0 for any other code. */
int64_t stemseg_hip_packed_weight_bytes_prec(int32_t Cout, int32_t Cin, int32_t taps, int32_t precision);
int stemseg_hip_pack_conv_weight_prec(const float* w, void* packed, int32_t Cout, int32_t Cin, int32_t taps, int32_t precision, void* stream);
/* (kt,kh,kw) additionally accepts (1,3,3): a 2-D 3x3 convolution over every t-plane (the encoder's frames).
 * epilogue may be NULL (plain conv + bias). */
int stemseg_hip_conv3d(const StemsegVolume* in, const float* packed_w, const float* bias, const StemsegVolume* out,
                       int32_t kt, int32_t kh, int32_t kw, int32_t tile_cfg, float* splitk_scratch,
                       int64_t splitk_scratch_floats, const StemsegConvEpilogue* epilogue, void* stream);

/* Convolution + the GroupNorm statistics of its output in ONE pass (embedding_decoder.py:21-23: Conv3d followed by
 * GroupNorm): the conv's epilogue -- or, for split-K launches, its reduce kernel -- leaves per-tile fp64 partial sums in
 * gn_scratch (>= stemseg_hip_conv3d_gn_scratch_doubles(Cout, groups) doubles) and one small launch combines them in fixed
 * order (no atomics: bit-identical run to run) into stats[2g] = mean, stats[2g+1] = 1/sqrt(biased_var + eps), so the
 * conv output is not read again for the statistics.  Plain bias epilogue only; groups of 4 or 8 channels take the fused
 * path, any other group size runs stemseg_hip_conv3d + stemseg_hip_groupnorm_stats (dense `out` required then). */
int64_t stemseg_hip_conv3d_gn_scratch_doubles(int32_t Cout, int32_t groups);
int stemseg_hip_conv3d_gn(const StemsegVolume* in, const float* packed_w, const float* bias, const StemsegVolume* out,
                          int32_t kt, int32_t kh, int32_t kw, int32_t tile_cfg, float* splitk_scratch,
                          int64_t splitk_scratch_floats, int32_t precision, int32_t groups, float eps,
                          float* stats, double* gn_scratch, void* stream);

/* GroupNorm statistics over a dense [C][S] tensor (S = T*H*W), `groups` contiguous channel groups:
 * stats[2g] = mean, stats[2g+1] = 1/sqrt(biased_var + eps).  scratch: >= groups*128 doubles. */
int stemseg_hip_groupnorm_stats(const float* x, int32_t C, int64_t S, int32_t groups, float eps,
                                float* stats, double* scratch, void* stream);

/* y = ReLU(GroupNorm(x)) optionally followed by AvgPool3d(3, stride (2,1,1), padding 1,
 * count_include_pad) -- embedding_decoder.py:22-24.  `x` dense [C][T][H][W]; `out` any volume with
 * out->T == (pool ? (T+1)/2 : T). */
int stemseg_hip_gn_relu_pool(const float* x, int32_t C, int32_t T, int32_t H, int32_t W, int32_t groups,
                             const float* stats, const float* gamma, const float* beta, int32_t pool,
                             const StemsegVolume* out, void* stream);

/* F.interpolate(mode='trilinear', align_corners=False) with integer scale (st, sy, sx) -- common.py:77-78 and
 * online_chainer.py:127-140 (sx = sy = 4, st = 1).  `in` dense [C][T][H][W]; out->T/H/W = scaled dims. */
int stemseg_hip_upsample_trilinear(const float* in, int32_t C, int32_t T, int32_t H, int32_t W,
                                   int32_t st, int32_t sy, int32_t sx, const StemsegVolume* out, void* stream);

/* Copy a dense feature stack into a volume (typically the zero-haloed layout).
 * layout 0: in = [C][T][H][W] (the head input of embedding_decoder.py:101-109);
 * layout 1: in = [T][C][H][W] (what the 2-D encoder emits, model_builder.py:154-169). */
int stemseg_hip_copy_to_volume(const float* in, int32_t layout, const StemsegVolume* out, void* stream);

/* Fused 1x1x1 heads (embedding_decoder.py:131-143, seediness_decoder.py:112, inference_model.py:148):
 * out[o, v] = act_o( bias[o] + sum_c w[o][c] * x[c, v] ),  x dense [Cin][V], out dense [n_out][V].
 * act codes: 0 identity, 1 tanh(0.25*z) + grid, 2 sigmoid, 3 exp(z)*10, 4 identity + grid.
 * grid_axis[o]: 0 none, 1 t, 2 y, 3 x -- the coordinate added for act 1/4 (embedding_utils.py:44-120);
 * grid_t/y/x are the linspace vectors of embedding_utils.py:28-41 (length T, H, W). */
int stemseg_hip_heads(const float* x, int32_t Cin, int32_t T, int32_t H, int32_t W,
                      const float* w, const float* bias, int32_t n_out, const int32_t* act_host,
                      const int32_t* grid_axis_host, const float* grid_t, const float* grid_y, const float* grid_x,
                      float* out, void* stream);

/* Overflow guard: flags[b] = 1 when chunk b (of n_flags equal chunks) of x holds an inf / NaN, else 0; every flag is rewritten by
 * every call (no reset needed, graph-replay safe).  The split convolution modes have a finite operand range (f16x3: |activation| <
 * 2.6e5); beyond it their outputs are non-finite BY CONSTRUCTION and the ReLUs / pools keep NaN, so a non-finite head output is the
 * signal to re-run the clip in STEMSEG_PRECISION_BF16X6 (fp32's range) -- never cluster such maps (ClipPipeline.step_checked). */
int stemseg_hip_nonfinite_flags(const float* x, int64_t n, int32_t* flags, int32_t n_flags, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole decoder: SqueezingExpandDecoder.forward (embedding_decoder.py:101-145) and the seediness twin
 * (seediness_decoder.py:92-112), optionally with the bandwidth activation of inference_model.py:148.
 * ---------------------------------------------------------------------------------------------- */
typedef struct StemsegDecoderDesc {
    int32_t struct_bytes;        /* = sizeof(StemsegDecoderDesc), checked                                   */
    int32_t in_channels;         /* FPN channels (256)                                                      */
    int32_t inter[4];            /* INTER_CHANNELS for the 32x,16x,8x,4x branches (256,256,128,128)          */
    int32_t T, H4, W4;           /* clip length and 1/4-resolution output size; H4 % 8 == 0, W4 % 8 == 0    */
    int32_t gn_groups;           /* 32; 0 = NORMALIZATION_LAYER 'none' (nn.Identity, model_builder.py:29-33): gn_w / gn_b must then be ones / zeros */
    float   gn_eps;              /* 1e-5                                                                    */
    int32_t pool[3];             /* common.py:8-24 : temporal pooling layer i: 0 absent, 1 AvgPool3d, 2 MaxPool3d (POOL_TYPE; T=8: 1,1,0) */
    int32_t t_scale[3];          /* common.py:27-35: temporal up-sampling factors (T=8: 1,2,2)              */
    int32_t n_out;               /* head output channels: 1..10 = fused heads kernel (emb + var + seed / seediness);
                                    a multiple of 32 = plain linear head through the 1x1x1 MFMA conv (semseg logits,
                                    semseg_decoder.py:116; weights->head_w is then a packed conv weight, act is ignored) */
    int32_t act[STEMSEG_MAX_EMB_DIMS * 2];        /* per output channel, see stemseg_hip_heads          */
    int32_t grid_axis[STEMSEG_MAX_EMB_DIMS * 2];
    int32_t input_layout;        /* 0: [C][T][h][w] dense, 1: [T][C][h][w] dense, 2: already zero-haloed     */
    int32_t concurrency;         /* 0: every launch on the caller's stream.  k >= 1: the 32x / 16x / 8x branches run
                                    on the library's internal stream set (k-1) % 4 beside the 4x branch (fork / join by
                                    events on the caller's stream; the call is still stream-ordered for the caller).
                                    Decoders that may overlap (embedding + seediness) should use different sets.     */
    int32_t detached;            /* with concurrency >= 1: ALL work (also the 4x branch and the heads) runs on the internal
                                    streams and the call returns without joining; the caller must enqueue
                                    stemseg_hip_decoder_join(concurrency, stream) before it consumes `out` or re-uses the
                                    inputs / workspace.  Lets a twin decoder be enqueued in between (both fill the chip). */
    int32_t precision;           /* STEMSEG_PRECISION_F32 | _BF16X6 | _F16X3 for every convolution of the decoder
                                    (weights in StemsegDecoderWeights must be packed for the same mode)                    */
    int32_t n_clips;             /* clip batch: the decoder runs n_clips clips (0 / 1: one) in ONE launch per stage -- the clip is a grid
                                    dimension of every kernel.  Every launch decision is taken on one clip's shape, so a clip's
                                    output is bit-identical whatever the batch.  The workspace holds n_clips clip plans.          */
    int64_t feat_clip_stride[4]; /* n_clips > 1: floats between consecutive clips' input feature buffers, per level (feats[i] is
                                    clip 0's)                                                                                     */
    int64_t out_clip_stride;     /* n_clips > 1: floats between consecutive clips' outputs (0: n_out * T * H4 * W4, i.e. dense)   */
} StemsegDecoderDesc;

typedef struct StemsegDecoderWeights {
    /* 3x3x3 convs in order block_32x.{0,4,8}, block_16x.{0,4}, block_8x.0, block_4x.0 (packed layout) */
    const float* conv_w[7];
    const float* conv_b[7];
    const float* gn_w[7];
    const float* gn_b[7];
    const float* fuse_w[3];      /* conv_16, conv_8, conv_4 (1x1x1, no bias), packed layout.  Between the last GroupNorm + ReLU of a branch and the
                                    heads' activations the decoder is LINEAR (up-sampling, concatenation, these convs, the 1x1x1 heads), and a
                                    caller may hand over the product matrices instead of the factors:
                                    * fuse_w[2] = NULL: conv_4 folded into the heads -- head_w = W_heads . W_conv4 over the inter[2] + inter[3]
                                      channels of the last concat buffer (dense [n_out][inter[2] + inter[3]], or the packed 1x1x1 weight with that
                                      Cin for a wide head); the inter[3]-channel map is never materialised;
                                    * fuse_w[0..2] = NULL (n_out <= STEMSEG_MAX_HEAD_OUT): the whole tail folded -- 1x1x1 convs commute with
                                      up-sampling, so heads(x) = up(up(up(M32 x32) + M16 y16) + M8 y8) + M4 y4 and head_w = [M32 | M16 | M8 | M4],
                                      M_l dense [n_out][inter[l]] (M4 = Wh W4b, M8 = Wh W4a W8b, M16 = Wh W4a W8a W16b, M32 = Wh W4a W8a W16a for
                                      W = [Wa | Wb] over the (up-sampled, own) halves of a concat input): every level adds its n_out-channel
                                      share at its own resolution, and the heads normalise the 4x branch's raw conv output as they read it.
                                    Either way the same function, within fp32 round-off of the step-by-step form. */
    const float* head_w;         /* [n_out][inter[3]] row-major */
    const float* head_b;         /* [n_out] (zero where the reference conv has no bias) */
    const float* grid_t;         /* [T], [H4], [W4] linspace vectors (may be NULL if no act uses the grid) */
    const float* grid_y;
    const float* grid_x;
} StemsegDecoderWeights;

size_t stemseg_hip_decoder_workspace_bytes(const StemsegDecoderDesc* desc);
/* zero the halos; call once per (workspace, desc) before the first forward */
int stemseg_hip_decoder_init_workspace(const StemsegDecoderDesc* desc, void* workspace, size_t ws_bytes, void* stream);
/* Debug check (SYNCHRONISES the stream): every slice of the workspace is followed by a 64-word guard block that init_workspace filled
 * with a canary; *n_bad_host = guard words that no longer hold it (an out-of-slice write by some kernel), *first_bad_host = float
 * offset of the first such word in the workspace (-1: none).  Costs one tiny launch + one 16-byte read-back. */
int stemseg_hip_decoder_check_workspace(const StemsegDecoderDesc* desc, const void* workspace, size_t ws_bytes, int32_t* n_bad_host,
                                        int64_t* first_bad_host, void* stream);
/* feats[0..3] = 32x,16x,8x,4x feature stacks in desc->input_layout; out = [n_out][T][H4][W4] dense */
int stemseg_hip_decoder_forward(const StemsegDecoderDesc* desc, const StemsegDecoderWeights* weights,
                                const float* const feats[4], float* out,
                                void* workspace, size_t ws_bytes, void* stream);

/* make `stream` wait for a detached decoder_forward issued with the same concurrency set */
int stemseg_hip_decoder_join(int32_t concurrency, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 2-D encoder: ResNet-50/101 + FPN over the T frames of a clip (backbone/resnet.py:105-113, fpn.py:47-69,
 * model_builder.py:154-169).  FrozenBatchNorm (make_layers.py:51-63) is folded into conv weight / bias by the caller.
 * ---------------------------------------------------------------------------------------------- */
#define STEMSEG_MAX_ENCODER_BLOCKS 40

typedef struct StemsegEncoderDesc {
    int32_t struct_bytes;        /* = sizeof(StemsegEncoderDesc)                                        */
    int32_t blocks[4];           /* bottleneck blocks per stage: R-50 {3,4,6,3}, R-101 {3,4,23,3}       */
    int32_t T, H, W;             /* frames per call and padded frame size (multiples of 32)              */
    int32_t out_channels;        /* 256                                                                  */
    int32_t precision;           /* STEMSEG_PRECISION_F32 | _BF16X6 | _F16X3 (every 3x3 / 1x1 conv; the 7x7 stem is always fp32-input MFMA) */
    int32_t n_clips;             /* >= 1: the T frames are n_clips consecutive clips of T / n_clips frames; each clip's four maps
                                    go to their own output volumes (frames are independent in the encoder, so several clips
                                    share one pass: layer3 / layer4 launches grow from 0.4 to n_clips x 0.4 waves of the chip) */
    int32_t clip_frames;         /* 0: T / n_clips.  > 0 with clip_stride > 0: the clips are OVERLAPPING windows of clip_frames
                                    frames every clip_stride frames of the pass ((n_clips - 1) * clip_stride + clip_frames == T),
                                    the way inference/main.py:23-49 cuts a sequence: a frame shared by two clips goes through
                                    the encoder once (the reference's cross-clip feature cache, inference_model.py:83-108) and
                                    each clip's window of the FPN maps is copied to its output volumes */
    int32_t clip_stride;
    int32_t plan_frames;         /* > 0: every convolution of the pass decides its tile shape and split-K factor as if the pass held
                                    this many frames (StemsegConvEpilogue.plan_frames): a frame's four maps are then bit-identical
                                    whatever T, n_clips and the windowing of the pass are -- one clip, a batch of clips, or the
                                    union of overlapping windows -- which is what makes an N-rank sequence job reproduce the
                                    one-rank labels.  0: decide on the real T (fastest for that T; results depend on it). */
    int32_t fuse_tail;           /* bit (stage - 1) set (1 | 2 | 4 = stages 1-3), f16x3 mode: conv3 (+ bias + identity + ReLU) of a bottleneck block and conv1 (+ bias + ReLU) of
                                    the next one run as ONE back-to-back kernel (resnet.py:262-282 of two consecutive blocks): the 4x-wide block
                                    output is written once and never read back by conv1, and conv2 hands its output over as fp16 operand
                                    planes.  Same operands and k order: bit-identical to the separate launches wherever those run without
                                    split-K.  0: three launches per block everywhere (rounds 1-5).  Bits 3-4 pick stage 3's kernel (A/B):
                                    0 = the library's choice (one wave per SIMD, bit-identical like stages 1-2), 1 (value 8) = the
                                    16-column form (fp32 round-off apart from the separate launches), 2 (value 16) = one wave per SIMD. */
} StemsegEncoderDesc;

typedef struct StemsegEncoderWeights {
    const float* stem_w;         /* [147][64] tap-major ((c*7+dy)*7+dx), BN folded */
    const float* stem_b;         /* [64] */
    const float* stem_w_s2d;     /* optional, f16x3 mode: the stem as a stride-1 4x4 convolution over the space-to-depth image -- weights
                                    [64][12][1][4][4] with W2[co][(p*2+q)*3+c][a][b] = w[co][c][2a+p-1][2b+q-1] (0 where an index is -1),
                                    packed by stemseg_hip_pack_conv_weight_prec(taps = 16, STEMSEG_PRECISION_F16X3).  NULL: the exact
                                    fp32-MFMA stem from stem_w in every mode */
    /* per bottleneck block, in network order; conv weights in the packed layout of stemseg_hip_pack_conv_weight */
    const float* conv1_w[STEMSEG_MAX_ENCODER_BLOCKS];   const float* conv1_b[STEMSEG_MAX_ENCODER_BLOCKS];
    const float* conv2_w[STEMSEG_MAX_ENCODER_BLOCKS];   const float* conv2_b[STEMSEG_MAX_ENCODER_BLOCKS];
    const float* conv3_w[STEMSEG_MAX_ENCODER_BLOCKS];   const float* conv3_b[STEMSEG_MAX_ENCODER_BLOCKS];
    const float* down_w[STEMSEG_MAX_ENCODER_BLOCKS];    const float* down_b[STEMSEG_MAX_ENCODER_BLOCKS];   /* first block of a stage only */
    const float* fpn_inner_w[4]; const float* fpn_inner_b[4];     /* fpn_inner1..4 (levels 4x, 8x, 16x, 32x) */
    const float* fpn_layer_w[4]; const float* fpn_layer_b[4];
} StemsegEncoderWeights;

size_t stemseg_hip_encoder_workspace_bytes(const StemsegEncoderDesc* desc);
/* Debugging aid: float offsets of the encoder plan's buffers inside the workspace (S0, X1, A, B, Cst[4], M1[4], M2, DS, XS, L[4], FO[4],
 * SK, total: 25 values, -1 = absent). */
int stemseg_hip_encoder_plan_offsets(const StemsegEncoderDesc* desc, int64_t* out25);

/* The stem alone (resnet.py:285-304 up to the ReLU; FrozenBN folded into w / bias): frames float32 [T][3][H][W] ->
 * out float32 [64][T][H/2][W/2] = relu(conv7x7 stride 2 pad 3 + bias).  w_tap_major: [3*7*7][64] (tap-major: the folded
 * [64][3][7][7] weight reshaped to [64][147] and transposed).  What stemseg_hip_encoder_forward runs first; exported for tests
 * and for the co-residency probe (tools/stem_corun_probe.py). */
int stemseg_hip_stem_conv(const float* frames, const float* w_tap_major, const float* bias, float* out, int32_t T, int32_t H, int32_t W,
                          void* stream);
int stemseg_hip_encoder_init_workspace(const StemsegEncoderDesc* desc, void* workspace, size_t ws_bytes, void* stream);
/* the encoder's twin of stemseg_hip_decoder_check_workspace */
int stemseg_hip_encoder_check_workspace(const StemsegEncoderDesc* desc, const void* workspace, size_t ws_bytes, int32_t* n_bad_host,
                                        int64_t* first_bad_host, void* stream);
/* frames: dense [T][3][H][W] (BGR, mean-subtracted).  out[4 * c + k], c < n_clips, k = 0..3: the four FPN maps (4x, 8x, 16x,
 * 32x) of clip c as volumes [256][clip frames][H/s][W/s] -- dense, or the interior view of the decoders' zero-haloed
 * inputs (then no copy is needed). */
int stemseg_hip_encoder_forward(const StemsegEncoderDesc* desc, const StemsegEncoderWeights* weights, const float* frames,
                                const StemsegVolume* out, void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Clustering side.
 * ---------------------------------------------------------------------------------------------- */

/* inference/main.py:93-103: per-frame running sum of seediness over the clips that contain the frame,
 * then (sum / count) > thr.  accumulate: acc = (first ? 0 : acc) + plane, n floats. */
int stemseg_hip_seediness_accumulate(float* acc, const float* plane, int64_t n, int32_t first, void* stream);
int stemseg_hip_fg_mask(const float* acc, float count, float thr, uint8_t* mask, int64_t n, void* stream);
/* The same for every frame of a sequence in one launch: acc [F][HW] holds the per-frame sums (e.g. accumulated with
 * stemseg_hip_semseg_accumulate on the 1-channel seediness maps, clip after clip), counts [F] (device) the number of clips
 * that contain each frame; mask[f][p] = acc[f][p] / counts[f] > thr, 0 where counts[f] == 0 (inference/main.py:93-103). */
int stemseg_hip_fg_mask_frames(const float* acc, const float* counts, float thr, uint8_t* mask, int32_t F, int64_t HW,
                               void* stream);

/* online_chainer.py:11-22 + :258-281 : compact the foreground voxels of a clip.
 * emb [E][V], bw [Ev][V], seed [V] dense with V = T*HW, fg uint8 [V].  Point order = flat voxel order
 * (frame-major, row-major).  Outputs: emb_out [N][E], bw_out [N][Ev], seed_out [N], voxel_index [N] (int32),
 * frame_offsets [T+1] int64 (exclusive prefix of per-frame counts; frame_offsets[T] = N).  Outputs must be
 * sized for N = V.  scratch: >= 8 * (V/1024 + 2) bytes.  No host synchronisation. */
int stemseg_hip_fg_gather(const float* emb, const float* bw, const float* seed, const uint8_t* fg,
                          int32_t E, int32_t Ev, int32_t T, int64_t HW,
                          float* emb_out, float* bw_out, float* seed_out, int32_t* voxel_index,
                          int64_t* frame_offsets, void* scratch, void* stream);

typedef struct StemsegClusterParams {
    float   primary_prob_thresh;    /* 0.5  (clusterers.py:40)  */
    float   secondary_prob_thresh;  /* 0.3                      */
    float   min_seediness_prob;     /* 0.8 / 0.95 (KITTI)       */
    int32_t max_instances;          /* 20                       */
    int32_t n_free_dims;
    float   free_dim_bandwidths[STEMSEG_MAX_EMB_DIMS];   /* 1/std^2, computed by the caller in fp32 */
} StemsegClusterParams;

/* read-back record, written by the last kernel into caller-provided DEVICE memory (copy it to the host
 * when convenient; it is what online_chainer.py needs to continue: K and the instance list). */
typedef struct StemsegClusterMeta {
    int32_t K;                      /* instances found; instance_labels = label_start + 0..K-1 */
    int32_t exhausted;              /* loop ran max_instances rounds (stale-mask quirk active) */
    int64_t n_points;
    int64_t n_unassigned_last;      /* num_unassigned_pts at the last evaluated loop header     */
    float   centers[STEMSEG_MAX_INSTANCES][STEMSEG_MAX_EMB_DIMS];
    float   bandwidths[STEMSEG_MAX_INSTANCES][STEMSEG_MAX_EMB_DIMS];   /* cat(bw_seed, free) ; std = sqrt(clamp(1/bw)) */
    float   seed_prob[STEMSEG_MAX_INSTANCES];
} StemsegClusterMeta;

size_t stemseg_hip_cluster_workspace_bytes(int64_t n_max);
/* SequentialClustering._process (inference/clusterers.py:60-166) including its quirks.
 * emb [N][E], bw [N][Ev], seed [N]; E = Ev + n_free_dims <= 8.  n_points_dev: optional device pointer to the
 * actual N (e.g. frame_offsets + T from stemseg_hip_fg_gather) -- if NULL, N = n_max.
 * labels [n_max] int64 (-1 = unassigned).  opt_masks: NULL or uint8 [max_instances][n_max] primary match masks;
 * opt_probs: NULL or float [max_instances][n_max] per-round probabilities (0 where unavailable).
 * Enqueues max_instances + 2 kernels; no host synchronisation. */
int stemseg_hip_cluster(const float* emb, const float* bw, const float* seed, int64_t n_max,
                        const int64_t* n_points_dev, int32_t E, int32_t Ev,
                        const StemsegClusterParams* params, int64_t label_start,
                        int64_t* labels, StemsegClusterMeta* meta_dev,
                        uint8_t* opt_masks, float* opt_probs,
                        void* workspace, size_t ws_bytes, void* stream);

/* The same for several independent point sets at once -- the clips of one step: every launch of the loop serves all of them
 * (grid.y = set), so a step's clustering costs max_instances + 2 launches instead of that per clip.  Sets are independent
 * (own workspace, own record); results are bit-identical to n_items separate stemseg_hip_cluster calls.  E, Ev and params are shared. */
typedef struct StemsegClusterItem {
    const float*   emb;              /* [n_max][E] */
    const float*   bw;               /* [n_max][Ev] */
    const float*   seed;             /* [n_max] */
    int64_t        n_max;
    const int64_t* n_points_dev;     /* optional device pointer to the actual N */
    int64_t        label_start;
    int64_t*       labels;           /* [n_max] */
    StemsegClusterMeta* meta_dev;
    uint8_t*       opt_masks;
    float*         opt_probs;
    void*          workspace;        /* >= stemseg_hip_cluster_workspace_bytes(n_max) */
    size_t         ws_bytes;
} StemsegClusterItem;
int stemseg_hip_cluster_batch(const StemsegClusterItem* items, int32_t n_items, int32_t E, int32_t Ev,
                              const StemsegClusterParams* params, void* stream);

/* online_chainer.py:291-343: label-pair statistics on the overlap frames.  lut_a / lut_b map (label + 1)
 * to a row / column index or -1 (ignore; the outlier label -1 maps through slot 0).  Outputs (int64,
 * zeroed by the call): inter [Ka][Kb], cnt_a [Ka], cnt_b [Kb].  No limit on Ka, Kb or on the label values
 * (the LUTs are indexed by id): small tables are counted in LDS, large ones with global atomics. */
int stemseg_hip_overlap_counts(const int64_t* labels_a, const int64_t* labels_b, int64_t n,
                               const int32_t* lut_a, int32_t lut_a_len, const int32_t* lut_b, int32_t lut_b_len,
                               int32_t Ka, int32_t Kb, int64_t* inter, int64_t* cnt_a, int64_t* cnt_b, void* stream);

/* online_chainer.py:304-308 (`unique()` of the overlap labels) and :43-49 (`highest id + 1`) in one pass:
 * present[id] = 1 for every label id in [0, cap) that occurs, present[cap] = 1 when a NEGATIVE label (the outlier id -1) occurs
 * -- `present` holds cap + 1 bytes; the reference's id enumeration order depends on whether -1 is in the set (online_chainer.py:308-309,
 * see reference_id_order) --, *max_plus_1 = max(label) + 1 over labels >= 0 (0 when there is none; ids >= cap still count towards the maximum).  accumulate = 0 zeroes both outputs
 * first; accumulate = 1 adds to what earlier calls left (several frames' label arrays -> one id set). */
int stemseg_hip_label_presence(const int64_t* labels, int64_t n, uint8_t* present, int32_t cap,
                               int64_t* max_plus_1, int32_t accumulate, void* stream);

/* in-place relabel: labels[i] = map[labels[i] + 1] for labels[i] + 1 in [0, map_len) (online_chainer.py:219-229) */
int stemseg_hip_relabel(int64_t* labels, int64_t n, const int64_t* map, int32_t map_len, void* stream);

/* ---- clip-parallel stitching: what a rank needs when the clips of ONE sequence are clustered on different GPUs
 * (the chain itself is online_chainer.py:193-236).  A clip clustered with label_start = 1 leaves ONE BYTE per
 * voxel -- 0 background, 1..K the clip-local instance (labels are i + label_start, clusterers.py:121), 255 the outlier label
 * -1 -- and only those planes are exchanged.  "bin" below maps a code to a table index: 0..B-2 as is, 255 -> B-1,
 * B = max_instances + 2. ---- */

/* masks_to_coord_list alone (online_chainer.py:11-22): voxel_index [N] + frame_offsets [T+1] of stemseg_hip_fg_gather, without
 * the head outputs.  scratch as for stemseg_hip_fg_gather. */
int stemseg_hip_fg_compact(const uint8_t* fg, int32_t T, int64_t HW, int32_t* voxel_index, int64_t* frame_offsets, void* scratch,
                           void* stream);

/* codes[0..V) = 0, then codes[voxel_index[i]] = labels[i] - label_start + 1 (255 for labels[i] < 0), i < N; N from
 * n_points_dev (device, e.g. frame_offsets + T) or n_max when NULL. */
int stemseg_hip_labels_to_codes(const int64_t* labels, const int32_t* voxel_index, const int64_t* n_points_dev, int64_t n_max,
                                int64_t label_start, uint8_t* codes, int64_t V, void* stream);

/* online_chainer.py:291-343's statistics for every (clip, frame) of a sequence in ONE launch.  codes: [planes][HW];
 * item k compares plane_a[k] (the frame as labelled by the clip that contributed it to the track container; -1: none, every
 * voxel counts under a = 0) with plane_b[k] (the same frame as labelled by the current clip) over the voxels that are
 * foreground in plane_b: tables[k][a][b] (int32, zeroed by the call) = #voxels with bin(code_a) = a, bin(code_b) = b.
 * plane_a / plane_b: DEVICE int32 [n_items]. */
int stemseg_hip_pair_tables(const uint8_t* codes, const int32_t* plane_a, const int32_t* plane_b, int32_t n_items, int64_t HW,
                            int32_t B, int32_t* tables, void* stream);

/* The relabel of online_chainer.py:219-224 as a gather: for item k = (src_begin, count, vbase, plane, out_begin) (DEVICE int64
 * [n_items][5]): out[out_begin + q] = lut[k][bin(codes[plane][voxel_index[src_begin + q] - vbase])], q < count.
 * lut: DEVICE int64 [n_items][B] (final track id per clip-local code).  max_count = max over items of count. */
int stemseg_hip_codes_to_labels(const uint8_t* codes, const int32_t* voxel_index, const int64_t* items, int32_t n_items,
                                int64_t max_count, const int64_t* lut, int64_t HW, int32_t B, int64_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Semantic-segmentation side (datasets with MODEL.USE_SEMSEG_HEAD: youtube_vis.yaml, kitti_mots_*.yaml).
 * ---------------------------------------------------------------------------------------------- */
#define STEMSEG_MAX_CLIP_FRAMES 32
#define STEMSEG_SEMSEG_NONE 0     /* foreground probability only                                   */
#define STEMSEG_SEMSEG_LOGITS 1   /* multiclass: mean class logits, float [F][C-1][HW]            */
#define STEMSEG_SEMSEG_PROBS 2    /* multiclass: softmax over the classes, float [F][C-1][HW]     */
#define STEMSEG_SEMSEG_ARGMAX 3   /* multiclass: class index, int64 [F][HW]                       */

/* inference_model.py:121-128: acc[frame_index[t]][c][:] += clip_logits[c][t][:] for the T slots of one clip.
 * acc: [n_frames][C][HW], ZERO-initialised by the caller before the first clip (the reference starts each frame's sum from
 * the float 0., :80); clip_logits: the semseg decoder's output [C][T][HW]; frame_index: HOST array of T frame numbers, -1 =
 * skip the slot; the non-negative entries of one call must be distinct (a clip that repeats a frame, inference/main.py:37-39,
 * is accumulated with one call per repetition so that the per-frame order of the fp32 adds stays slot order, :126-128). */
int stemseg_hip_semseg_accumulate(float* acc, const float* clip_logits, int32_t C, int32_t T, int64_t HW,
                                  const int32_t* frame_index, int32_t n_frames, void* stream);

/* InferenceModel.get_semseg_masks (inference_model.py:197-231) on the accumulated sums: mean = acc / counts[f] (device
 * float [F]); C > 2: fg = sigmoid(mean[C-1]), multiclass from mean[0..C-2] per output_type; C == 2: fg = softmax(mean)[1]
 * and no multiclass output (the reference then fails on `[].cpu()`, :231 -- here the buffer is simply left untouched).
 * fg: float [F][HW]. */
int stemseg_hip_semseg_masks(const float* acc, const float* counts, int32_t F, int32_t C, int64_t HW, int32_t output_type,
                             float* fg, void* multiclass, void* stream);

/* The same for ONE clip on its own (independent clips: every frame belongs to exactly one clip, the mean over clips is the
 * clip's own logits): clip_logits [C][T][HW] -> fg_prob float [T][HW] (may be NULL) and fg_mask uint8 [T][HW] = fg_prob > thr
 * (may be NULL) -- inference_model.py:197-231 + inference/main.py:142-144 without the [F][C][HW] accumulator round trip. */
int stemseg_hip_semseg_fg_clip(const float* clip_logits, int32_t C, int32_t T, int64_t HW, float thr, float* fg_prob,
                               uint8_t* fg_mask, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pre-processing (inference_image_loader.py:23-43, data/common.py:12-30, structures/image_list.py:93-104), one launch:
 * frames uint8 [T][H0][W0][3] (device) -> bilinear resize to (new_h, new_w), align_corners=False -> optional / 255 ->
 * (x - mean[c]) / std[c] -> optional channel flip (RGB models) -> out float [T][3][pad_h][pad_w], zero right / bottom padding.
 * mean / std are HOST arrays of 3 floats in the frames' channel order.
 * ---------------------------------------------------------------------------------------------- */
int stemseg_hip_preprocess_frames(const uint8_t* frames, int32_t T, int32_t H0, int32_t W0, int32_t new_h, int32_t new_w,
                                  int32_t pad_h, int32_t pad_w, const float mean[3], const float std[3], int32_t unit_scale,
                                  int32_t flip_channels, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Mask materialisation for the writers (output_utils/davis.py:76-110; same chain in youtube_vis.py:118-155 and
 * kitti_mots.py:89-130).  File formats stay on the host.
 * ---------------------------------------------------------------------------------------------- */

/* davis.py:76-77: dense [H][W] uint8 := 0, then dense[ys[i]][xs[i]] = lut[labels[i] + 1] -- lut maps (track label + 1) to
 * "index in instances_to_keep + 1" (1..255) or 0 for labels that are not kept (outliers, tracks beyond max_tracks). */
int stemseg_hip_scatter_instance_index(const int64_t* ys, const int64_t* xs, const int64_t* labels, int64_t n, const int32_t* lut,
                                       int32_t lut_len, uint8_t* dense, int32_t H, int32_t W, void* stream);

/* davis.py:79-110 fused: one-hot planes of `dense` [h][w] -> bilinear x mask_scale (align_corners=False) -> crop to
 * (crop_h, crop_w) = the resized network input without its zero padding -> bilinear resize to (out_h, out_w) -> > 0.5 ->
 * out [out_h][out_w] uint8 = kept-instance index + 1 (0 = none).  mask_scale = 1 reproduces `upscaled_inputs`. */
int stemseg_hip_resample_instance_masks(const uint8_t* dense, int32_t h, int32_t w, float mask_scale, int32_t crop_h, int32_t crop_w,
                                        int32_t out_h, int32_t out_w, uint8_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STEMSEG_HIP_H */

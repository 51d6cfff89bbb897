// Host-side orchestration of the 3-D squeeze-expand decoder (embedding and seediness variants) and the
// library's bookkeeping entry points (version, error string, geometry).
//
// Reference: SqueezingExpandDecoder.forward, /root/reference/stemseg/modeling/embedding_decoder.py:101-145 and
// seediness_decoder.py:92-112; topology table common.py:8-35.  The reference materialises every intermediate
// (conv out, GN out, ReLU in place, pool out, interpolate out, cat copy); here each stage writes straight into
// the layout its consumer reads:
//     conv (MFMA)  -> dense scratch D
//     GN stats     <- D
//     GN+ReLU+pool <- D -> zero-haloed input of the next conv | channel slice of the concat buffer
//     trilinear up -> the other channel slice of the concat buffer          (no torch.cat copy)
//     1x1x1 fuse   <- concat buffer viewed as [C][V]
//     heads        <- X4, one read, all activations + grid + bandwidth fused
// All buffers live in one caller-provided workspace whose halos are zeroed once.
#include "common.h"
#include <algorithm>
#include <mutex>
#include <vector>

namespace stemseg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- profiler ----------------------------------------------------------------------------------------
struct ProfRec { hipEvent_t a, b; int tag; double work; };
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;
static bool g_prof_on = false;
constexpr size_t PROF_MAX = 1 << 16;

void* profile_begin(int tag, double work, hipStream_t s) {
    if (!g_prof_on) return nullptr;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.size() >= PROF_MAX) return nullptr;
    ProfRec r;
    r.tag = tag; r.work = work;
    if (hipEventCreate(&r.a) != hipSuccess) return nullptr;
    if (hipEventCreate(&r.b) != hipSuccess) { (void)hipEventDestroy(r.a); return nullptr; }
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    return reinterpret_cast<void*>(g_prof.size());   // index + 1
}
void profile_end(void* handle, hipStream_t s) {
    if (!handle) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const size_t i = reinterpret_cast<size_t>(handle) - 1;
    if (i < g_prof.size()) (void)hipEventRecord(g_prof[i].b, s);
}

// ---- workspace canaries (common.h) ----------------------------------------------------------------------
__global__ void canary_fill_kernel(float* ws, const GuardList g) {
    reinterpret_cast<uint32_t*>(ws + g.off[blockIdx.x])[threadIdx.x] = WS_CANARY;
}
__global__ void canary_check_kernel(const float* ws, const GuardList g, unsigned long long* res) {      // res[0] = bad words, res[1] = min bad offset
    const int64_t o = g.off[blockIdx.x] + threadIdx.x;
    if (reinterpret_cast<const uint32_t*>(ws)[o] != WS_CANARY) {
        atomicAdd(&res[0], 1ull);
        atomicMin(&res[1], (unsigned long long)o);
    }
}
int launch_canary_fill(float* ws, const GuardList& g, hipStream_t s) {
    if (g.n == 0) return STEMSEG_OK;
    hipLaunchKernelGGL(canary_fill_kernel, dim3(g.n), dim3(WS_GUARD_FLOATS), 0, s, ws, g);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}
int canary_check(const float* ws, const GuardList& g, int32_t* n_bad_host, int64_t* first_bad_host, hipStream_t s) {
    SS_CHECK_ARG(n_bad_host && first_bad_host, "check_workspace: null result pointer");
    unsigned long long* res = nullptr;
    unsigned long long host[2] = {0ull, ~0ull};
    SS_HIP(hipMalloc(&res, sizeof(host)));
    hipError_t e = hipMemcpyAsync(res, host, sizeof(host), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && g.n > 0) {
        hipLaunchKernelGGL(canary_check_kernel, dim3(g.n), dim3(WS_GUARD_FLOATS), 0, s, ws, g, res);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, res, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(res);
    if (e != hipSuccess) {
        set_error("check_workspace failed: %s", hipGetErrorString(e));
        return STEMSEG_E_HIP;
    }
    *n_bad_host = (int32_t)std::min<unsigned long long>(host[0], 0x7fffffffull);
    *first_bad_host = host[0] ? (int64_t)host[1] : -1;
    return STEMSEG_OK;
}

struct DecoderPlan {
    GuardList guards;              // guard blocks of ONE clip's plan (offsets from that clip's base)
    GuardList tail_guards;         // guard blocks of the shared split-K scratch behind the clip plans (offsets from the workspace base)
    int nb;                        // clips per call (>= 1); clip c's plan starts at c * clip_floats
    int64_t clip_floats;
    int64_t out_bs;                // floats between the clips' outputs
    int cin, c32, c16, c8, c4, T, G;
    int h[4], w[4];                 // 32x, 16x, 8x, 4x
    int Ta1, Ta2, Ta3, Tb1, Tb2, Tc1, T16, T8;
    // workspace offsets in floats
    int64_t pin[4], D[4], S[4], Sfloats[4], P32b, P32c, X32, cat16, P16b, X16, cat8, X8, cat4, X4, stats[4], gn_scratch[4], total;
};

static inline int pooled(int T, int on) { return on ? (T + 1) / 2 : T; }

static int make_plan(const StemsegDecoderDesc* d, DecoderPlan& p) {
    SS_CHECK_ARG(d, "decoder: null descriptor");
    SS_CHECK_ARG(d->struct_bytes == (int32_t)sizeof(StemsegDecoderDesc), "decoder: descriptor size mismatch (%d vs %d): ABI skew",
                 d->struct_bytes, (int)sizeof(StemsegDecoderDesc));
    SS_CHECK_ARG(d->T >= 1 && d->H4 >= 8 && d->W4 >= 8 && d->H4 % 8 == 0 && d->W4 % 8 == 0,
                 "decoder: T=%d H4=%d W4=%d (H4, W4 must be positive multiples of 8)", d->T, d->H4, d->W4);
    SS_CHECK_ARG(d->in_channels % 4 == 0 && d->in_channels > 0, "decoder: in_channels %% 4");
    SS_CHECK_ARG(d->gn_groups >= 0 && d->gn_groups <= 64, "decoder: gn_groups %d (0 = no normalisation layer, else 1..64)", d->gn_groups);
    for (int i = 0; i < 4; ++i) SS_CHECK_ARG(d->inter[i] > 0 && d->inter[i] % 32 == 0 && (d->gn_groups == 0 || d->inter[i] % d->gn_groups == 0),
                                             "decoder: inter[%d]=%d must be a multiple of 32 and of gn_groups", i, d->inter[i]);
    for (int i = 0; i < 3; ++i) SS_CHECK_ARG(d->pool[i] >= 0 && d->pool[i] <= 2, "decoder: pool[%d]=%d (0 none, 1 AvgPool3d, 2 MaxPool3d)", i, d->pool[i]);
    SS_CHECK_ARG((d->n_out >= 1 && d->n_out <= STEMSEG_MAX_HEAD_OUT) || (d->n_out <= 256 && d->n_out % 32 == 0),
                 "decoder: n_out=%d (1..%d through the fused heads kernel, or a multiple of 32 <= 256 through the 1x1x1 MFMA conv)", d->n_out, STEMSEG_MAX_HEAD_OUT);
    SS_CHECK_ARG(d->input_layout >= 0 && d->input_layout <= 2, "decoder: input_layout");
    p.cin = d->in_channels; p.c32 = d->inter[0]; p.c16 = d->inter[1]; p.c8 = d->inter[2]; p.c4 = d->inter[3];
    p.T = d->T; p.G = d->gn_groups;
    SS_CHECK_ARG(d->n_clips >= 0 && d->n_clips <= 4096, "decoder: n_clips=%d", d->n_clips);
    p.nb = d->n_clips > 1 ? d->n_clips : 1;
    for (int i = 0; i < 4; ++i) { p.h[i] = d->H4 >> (3 - i); p.w[i] = d->W4 >> (3 - i); }
    const int64_t out_dense = (int64_t)d->n_out * d->T * d->H4 * d->W4;
    p.out_bs = (p.nb > 1 && d->out_clip_stride > 0) ? d->out_clip_stride : out_dense;
    SS_CHECK_ARG(p.out_bs >= out_dense && p.out_bs % 4 == 0, "decoder: out_clip_stride %lld (>= %lld, a multiple of 4)", (long long)p.out_bs, (long long)out_dense);
    if (p.nb > 1 && d->input_layout == 2)
        for (int i = 0; i < 4; ++i)
            SS_CHECK_ARG(d->feat_clip_stride[i] >= PaddedGeom(d->in_channels, d->T, p.h[i], p.w[i]).total && d->feat_clip_stride[i] % 4 == 0,
                         "decoder: feat_clip_stride[%d]=%lld (>= one zero-haloed buffer, a multiple of 4)", i, (long long)d->feat_clip_stride[i]);
    p.Ta1 = pooled(p.T, d->pool[0]); p.Ta2 = pooled(p.Ta1, d->pool[1]); p.Ta3 = pooled(p.Ta2, d->pool[2]);
    p.Tb1 = pooled(p.T, d->pool[0]); p.Tb2 = pooled(p.Tb1, d->pool[1]);
    p.Tc1 = pooled(p.T, d->pool[0]);
    p.T16 = p.Ta3 * d->t_scale[0];
    p.T8 = p.T16 * d->t_scale[1];
    SS_CHECK_ARG(p.T16 == p.Tb2 && p.T8 == p.Tc1 && p.T8 * d->t_scale[2] == p.T,
                 "decoder: pool / t_scale tables inconsistent for T=%d (32x->%d*%d vs 16x %d; ->%d vs 8x %d; ->%d vs %d)", p.T, p.Ta3,
                 d->t_scale[0], p.Tb2, p.T8, p.Tc1, p.T8 * d->t_scale[2], p.T);
    int64_t off = 0;
    p.guards.n = 0;
    auto take = [&](int64_t floats) {            // a slice + its guard block
        int64_t o = off;
        off += round_up(floats, 64);
        p.guards.push(off);
        off += WS_GUARD_FLOATS;
        return o;
    };
    for (int i = 0; i < 4; ++i) p.pin[i] = (d->input_layout == 2) ? -1 : take(PaddedGeom(p.cin, p.T, p.h[i], p.w[i]).total);
    // one dense conv-output scratch per branch: the four branches run concurrently on their own streams
    p.D[0] = take((int64_t)p.c32 * p.T * p.h[0] * p.w[0]);
    p.D[1] = take((int64_t)p.c16 * p.T * p.h[1] * p.w[1]);
    p.D[2] = take((int64_t)p.c8 * p.T * p.h[2] * p.w[2]);
    p.D[3] = take((int64_t)p.c4 * p.T * p.h[3] * p.w[3]);
    // split-K partial-sum scratch: small-map branches up to 16 / 4 / 2 slabs; the 4x branch splits only the rows of its
    // tail (launch_rows_balanced), 2 slabs' worth covers any (rows, k) the cost model picks
    p.Sfloats[0] = 16 * (int64_t)p.c32 * p.T * p.h[0] * p.w[0];
    p.Sfloats[1] = 4 * (int64_t)p.c16 * p.T * p.h[1] * p.w[1];
    p.Sfloats[2] = 4 * (int64_t)p.c8 * p.T * p.h[2] * p.w[2];
    p.Sfloats[3] = 2 * (int64_t)p.c4 * p.T * p.h[3] * p.w[3];
    p.P32b = take(PaddedGeom(p.c32, p.Ta1, p.h[0], p.w[0]).total);
    p.P32c = take(PaddedGeom(p.c32, p.Ta2, p.h[0], p.w[0]).total);
    p.X32 = take((int64_t)p.c32 * p.Ta3 * p.h[0] * p.w[0]);
    p.cat16 = take((int64_t)(p.c32 + p.c16) * p.T16 * p.h[1] * p.w[1]);
    p.P16b = take(PaddedGeom(p.c16, p.Tb1, p.h[1], p.w[1]).total);
    p.X16 = take((int64_t)p.c16 * p.T16 * p.h[1] * p.w[1]);
    p.cat8 = take((int64_t)(p.c16 + p.c8) * p.T8 * p.h[2] * p.w[2]);
    p.X8 = take((int64_t)p.c8 * p.T8 * p.h[2] * p.w[2]);
    p.cat4 = take((int64_t)(p.c8 + p.c4) * p.T * p.h[3] * p.w[3]);
    p.X4 = take((int64_t)p.c4 * p.T * p.h[3] * p.w[3]);
    for (int i = 0; i < 4; ++i) {
        p.stats[i] = take(2 * 64);
        p.gn_scratch[i] = take(2 * gn_scratch_doubles(64 * 8, p.G > 0 ? p.G : 1));   // doubles (2 floats each): the decoder's OWN group count x GN_SLOT_CAP x 2
    }
    p.clip_floats = round_up(off, 256);
    // the split-K scratch of the four branches, shared by the clips of a call (a batched launch keeps clip c's slabs behind clip c - 1's):
    // nb times one clip's worth, behind the clip plans
    off = p.clip_floats * p.nb;
    p.tail_guards.n = 0;
    for (int i = 0; i < 4; ++i) {
        p.Sfloats[i] *= p.nb;
        p.S[i] = off;
        off += round_up(p.Sfloats[i], 64);
        p.tail_guards.push(off);
        off += WS_GUARD_FLOATS;
    }
    p.total = off;
    SS_CHECK_ARG(!p.guards.full && !p.tail_guards.full, "decoder: the plan has more than %d workspace slices (GuardList)", WS_MAX_GUARDS);
    return STEMSEG_OK;
}

// ---- internal branch streams: the 32x / 16x / 8x branches of a decoder are independent until the fuse convs ----
constexpr int STREAM_SETS = 4;
struct BranchStreams {
    bool ok = false;
    hipStream_t s[4];            // 32x, 16x, 8x branches + (detached mode) the 4x branch / tail
    hipEvent_t start, done[4];
};
static BranchStreams g_streams[16][STREAM_SETS];   // [device][set]
static std::mutex g_streams_mu;

static BranchStreams* get_streams(int set) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(g_streams_mu);
    BranchStreams& b = g_streams[dev][((set % STREAM_SETS) + STREAM_SETS) % STREAM_SETS];
    if (!b.ok) {
        for (int i = 0; i < 4; ++i) {
            if (hipStreamCreateWithFlags(&b.s[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&b.done[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (hipEventCreateWithFlags(&b.start, hipEventDisableTiming) != hipSuccess) return nullptr;
        b.ok = true;
    }
    return &b;
}

// channel slice [c0, c0+C) of a dense [Ctot][T][H][W] buffer
static inline StemsegVolume slice_volume(float* base, int c0, int C, int T, int H, int W) {
    const int64_t cs = (int64_t)T * H * W;
    return make_volume(base + (int64_t)c0 * cs, cs, (int64_t)H * W, W, C, T, H, W, (int64_t)C * cs);
}
// dense [C][V] viewed as a 1-row volume for the 1x1x1 convolution
static inline StemsegVolume flat_volume(float* base, int C, int64_t V) {
    return make_volume(base, V, 0, 0, C, 1, 1, (int)V, (int64_t)C * V);
}

// one conv -> GroupNorm -> ReLU (-> pool) stage for the nb clips of the call: the input volumes lie in_bs floats apart, everything inside
// the workspace (D, stats, scratch, dst) ws_bs floats
static int conv_gn(const StemsegVolume& in_halo, const float* w, const float* b, const float* gw, const float* gb, int Cout, int T, int H,
                   int W, int pool, const StemsegVolume& dst, float* D, float* stats, double* scratch, int G, float eps, hipStream_t s,
                   float* splitk, int64_t splitk_floats, int precision, int nb, int64_t in_bs, int64_t ws_bs, bool apply = true) {
    StemsegVolume d = dense_volume(D, Cout, T, H, W);
    ConvEpilogue e;
    e.precision = precision;
    e.nb = nb; e.in_bs = in_bs; e.out_bs = ws_bs; e.gn_bs = ws_bs / 2;
    ClipBatch cb;
    cb.nb = nb; cb.in_bs = ws_bs; cb.out_bs = ws_bs; cb.stats_bs = ws_bs;
    if (G == 0) {      // NORMALIZATION_LAYER 'none' (model_builder.py:29-33): conv -> ReLU -> pool; gw / gb are ones / zeros from the caller
        int rc0 = launch_conv3d(in_halo, w, b, d, 3, 3, 3, 0, s, splitk, splitk_floats, &e);
        for (int c = 0; c < nb && !rc0; ++c) rc0 = launch_gn_identity_stats(stats + c * ws_bs, 1, s);
        if (rc0 || !apply) return rc0;      // (!apply: the consumer normalises as it reads D -- the heads of the linear tail)
        return launch_gn_relu_pool(D, Cout, T, H, W, 1, stats, gw, gb, pool, dst, s, cb);
    }
    // the conv's epilogue (or its split-K reduce) leaves the GroupNorm partial sums: its output is not read again for them
    int rc = launch_conv3d_gn(in_halo, w, b, d, 3, 3, 3, 0, s, splitk, splitk_floats, &e, G, eps, stats, scratch, ws_bs);
    if (rc || !apply) return rc;
    return launch_gn_relu_pool(D, Cout, T, H, W, G, stats, gw, gb, pool, dst, s, cb);
}

}  // namespace stemseg

using namespace stemseg;

extern "C" int stemseg_hip_version(void) { return STEMSEG_HIP_ABI_VERSION; }
extern "C" const char* stemseg_hip_last_error(void) { return g_err; }
extern "C" int stemseg_hip_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return STEMSEG_E_HIP;
    }
    return n;
}

extern "C" int stemseg_hip_profile_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return STEMSEG_OK;
}

// Synchronises the device, then sums (elapsed ms, work, launches) per tag into out[tag*3 + {0,1,2}] and clears the log.
extern "C" int stemseg_hip_profile_read(double* out_host, int32_t n_tags) {
    SS_CHECK_ARG(out_host && n_tags > 0, "profile_read: bad arguments");
    SS_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < n_tags * 3; ++i) out_host[i] = 0.0;
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.tag >= 0 && r.tag < n_tags) {
            out_host[r.tag * 3 + 0] += ms;
            out_host[r.tag * 3 + 1] += r.work;
            out_host[r.tag * 3 + 2] += 1.0;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_padded_geometry(int32_t C, int32_t T, int32_t H, int32_t W, int64_t out[5]) {
    SS_CHECK_ARG(out && C > 0 && T > 0 && H > 0 && W > 0, "padded_geometry: bad arguments");
    PaddedGeom g(C, T, H, W);
    out[0] = g.pitch; out[1] = g.ts; out[2] = g.cs; out[3] = g.total; out[4] = g.interior;
    return STEMSEG_OK;
}

extern "C" size_t stemseg_hip_decoder_workspace_bytes(const StemsegDecoderDesc* desc) {
    DecoderPlan p;
    if (make_plan(desc, p) != STEMSEG_OK) return 0;
    return (size_t)p.total * sizeof(float);
}

extern "C" int stemseg_hip_decoder_init_workspace(const StemsegDecoderDesc* desc, void* workspace, size_t ws_bytes, void* stream) {
    DecoderPlan p;
    int rc = make_plan(desc, p);
    if (rc) return rc;
    SS_CHECK_ARG(workspace && (reinterpret_cast<uintptr_t>(workspace) % 256 == 0), "decoder: workspace must be 256-byte aligned");
    if (ws_bytes < (size_t)p.total * sizeof(float)) {
        set_error("decoder: workspace too small (%zu < %zu bytes)", ws_bytes, (size_t)p.total * sizeof(float));
        return STEMSEG_E_WORKSPACE;
    }
    SS_HIP(hipMemsetAsync(workspace, 0, (size_t)p.total * sizeof(float), as_stream(stream)));
    for (int c = 0; c < p.nb; ++c) {
        rc = launch_canary_fill(reinterpret_cast<float*>(workspace) + c * p.clip_floats, p.guards, as_stream(stream));
        if (rc) return rc;
    }
    return launch_canary_fill(reinterpret_cast<float*>(workspace), p.tail_guards, as_stream(stream));
}

extern "C" int stemseg_hip_decoder_check_workspace(const StemsegDecoderDesc* desc, const void* workspace, size_t ws_bytes, int32_t* n_bad_host,
                                                   int64_t* first_bad_host, void* stream) {
    DecoderPlan p;
    int rc = make_plan(desc, p);
    if (rc) return rc;
    SS_CHECK_ARG(workspace && ws_bytes >= (size_t)p.total * sizeof(float) && n_bad_host && first_bad_host, "decoder_check_workspace: bad workspace");
    int32_t bad = 0;
    int64_t first = -1;
    for (int c = 0; c <= p.nb; ++c) {                      // every clip's plan, then the shared tail
        int32_t n = 0;
        int64_t f = -1;
        const int64_t base = c < p.nb ? c * p.clip_floats : 0;
        rc = canary_check(reinterpret_cast<const float*>(workspace) + base, c < p.nb ? p.guards : p.tail_guards, &n, &f, as_stream(stream));
        if (rc) return rc;
        if (n && first < 0) first = base + f;
        bad += n;
    }
    *n_bad_host = bad;
    *first_bad_host = first;
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_decoder_forward(const StemsegDecoderDesc* desc, const StemsegDecoderWeights* wts, const float* const feats[4],
                                           float* out, void* workspace, size_t ws_bytes, void* stream) {
    DecoderPlan p;
    int rc = make_plan(desc, p);
    if (rc) return rc;
    SS_CHECK_ARG(wts && feats && out && workspace, "decoder_forward: null pointer");
    if (ws_bytes < (size_t)p.total * sizeof(float)) {
        set_error("decoder: workspace too small (%zu < %zu bytes)", ws_bytes, (size_t)p.total * sizeof(float));
        return STEMSEG_E_WORKSPACE;
    }
    for (int i = 0; i < 7; ++i) SS_CHECK_ARG(wts->conv_w[i] && wts->conv_b[i] && wts->gn_w[i] && wts->gn_b[i], "decoder_forward: null block weight %d", i);
    // fuse_w[2] NULL: conv_4 folded into the heads; all three NULL: the whole linear tail folded into per-level head matrices (below)
    const bool lin_tail = !wts->fuse_w[0] && !wts->fuse_w[1] && !wts->fuse_w[2];
    if (!lin_tail) for (int i = 0; i < 2; ++i) SS_CHECK_ARG(wts->fuse_w[i], "decoder_forward: null fuse weight %d", i);
    SS_CHECK_ARG(!lin_tail || desc->n_out <= STEMSEG_MAX_HEAD_OUT, "decoder_forward: the linear tail serves the fused heads kernel (n_out <= %d), not the wide head", STEMSEG_MAX_HEAD_OUT);
    SS_CHECK_ARG(wts->head_w, "decoder_forward: null head weight");
    for (int i = 0; i < 4; ++i) SS_CHECK_ARG(feats[i], "decoder_forward: null feature map %d", i);

    hipStream_t s = as_stream(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    // clip batch: clip c's plan at ws + c * WS, its inputs at feats[i] + c * pin_bs[i], its output at out + c * p.out_bs; every launch
    // below covers all nb clips (the clip is a grid dimension of every kernel)
    const int nb = p.nb;
    const int64_t WS = p.clip_floats;
    ClipBatch wsb;                                          // a stage that reads and writes inside the workspace
    wsb.nb = nb; wsb.in_bs = WS; wsb.out_bs = WS; wsb.stats_bs = WS;
    SS_CHECK_ARG(desc->precision == STEMSEG_PRECISION_F32 || desc->precision == STEMSEG_PRECISION_BF16X6 || desc->precision == STEMSEG_PRECISION_F16X3,
                 "decoder: precision must be 0 (f32), 2 (bf16x6) or 3 (f16x3)");
    ConvEpilogue fuse_epi;
    fuse_epi.precision = desc->precision;
    fuse_epi.nb = nb; fuse_epi.in_bs = WS; fuse_epi.out_bs = WS;
    const float eps = desc->gn_eps;
    const int G = p.G, T = p.T;
    float* D[4]; float* stats[4]; double* scratch[4];
    for (int i = 0; i < 4; ++i) {
        D[i] = ws + p.D[i];
        stats[i] = ws + p.stats[i];
        scratch[i] = reinterpret_cast<double*>(ws + p.gn_scratch[i]);
    }
    // Branch streams: main stream runs the 4x branch (65 % of the FLOPs); 32x / 16x / 8x run beside it and join at
    // the fuse convolutions.  concurrency == 0 keeps everything on the caller's stream.
    BranchStreams* bs = desc->concurrency ? get_streams(desc->concurrency - 1) : nullptr;
    hipStream_t s32 = bs ? bs->s[0] : s, s16 = bs ? bs->s[1] : s, s8 = bs ? bs->s[2] : s;
    // detached: even the 4x branch and the tail run on an internal stream; the caller joins later with
    // stemseg_hip_decoder_join, after it has enqueued other work (the twin decoder) on its own stream
    const bool detached = bs && desc->detached;
    hipStream_t sm = detached ? bs->s[3] : s;

    // 0. inputs into the zero-haloed layout (skipped when the caller already provides it)
    float* pin[4];
    int64_t pin_bs[4];
    for (int i = 0; i < 4; ++i) {
        if (desc->input_layout == 2) { pin[i] = const_cast<float*>(feats[i]); pin_bs[i] = nb > 1 ? desc->feat_clip_stride[i] : 0; }
        else {
            pin[i] = ws + p.pin[i];
            pin_bs[i] = WS;
            const int64_t dense = (int64_t)p.cin * T * p.h[i] * p.w[i];
            const int64_t fs = (nb > 1 && desc->feat_clip_stride[i] > 0) ? desc->feat_clip_stride[i] : dense;
            SS_CHECK_ARG(fs >= dense, "decoder: feat_clip_stride[%d] smaller than one dense feature map", i);
            for (int c = 0; c < nb; ++c) {
                rc = launch_copy_to_volume(feats[i] + c * fs, desc->input_layout, padded_interior_view(pin[i] + c * WS, p.cin, T, p.h[i], p.w[i]), s);
                if (rc) return rc;
            }
        }
    }
    if (bs) {
        SS_HIP(hipEventRecord(bs->start, s));
        for (int i = 0; i < (detached ? 4 : 3); ++i) SS_HIP(hipStreamWaitEvent(bs->s[i], bs->start, 0));
    }
    // 1. block_32x on s32: three conv/GN/ReLU(/pool) stages (embedding_decoder.py:20-35), then upsample into cat16[0:c32]
    rc = conv_gn(padded_halo_view(pin[0], p.cin, T, p.h[0], p.w[0]), wts->conv_w[0], wts->conv_b[0], wts->gn_w[0], wts->gn_b[0], p.c32, T,
                 p.h[0], p.w[0], desc->pool[0], padded_interior_view(ws + p.P32b, p.c32, p.Ta1, p.h[0], p.w[0]), D[0], stats[0], scratch[0], G, eps, s32, ws + p.S[0], p.Sfloats[0], desc->precision, nb, pin_bs[0], WS);
    if (rc) return rc;
    rc = conv_gn(padded_halo_view(ws + p.P32b, p.c32, p.Ta1, p.h[0], p.w[0]), wts->conv_w[1], wts->conv_b[1], wts->gn_w[1], wts->gn_b[1], p.c32,
                 p.Ta1, p.h[0], p.w[0], desc->pool[1], padded_interior_view(ws + p.P32c, p.c32, p.Ta2, p.h[0], p.w[0]), D[0], stats[0], scratch[0], G, eps, s32, ws + p.S[0], p.Sfloats[0], desc->precision, nb, WS, WS);
    if (rc) return rc;
    rc = conv_gn(padded_halo_view(ws + p.P32c, p.c32, p.Ta2, p.h[0], p.w[0]), wts->conv_w[2], wts->conv_b[2], wts->gn_w[2], wts->gn_b[2], p.c32,
                 p.Ta2, p.h[0], p.w[0], desc->pool[2], dense_volume(ws + p.X32, p.c32, p.Ta3, p.h[0], p.w[0]), D[0], stats[0], scratch[0], G, eps, s32, ws + p.S[0], p.Sfloats[0], desc->precision, nb, WS, WS);
    if (rc) return rc;
    // THE LINEAR TAIL.  Between the last GroupNorm + ReLU of every branch and the heads' activations the reference applies only linear maps:
    // trilinear up-sampling, concatenation, the bias-free 1x1x1 fuse convs conv_16 / conv_8 / conv_4 (embedding_decoder.py:64-80,112-129) and the
    // 1x1x1 heads (:131-143).  Channel mixing commutes with up-sampling, so
    //   heads(x) = up(up(up(M32 x32) + M16 y16) + M8 y8) + M4 y4,   M4 = Wh W4b, M8 = Wh W4a W8b, M16 = Wh W4a W8a W16b, M32 = Wh W4a W8a W16a
    // (W4 = [W4a | W4b] over the (up-sampled, own) halves of its concat input, likewise W8, W16): with fuse_w[0..2] = NULL the caller hands over
    // head_w = [M32 | M16 | M8 | M4] (fp64 products, rounded once) and every level contributes its n_out-channel share AT ITS OWN RESOLUTION --
    // the up-samplings move n_out channels instead of 256 / 256 / 128, the three fuse convs and the concat halves they read disappear.  The same
    // function, within fp32 round-off of the step-by-step form (one rounding of the product matrices instead of one per stage).
    const int64_t V32 = (int64_t)p.Ta3 * p.h[0] * p.w[0];
    const int64_t V16l = (int64_t)p.T16 * p.h[1] * p.w[1], V8l = (int64_t)p.T8 * p.h[2] * p.w[2];
    const int NO = desc->n_out;
    const float* M32 = wts->head_w;
    const float* M16 = M32 + (int64_t)NO * p.c32;
    const float* M8 = M16 + (int64_t)NO * p.c16;
    const float* M4 = M8 + (int64_t)NO * p.c8;
    // n_out-channel level maps live in the (otherwise unused) X16 / X8 / X4 slices: [z32 | up16 | z16], [up8 | z8], [up4]
    float* z32 = ws + p.X16;
    float* up16 = z32 + (int64_t)NO * V16l;
    float* z16 = up16 + (int64_t)NO * V16l;
    float* up8 = ws + p.X8;
    float* z8 = up8 + (int64_t)NO * V8l;
    float* up4 = ws + p.X4;
    if (lin_tail) {
        SS_CHECK_ARG(3 * NO <= p.c16 && 2 * NO <= p.c8 && NO <= p.c4 && V32 <= V16l, "decoder: level maps do not fit the fuse slices");
        rc = launch_level_head(ws + p.X32, p.c32, V32, M32, NO, nullptr, z32, s32, nb, WS, 0, WS);
        if (rc) return rc;
        rc = launch_upsample(z32, NO, p.Ta3, p.h[0], p.w[0], desc->t_scale[0], 2, 2, dense_volume(up16, NO, p.T16, p.h[1], p.w[1]), s32, wsb);
    } else {
        rc = launch_upsample(ws + p.X32, p.c32, p.Ta3, p.h[0], p.w[0], desc->t_scale[0], 2, 2,
                             slice_volume(ws + p.cat16, 0, p.c32, p.T16, p.h[1], p.w[1]), s32, wsb);
    }
    if (rc) return rc;
    if (bs) SS_HIP(hipEventRecord(bs->done[0], s32));
    // 2. block_16x on s16 into cat16[c32:], join 32x, 1x1x1 fuse, upsample into cat8[0:c16]  (:112-117)
    rc = conv_gn(padded_halo_view(pin[1], p.cin, T, p.h[1], p.w[1]), wts->conv_w[3], wts->conv_b[3], wts->gn_w[3], wts->gn_b[3], p.c16, T,
                 p.h[1], p.w[1], desc->pool[0], padded_interior_view(ws + p.P16b, p.c16, p.Tb1, p.h[1], p.w[1]), D[1], stats[1], scratch[1], G, eps, s16, ws + p.S[1], p.Sfloats[1], desc->precision, nb, pin_bs[1], WS);
    if (rc) return rc;
    rc = conv_gn(padded_halo_view(ws + p.P16b, p.c16, p.Tb1, p.h[1], p.w[1]), wts->conv_w[4], wts->conv_b[4], wts->gn_w[4], wts->gn_b[4], p.c16,
                 p.Tb1, p.h[1], p.w[1], desc->pool[1], slice_volume(ws + p.cat16, p.c32, p.c16, p.T16, p.h[1], p.w[1]), D[1], stats[1], scratch[1], G, eps, s16, ws + p.S[1], p.Sfloats[1], desc->precision, nb, WS, WS);
    if (rc) return rc;
    const int64_t V16 = (int64_t)p.T16 * p.h[1] * p.w[1], V8 = (int64_t)p.T8 * p.h[2] * p.w[2], V4 = (int64_t)T * p.h[3] * p.w[3];
    if (bs) SS_HIP(hipStreamWaitEvent(s16, bs->done[0], 0));
    if (lin_tail) {
        rc = launch_level_head(ws + p.cat16 + (int64_t)p.c32 * V16, p.c16, V16, M16, NO, up16, z16, s16, nb, WS, WS, WS);
        if (rc) return rc;
        rc = launch_upsample(z16, NO, p.T16, p.h[1], p.w[1], desc->t_scale[1], 2, 2, dense_volume(up8, NO, p.T8, p.h[2], p.w[2]), s16, wsb);
    } else {
        rc = launch_conv3d(flat_volume(ws + p.cat16, p.c32 + p.c16, V16), wts->fuse_w[0], nullptr, flat_volume(ws + p.X16, p.c16, V16), 1, 1, 1, 0, s16, ws + p.S[1], p.Sfloats[1], &fuse_epi);
        if (rc) return rc;
        rc = launch_upsample(ws + p.X16, p.c16, p.T16, p.h[1], p.w[1], desc->t_scale[1], 2, 2,
                             slice_volume(ws + p.cat8, 0, p.c16, p.T8, p.h[2], p.w[2]), s16, wsb);
    }
    if (rc) return rc;
    if (bs) SS_HIP(hipEventRecord(bs->done[1], s16));
    // 3. block_8x on s8 into cat8[c16:], join 16x, fuse, upsample into cat4[0:c8]  (:119-123)
    rc = conv_gn(padded_halo_view(pin[2], p.cin, T, p.h[2], p.w[2]), wts->conv_w[5], wts->conv_b[5], wts->gn_w[5], wts->gn_b[5], p.c8, T, p.h[2],
                 p.w[2], desc->pool[0], slice_volume(ws + p.cat8, p.c16, p.c8, p.T8, p.h[2], p.w[2]), D[2], stats[2], scratch[2], G, eps, s8, ws + p.S[2], p.Sfloats[2], desc->precision, nb, pin_bs[2], WS);
    if (rc) return rc;
    if (bs) SS_HIP(hipStreamWaitEvent(s8, bs->done[1], 0));
    if (lin_tail) {
        rc = launch_level_head(ws + p.cat8 + (int64_t)p.c16 * V8, p.c8, V8, M8, NO, up8, z8, s8, nb, WS, WS, WS);
        if (rc) return rc;
        rc = launch_upsample(z8, NO, p.T8, p.h[2], p.w[2], desc->t_scale[2], 2, 2, dense_volume(up4, NO, T, p.h[3], p.w[3]), s8, wsb);
    } else {
        rc = launch_conv3d(flat_volume(ws + p.cat8, p.c16 + p.c8, V8), wts->fuse_w[1], nullptr, flat_volume(ws + p.X8, p.c8, V8), 1, 1, 1, 0, s8, ws + p.S[2], p.Sfloats[2], &fuse_epi);
        if (rc) return rc;
        rc = launch_upsample(ws + p.X8, p.c8, p.T8, p.h[2], p.w[2], desc->t_scale[2], 2, 2, slice_volume(ws + p.cat4, 0, p.c8, T, p.h[3], p.w[3]), s8, wsb);
    }
    if (rc) return rc;
    if (bs) SS_HIP(hipEventRecord(bs->done[2], s8));
    // 4. block_4x on the caller's stream into cat4[c8:], join 8x, fuse  (:125-129)
    rc = conv_gn(padded_halo_view(pin[3], p.cin, T, p.h[3], p.w[3]), wts->conv_w[6], wts->conv_b[6], wts->gn_w[6], wts->gn_b[6], p.c4, T, p.h[3],
                 p.w[3], 0, slice_volume(ws + p.cat4, p.c8, p.c4, T, p.h[3], p.w[3]), D[3], stats[3], scratch[3], G, eps, sm, ws + p.S[3], p.Sfloats[3], desc->precision, nb, pin_bs[3], WS,
                 /* the linear tail's heads are the only reader of the normalised 4x map: they normalise the raw conv output as they read it */ !lin_tail);
    if (rc) return rc;
    if (bs) SS_HIP(hipStreamWaitEvent(sm, bs->done[2], 0));
    // conv_4 (1x1x1, no bias, no activation: embedding_decoder.py:80,129) feeds nothing but the heads (1x1x1 as well): a caller that hands over
    // head weights ALREADY MULTIPLIED by conv_4's -- head_w = W_heads . W_conv4, [n_out][c8 + c4], fuse_w[2] = NULL -- gets the heads straight off
    // the concat buffer: the c4-channel map (106 MB per clip at 480p) is neither computed, written nor read back.  One linear map instead of two:
    // the same fold as FrozenBN into its convolution, results within fp32 round-off of the two-step form.
    const bool fold4 = wts->fuse_w[2] == nullptr;
    const float* head_in = lin_tail ? D[3] : (fold4 ? ws + p.cat4 : ws + p.X4);      // (linear tail: the 4x branch's RAW conv output, normalised on the fly)
    HeadsGN hgn;
    hgn.stats = stats[3]; hgn.gamma = wts->gn_w[6]; hgn.beta = wts->gn_b[6]; hgn.cpg = G > 0 ? p.c4 / G : p.c4; hgn.stats_bs = WS;
    const int head_cin = lin_tail ? p.c4 : (fold4 ? p.c8 + p.c4 : p.c4);
    if (!fold4) {
        rc = launch_conv3d(flat_volume(ws + p.cat4, p.c8 + p.c4, V4), wts->fuse_w[2], nullptr, flat_volume(ws + p.X4, p.c4, V4), 1, 1, 1, 0, sm, ws + p.S[3], p.Sfloats[3], &fuse_epi);
        if (rc) return rc;
    }
    // 5. heads (:131-143)
    if (desc->n_out > STEMSEG_MAX_HEAD_OUT) {
        // wide linear head (semseg_decoder.py:116: conv_out, class logits, no activation): the 1x1x1 MFMA conv; head_w is
        // then a PACKED conv weight with Cout = n_out (zero-padded to a multiple of 32 by the caller), head_b may be NULL
        ConvEpilogue head_epi = fuse_epi;
        head_epi.out_bs = p.out_bs;
        rc = launch_conv3d(flat_volume(const_cast<float*>(head_in), head_cin, V4), wts->head_w, wts->head_b, flat_volume(out, desc->n_out, V4), 1, 1, 1, 0, sm, nullptr, 0, &head_epi);
        if (rc) return rc;
    } else {
        HeadSpec hs;
        hs.n_out = desc->n_out;
        for (int o = 0; o < desc->n_out; ++o) { hs.act[o] = desc->act[o]; hs.grid_axis[o] = desc->grid_axis[o]; }
        ClipBatch hb;
        hb.nb = nb; hb.in_bs = WS; hb.out_bs = p.out_bs;
        rc = launch_heads(head_in, head_cin, T, p.h[3], p.w[3], lin_tail ? M4 : wts->head_w, wts->head_b, hs, wts->grid_t, wts->grid_y, wts->grid_x, out, sm, hb,
                          lin_tail ? up4 : nullptr, WS, lin_tail ? &hgn : nullptr);
        if (rc) return rc;
    }
    if (detached) SS_HIP(hipEventRecord(bs->done[3], sm));
    else if (bs) {
        // join every branch stream DIRECTLY into the caller's stream as well (stream capture only recognises direct joins
        // into the origin stream; the transitive 32x -> 16x -> 8x -> caller chain above is not enough for hipGraph capture)
        SS_HIP(hipStreamWaitEvent(s, bs->done[0], 0));
        SS_HIP(hipStreamWaitEvent(s, bs->done[1], 0));
    }
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_decoder_join(int32_t concurrency, void* stream) {
    SS_CHECK_ARG(concurrency >= 1, "decoder_join: concurrency set must be >= 1");
    BranchStreams* bs = get_streams(concurrency - 1);
    SS_CHECK_ARG(bs, "decoder_join: no stream set");
    for (int i = 0; i < 4; ++i) SS_HIP(hipStreamWaitEvent(as_stream(stream), bs->done[i], 0));
    return STEMSEG_OK;
}

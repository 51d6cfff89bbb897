// Foreground gather + sequential seed-and-threshold clustering + overlap statistics.
//
// Reference: /root/reference/stemseg/inference/clusterers.py:60-175 (SequentialClustering._process),
// inference/online_chainer.py:11-22,258-281 (fg gather), :291-343 (label association counts),
// inference/main.py:93-103 (fg mask from averaged seediness).
//
// The reference runs <= 20 Python rounds of ~25 small launches with 3-4 host syncs each.  Here a round is ONE
// launch with no host involvement: the kernel for round i first reduces the per-block (best seediness, index,
// #unassigned) partials left by the previous launch -- every block redundantly, so no inter-block hand-off,
// no spin-wait, no atomics -- decides termination exactly like the loop header of clusterers.py:106-118,
// then streams its slice of the points once (distance to the new centre, exp, threshold, label) while
// building the partials for round i+1.  All per-point state is a 4-byte round index; the 36 B/point of
// inputs stay L2 / Infinity-Cache resident across rounds.  The secondary pass recomputes the per-round
// distances instead of materialising the [N, K] matrix of clusterers.py:148-151.
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace stemseg {

constexpr int CL_THREADS = 256;
constexpr int CL_MAX_BLOCKS = 1024;

struct ClusterPartial {   // 16 bytes
    float best_seed;
    int   best_idx;       // -1: no unassigned point in the block
    int   n_unassigned;
    int   pad;
};

struct ClusterState {
    int done;
    int K;
    int exhausted;
    int pad;
    long long n_un_last;
};

struct ClusterKParams {
    const float* emb;
    const float* bw;
    const float* seed;
    const long long* n_dev;
    long long n_max;
    int E, Ev, n_free;
    float free_bw[STEMSEG_MAX_EMB_DIMS];
    float primary, secondary, min_seed;
    int max_instances;
    long long label_start;
    int* round_of;                // [n_max] -1 = unassigned, else the round that claimed the point
    ClusterPartial* partials;     // [2][CL_MAX_BLOCKS]
    ClusterState* state;
    StemsegClusterMeta* meta;
    long long* labels;
    unsigned char* masks;
    float* probs;
    int nblk;
};

// Several independent point sets (the clips of one step) through the same launches: blockIdx.y selects the set.  The parameter
// blocks travel in the kernel arguments and are read in place (a reference into the kernarg segment: scalar loads, no private copy).
constexpr int CL_MAX_BATCH = 8;
struct ClusterBatch {
    ClusterKParams p[CL_MAX_BATCH];
};
__device__ __forceinline__ long long cl_n(const ClusterKParams& p) {
    long long n = p.n_dev ? *p.n_dev : p.n_max;
    return n < p.n_max ? n : p.n_max;
}

// is (s1, i1) a better seed than (s0, i0)?  torch.argmax semantics: first maximal element.
__device__ __forceinline__ bool better(float s1, int i1, float s0, int i0) {
    if (i1 < 0) return false;
    if (i0 < 0) return true;
    return (s1 > s0) || (s1 == s0 && i1 < i0);
}

__device__ __forceinline__ void block_reduce_partial(float& bs, int& bi, int& cnt, float* sh_s, int* sh_i, int* sh_c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float s2 = __shfl_down(bs, o, 64);
        const int i2 = __shfl_down(bi, o, 64);
        const int c2 = __shfl_down(cnt, o, 64);
        if (better(s2, i2, bs, bi)) { bs = s2; bi = i2; }
        cnt += c2;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh_s[w] = bs; sh_i[w] = bi; sh_c[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < CL_THREADS / 64; ++k) {
            if (better(sh_s[k], sh_i[k], bs, bi)) { bs = sh_s[k]; bi = sh_i[k]; }
            cnt += sh_c[k];
        }
        sh_s[0] = bs; sh_i[0] = bi; sh_c[0] = cnt;
    }
    __syncthreads();
    bs = sh_s[0]; bi = sh_i[0]; cnt = sh_c[0];
    __syncthreads();
}

// clusterers.py:56-58: (pow(x - c, 2) * bw).sum(-1).sqrt() -- unfused fp32 ops; the summation order mirrors torch's
// CPU inner-dim reduce (sequential for E <= 4; 4-wide vector + tail for E == 5, i.e. ((t0+t4)+t1)+t2)+t3).
__device__ __forceinline__ float cl_distance(const float* __restrict__ x, const float* c, const float* b, int E) {
    float term[STEMSEG_MAX_EMB_DIMS];
#pragma unroll
    for (int e = 0; e < STEMSEG_MAX_EMB_DIMS; ++e) {
        if (e < E) {
            const float d = __fsub_rn(x[e], c[e]);
            term[e] = __fmul_rn(__fmul_rn(d, d), b[e]);
        } else term[e] = 0.f;
    }
    float acc;
    if (E == 5) {
        acc = __fadd_rn(term[0], term[4]);
        acc = __fadd_rn(acc, term[1]);
        acc = __fadd_rn(acc, term[2]);
        acc = __fadd_rn(acc, term[3]);
    } else {
        acc = term[0];
#pragma unroll
        for (int e = 1; e < STEMSEG_MAX_EMB_DIMS; ++e)
            if (e < E) acc = __fadd_rn(acc, term[e]);
    }
    return __fsqrt_rn(acc);
}

__device__ __forceinline__ float cl_prob(float d) { return expf(__fmul_rn(-0.5f, d)); }   // clusterers.py:52-54

// round == -1 : initialise (labels = -1) and build the partials for round 0.
// round >= 0  : loop body of clusterers.py:106-146 for instance `round`.
// Latency is what a round costs (20 of them run back to back on ~2e5 points), so (a) a thread owns ONE point wherever the
// grid allows (n_max <= 256 * CL_MAX_BLOCKS) and (b) its point's state, embedding and seediness are requested BEFORE the loop
// header's chain of dependent reads (termination flag -> partials -> the seed's centre), which they do not depend on.
__global__ __launch_bounds__(CL_THREADS) void cluster_round_kernel(const ClusterBatch batch, const int round) {
    const ClusterKParams& p = batch.p[blockIdx.y];
    if ((int)blockIdx.x >= p.nblk) return;                  // (sets of a batch differ in size: the grid is cut for the largest)
    __shared__ float sh_s[CL_THREADS / 64];
    __shared__ int sh_i[CL_THREADS / 64];
    __shared__ int sh_c[CL_THREADS / 64];
    __shared__ float sh_center[2 * STEMSEG_MAX_EMB_DIMS];
    __shared__ int sh_done;

    const long long N = cl_n(p);
    const long long chunk = (N + p.nblk - 1) / p.nblk;
    const long long beg = (long long)blockIdx.x * chunk;
    const long long end = (beg + chunk < N) ? beg + chunk : N;

    // ---- this thread's first point, requested up front ------------------------------------------------
    const long long i0 = beg + threadIdx.x;
    int ro0 = -1;
    float s0 = 0.f, x0[STEMSEG_MAX_EMB_DIMS];
#pragma unroll
    for (int e = 0; e < STEMSEG_MAX_EMB_DIMS; ++e) x0[e] = 0.f;
    if (i0 < end) {
        if (round >= 0) ro0 = p.round_of[i0];
        s0 = p.seed[i0];
        if (round >= 0) {
            if (p.E == 4) {                                            // (rows of 16 B: one load)
                const float4 v = *reinterpret_cast<const float4*>(p.emb + i0 * 4);
                x0[0] = v.x; x0[1] = v.y; x0[2] = v.z; x0[3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < STEMSEG_MAX_EMB_DIMS; ++e) if (e < p.E) x0[e] = p.emb[i0 * p.E + e];
            }
        }
    }

    float center[STEMSEG_MAX_EMB_DIMS], bwv[STEMSEG_MAX_EMB_DIMS];
    if (round < 0 && blockIdx.x == 0) {
        // zero the loop state and the output record here rather than with hipMemsetAsync: no block of THIS launch reads
        // them, and memset nodes in a captured hipGraph were seen to land out of order with the short kernels around them
        for (int k = threadIdx.x; k < (int)(sizeof(StemsegClusterMeta) / 4); k += CL_THREADS) reinterpret_cast<int*>(p.meta)[k] = 0;
        for (int k = threadIdx.x; k < (int)(sizeof(ClusterState) / 4); k += CL_THREADS) reinterpret_cast<int*>(p.state)[k] = 0;
    }
    if (round >= 0) {
        // ---- loop header: termination flag + the previous launch's partials, requested together (identically in every block) ----
        const ClusterPartial* prev = p.partials + (size_t)(round & 1) * CL_MAX_BLOCKS;
        if (threadIdx.x == 0) sh_done = p.state->done;             // (another block of this launch may set it meanwhile: read once)
        float bs = 0.f; int bi = -1; int cnt = 0;
        for (int k = threadIdx.x; k < p.nblk; k += CL_THREADS) {
            const ClusterPartial q = prev[k];
            if (better(q.best_seed, q.best_idx, bs, bi)) { bs = q.best_seed; bi = q.best_idx; }
            cnt += q.n_unassigned;
        }
        __syncthreads();
        if (sh_done) return;
        block_reduce_partial(bs, bi, cnt, sh_s, sh_i, sh_c);
        const bool stop = (cnt == 0) || (bs < p.min_seed);       // clusterers.py:109-110, 116-117
        if (stop) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                p.state->n_un_last = cnt;
                p.state->done = 1;
            }
            return;
        }
        if (threadIdx.x < p.E) {
            const int e = threadIdx.x;
            sh_center[e] = p.emb[(long long)bi * p.E + e];
            sh_center[STEMSEG_MAX_EMB_DIMS + e] = (e < p.Ev) ? p.bw[(long long)bi * p.Ev + e] : p.free_bw[e - p.Ev];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < STEMSEG_MAX_EMB_DIMS; ++e) {
            center[e] = (e < p.E) ? sh_center[e] : 0.f;
            bwv[e] = (e < p.E) ? sh_center[STEMSEG_MAX_EMB_DIMS + e] : 0.f;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            p.state->K = round + 1;
            p.state->n_un_last = cnt;
            if (round == p.max_instances - 1) p.state->exhausted = 1;
            for (int e = 0; e < p.E; ++e) {
                p.meta->centers[round][e] = center[e];
                p.meta->bandwidths[round][e] = bwv[e];
            }
            p.meta->seed_prob[round] = bs;
        }
    }

    // ---- stream this block's slice once ---------------------------------------------------------
    float bs = 0.f; int bi = -1; int cnt = 0;
    for (long long i = i0; i < end; i += CL_THREADS) {
        const bool first = (i == i0);
        bool avail;
        if (round < 0) { p.round_of[i] = -1; avail = true; }
        else avail = ((first ? ro0 : p.round_of[i]) < 0);
        bool still = avail;
        if (round >= 0) {
            float pr = 0.f;
            bool match = false;
            if (avail) {
                float xv[STEMSEG_MAX_EMB_DIMS];
#pragma unroll
                for (int e = 0; e < STEMSEG_MAX_EMB_DIMS; ++e) xv[e] = first ? x0[e] : ((e < p.E) ? p.emb[i * p.E + e] : 0.f);
                pr = cl_prob(cl_distance(xv, center, bwv, p.E));
                match = pr > p.primary;                               // clusterers.py:140
                if (match) { p.round_of[i] = round; still = false; }
            }
            if (p.masks) p.masks[(long long)round * p.n_max + i] = match ? 1 : 0;
            if (p.probs) p.probs[(long long)round * p.n_max + i] = pr;
        }
        if (still) {
            const float s = first ? s0 : p.seed[i];
            if (better(s, (int)i, bs, bi)) { bs = s; bi = (int)i; }
            ++cnt;
        }
    }
    block_reduce_partial(bs, bi, cnt, sh_s, sh_i, sh_c);
    if (threadIdx.x == 0) {
        ClusterPartial q;
        q.best_seed = bs; q.best_idx = bi; q.n_unassigned = cnt; q.pad = 0;
        p.partials[(size_t)((round + 1) & 1) * CL_MAX_BLOCKS + blockIdx.x] = q;
    }
}

// secondary assignment (clusterers.py:148-159) + int64 labels + meta record
__global__ __launch_bounds__(CL_THREADS) void cluster_final_kernel(const ClusterBatch batch) {
    const ClusterKParams& p = batch.p[blockIdx.y];
    __shared__ float sh_c[STEMSEG_MAX_INSTANCES][STEMSEG_MAX_EMB_DIMS];
    __shared__ float sh_b[STEMSEG_MAX_INSTANCES][STEMSEG_MAX_EMB_DIMS];
    const long long N = cl_n(p);
    const int K = p.state->K;
    const int exhausted = p.state->exhausted;
    const long long n_un_last = p.state->n_un_last;
    for (int k = threadIdx.x; k < K * STEMSEG_MAX_EMB_DIMS; k += CL_THREADS) {
        sh_c[k / STEMSEG_MAX_EMB_DIMS][k % STEMSEG_MAX_EMB_DIMS] = p.meta->centers[k / STEMSEG_MAX_EMB_DIMS][k % STEMSEG_MAX_EMB_DIMS];
        sh_b[k / STEMSEG_MAX_EMB_DIMS][k % STEMSEG_MAX_EMB_DIMS] = p.meta->bandwidths[k / STEMSEG_MAX_EMB_DIMS][k % STEMSEG_MAX_EMB_DIMS];
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.meta->K = K;
        p.meta->exhausted = exhausted;
        p.meta->n_points = N;
        p.meta->n_unassigned_last = n_un_last;
    }
    const bool secondary = (n_un_last > 0) && (K > 0);
    for (long long i = (long long)blockIdx.x * CL_THREADS + threadIdx.x; i < p.n_max; i += (long long)gridDim.x * CL_THREADS) {
        if (i >= N) { p.labels[i] = -1; continue; }
        int r = p.round_of[i];
        // `available_embeddings_mask` as left by the last evaluated loop header: fresh unless the loop ran out of
        // rounds, in which case points claimed in the final round are still marked available (stale mask).
        const bool avail = (r < 0) || (exhausted && r == p.max_instances - 1);
        if (secondary && avail) {
            float xv[STEMSEG_MAX_EMB_DIMS];
#pragma unroll
            for (int e = 0; e < STEMSEG_MAX_EMB_DIMS; ++e) xv[e] = (e < p.E) ? p.emb[i * p.E + e] : 0.f;
            float m = -1.f; int a = 0;
            for (int k = 0; k < K; ++k) {                  // the point was available in every round 0..K-1
                const float d = cl_distance(xv, sh_c[k], sh_b[k], p.E);
                if (d > m) { m = d; a = k; }               // QUIRK: max distance, first index on ties
            }
            if (cl_prob(m) > p.secondary) r = a;
        }
        p.labels[i] = (r >= 0) ? (long long)r + p.label_start : -1;
    }
}

// ---- fg mask ---------------------------------------------------------------------------------------
__global__ void seed_accumulate_kernel(float* acc, const float* plane, long long n, int first) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        acc[i] = __fadd_rn(first ? 0.f : acc[i], plane[i]);
}
__global__ void fg_mask_kernel(const float* acc, float count, float thr, unsigned char* mask, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        mask[i] = (__fdiv_rn(acc[i], count) > thr) ? 1 : 0;
}

// all frames of a sequence at once: mask[f][p] = acc[f][p] / counts[f] > thr (counts[f] = clips that contain frame f; 0 -> background)
__global__ void fg_mask_frames_kernel(const float* acc, const float* counts, float thr, unsigned char* mask, int F, long long HW) {
    const long long n = (long long)F * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float c = counts[i / HW];
        mask[i] = (c > 0.f && __fdiv_rn(acc[i], c) > thr) ? 1 : 0;
    }
}

// ---- fg gather: count -> scan -> scatter ------------------------------------------------------------
constexpr int GA_BLOCK = 1024;   // voxels per block (256 threads x 4)

__global__ __launch_bounds__(256) void gather_count_kernel(const unsigned char* __restrict__ fg, long long V, int* block_counts) {
    const long long base = (long long)blockIdx.x * GA_BLOCK;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long v = base + k * 256 + threadIdx.x;
        if (v < V && fg[v]) ++c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    __shared__ int sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// single block: exclusive scan of block_counts (nb entries) -> block_offsets (long long), total at [nb]
__global__ __launch_bounds__(1024) void gather_scan_kernel(const int* block_counts, long long* block_offsets, int nb) {
    __shared__ long long sh[1024];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int k = base + threadIdx.x;
        const long long v = (k < nb) ? block_counts[k] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            long long t = (threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (k < nb) block_offsets[k] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_offsets[nb] = carry;
}

struct GatherParams {
    const float* emb; const float* bw; const float* seed; const unsigned char* fg;
    int E, Ev, T;
    long long HW, V;
    float* emb_out; float* bw_out; float* seed_out; int* vox; long long* frame_offsets;
    const long long* block_offsets;
};

__global__ __launch_bounds__(256) void gather_scatter_kernel(const GatherParams p) {
    __shared__ int wave_tot[4][4];
    const long long base = (long long)blockIdx.x * GA_BLOCK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long blk_off = p.block_offsets[blockIdx.x];
    bool f[4];
    int rank_in_wave[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long v = base + k * 256 + threadIdx.x;
        f[k] = (v < p.V) && p.fg[v];
        const unsigned long long b = __ballot(f[k]);
        rank_in_wave[k] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[k][w] = __popcll(b);
    }
    __syncthreads();
    // voxel order inside the block: k-major (k*256 + tid), so ranks accumulate over k, then wave, then lane
    int run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int before = run;
        for (int ww = 0; ww < 4; ++ww) {
            if (ww < w) before += wave_tot[k][ww];
            run += wave_tot[k][ww];
        }
        const long long v = base + k * 256 + threadIdx.x;
        const long long dst = blk_off + before + rank_in_wave[k];
        if (v < p.V && (v % p.HW) == 0) p.frame_offsets[v / p.HW] = dst;      // exclusive prefix at the frame start
        if (f[k]) {
            for (int e = 0; e < p.E; ++e) p.emb_out[dst * p.E + e] = p.emb[(long long)e * p.V + v];
            for (int e = 0; e < p.Ev; ++e) p.bw_out[dst * p.Ev + e] = p.bw[(long long)e * p.V + v];
            p.seed_out[dst] = p.seed[v];
            p.vox[dst] = (int)v;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) p.frame_offsets[p.T] = p.block_offsets[gridDim.x];
}

// ---- overlap statistics / relabel ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void overlap_counts_kernel(const long long* la, const long long* lb, long long n,
                                                             const int* lut_a, int len_a, const int* lut_b, int len_b,
                                                             int Ka, int Kb, unsigned long long* inter,
                                                             unsigned long long* cnt_a, unsigned long long* cnt_b) {
    extern __shared__ unsigned int hist[];   // [Ka*Kb + Ka + Kb]
    const int nh = Ka * Kb + Ka + Kb;
    for (int k = threadIdx.x; k < nh; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long a = la[i] + 1, b = lb[i] + 1;
        const int ia = (a >= 0 && a < len_a) ? lut_a[a] : -1;
        const int ib = (b >= 0 && b < len_b) ? lut_b[b] : -1;
        if (ia >= 0) atomicAdd(&hist[Ka * Kb + ia], 1u);
        if (ib >= 0) atomicAdd(&hist[Ka * Kb + Ka + ib], 1u);
        if (ia >= 0 && ib >= 0) atomicAdd(&hist[ia * Kb + ib], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nh; k += blockDim.x) {
        const unsigned int v = hist[k];
        if (!v) continue;
        if (k < Ka * Kb) atomicAdd(&inter[k], (unsigned long long)v);
        else if (k < Ka * Kb + Ka) atomicAdd(&cnt_a[k - Ka * Kb], (unsigned long long)v);
        else atomicAdd(&cnt_b[k - Ka * Kb - Ka], (unsigned long long)v);
    }
}

// Large label sets (Ka*Kb+Ka+Kb beyond the LDS budget): same statistics with one wave-uniform check and global atomics.
__global__ __launch_bounds__(256) void overlap_counts_global_kernel(const long long* la, const long long* lb, long long n,
                                                                    const int* lut_a, int len_a, const int* lut_b, int len_b,
                                                                    int Kb, unsigned long long* inter,
                                                                    unsigned long long* cnt_a, unsigned long long* cnt_b) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long a = la[i] + 1, b = lb[i] + 1;
        const int ia = (a >= 0 && a < len_a) ? lut_a[a] : -1;
        const int ib = (b >= 0 && b < len_b) ? lut_b[b] : -1;
        if (ia >= 0) atomicAdd(&cnt_a[ia], 1ull);
        if (ib >= 0) atomicAdd(&cnt_b[ib], 1ull);
        if (ia >= 0 && ib >= 0) atomicAdd(&inter[(long long)ia * Kb + ib], 1ull);
    }
}

// Which ids occur (online_chainer.py:304-308's torch.unique, minus the sort: the flags are indexed by id) and the
// highest id + 1 (TrackContainer.add_labels :43-49).  present[id] = 1 for 0 <= id < cap; ids >= cap only raise max.
__global__ __launch_bounds__(256) void label_presence_kernel(const long long* labels, long long n, unsigned char* present,
                                                             int cap, unsigned long long* max_plus_1) {
    unsigned long long m = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long l = labels[i];
        if (l >= 0) {
            if (l < cap && !present[l]) present[l] = 1;     // benign race: every writer stores 1
            m = max(m, (unsigned long long)l + 1ull);
        } else if (!present[cap]) present[cap] = 1;         // a negative (outlier) label occurs: byte `cap` of the table
    }
    for (int o = 32; o; o >>= 1) m = max(m, (unsigned long long)__shfl_xor((long long)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(max_plus_1, m);
}

__global__ void relabel_kernel(long long* labels, long long n, const long long* map, int map_len) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long k = labels[i] + 1;
        if (k >= 0 && k < map_len) labels[i] = map[k];
    }
}


// ---- clip-parallel stitching (pipeline.run_sequence_sharded) ---------------------------------------
// Every rank clusters ITS clips with label_start = 1 (clusterers.py:121: labels are i + label_start, so the global id is an
// offset applied later), leaves the result as one byte per voxel -- 0 = background, 1..K = local instance, 255 = outlier --
// and only those planes travel.  The chain of online_chainer.py:193-236 then needs, per clip and frame, the K1 x K2 table of
// (label of the clip that first contributed the frame, label of this clip) pairs: one launch for the whole sequence.

// compaction of a foreground mask alone: voxel_index + frame_offsets of stemseg_hip_fg_gather without the head outputs
__global__ __launch_bounds__(256) void compact_scatter_kernel(const unsigned char* __restrict__ fg, long long V, long long HW, int T,
                                                              const long long* __restrict__ block_offsets, int* vox, long long* frame_offsets) {
    __shared__ int wave_tot[4][4];
    const long long base = (long long)blockIdx.x * GA_BLOCK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long blk_off = block_offsets[blockIdx.x];
    bool f[4];
    int rank_in_wave[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long v = base + k * 256 + threadIdx.x;
        f[k] = (v < V) && fg[v];
        const unsigned long long b = __ballot(f[k]);
        rank_in_wave[k] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[k][w] = __popcll(b);
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int before = run;
        for (int ww = 0; ww < 4; ++ww) {
            if (ww < w) before += wave_tot[k][ww];
            run += wave_tot[k][ww];
        }
        const long long v = base + k * 256 + threadIdx.x;
        const long long dst = blk_off + before + rank_in_wave[k];
        if (v < V && (v % HW) == 0) frame_offsets[v / HW] = dst;
        if (f[k]) vox[dst] = (int)v;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) frame_offsets[T] = block_offsets[gridDim.x];
}

__global__ void labels_to_codes_kernel(const long long* __restrict__ labels, const int* __restrict__ vox, const long long* n_dev,
                                       long long n_max, long long label_start, unsigned char* codes) {
    long long N = n_dev ? *n_dev : n_max;
    N = N < n_max ? N : n_max;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const long long l = labels[i];
        const long long c = l - label_start + 1;
        codes[vox[i]] = (l < 0) ? 255 : (unsigned char)(c < 1 ? 254 : (c > 254 ? 254 : c));      // (254: out of range, never produced)
    }
}

__device__ __forceinline__ int code_bin(unsigned int c, int B) { return c == 255u ? B - 1 : (c < (unsigned)(B - 1) ? (int)c : B - 1); }

// tables[item][a][b] = #voxels v with bin(codes[plane_a[item]][v]) == a and bin(codes[plane_b[item]][v]) == b, over the voxels where
// plane_b is foreground (code != 0); plane_a == -1: a = 0 for every voxel.  grid = (chunks, items); LDS histogram, integer atomics.
__global__ __launch_bounds__(256) void pair_tables_kernel(const unsigned char* __restrict__ codes, const int* __restrict__ plane_a,
                                                          const int* __restrict__ plane_b, long long HW, int B, unsigned int* tables) {
    extern __shared__ unsigned int hist[];
    const int item = blockIdx.y;
    const int pa = plane_a[item], pb = plane_b[item];
    for (int k = threadIdx.x; k < B * B; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    const unsigned char* cb = codes + (long long)pb * HW;
    const unsigned char* ca = pa >= 0 ? codes + (long long)pa * HW : nullptr;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < HW; v += (long long)gridDim.x * blockDim.x) {
        const unsigned int b = cb[v];
        if (!b) continue;
        const unsigned int a = ca ? ca[v] : 0u;
        atomicAdd(&hist[code_bin(a, B) * B + code_bin(b, B)], 1u);
    }
    __syncthreads();
    unsigned int* out = tables + (long long)item * B * B;
    for (int k = threadIdx.x; k < B * B; k += blockDim.x)
        if (hist[k]) atomicAdd(&out[k], hist[k]);
}

// out[item.out_begin + q] = lut[item][bin(codes[item.plane][vox[item.src_begin + q] - item.vbase])], q < item.count
// items: int64 [n_items][5] = (src_begin, count, vbase, plane, out_begin).  grid = (chunks, items)
__global__ __launch_bounds__(256) void codes_to_labels_kernel(const unsigned char* __restrict__ codes, const int* __restrict__ vox,
                                                              const long long* __restrict__ items, const long long* __restrict__ lut,
                                                              long long HW, int B, long long* out) {
    const long long* it = items + (long long)blockIdx.y * 5;
    const long long src = it[0], cnt = it[1], vbase = it[2], plane = it[3], dst = it[4];
    const long long* l = lut + (long long)blockIdx.y * B;
    const unsigned char* c = codes + plane * HW;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < cnt; q += (long long)gridDim.x * blockDim.x)
        out[dst + q] = l[code_bin(c[vox[src + q] - vbase], B)];
}

static int grid_for(long long n, int threads, int cap) { return (int)std::max<long long>(1, std::min<long long>(ceil_div(n, threads), cap)); }

}  // namespace stemseg

using namespace stemseg;

extern "C" int stemseg_hip_seediness_accumulate(float* acc, const float* plane, int64_t n, int32_t first, void* stream) {
    SS_CHECK_ARG(acc && plane && n >= 0, "seediness_accumulate: bad arguments");
    if (n == 0) return STEMSEG_OK;
    hipLaunchKernelGGL(seed_accumulate_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, as_stream(stream), acc, plane, (long long)n, first);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_fg_mask(const float* acc, float count, float thr, uint8_t* mask, int64_t n, void* stream) {
    SS_CHECK_ARG(acc && mask && n >= 0 && count > 0.f, "fg_mask: bad arguments");
    if (n == 0) return STEMSEG_OK;
    hipLaunchKernelGGL(fg_mask_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, as_stream(stream), acc, count, thr, mask, (long long)n);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_fg_mask_frames(const float* acc, const float* counts, float thr, uint8_t* mask, int32_t F, int64_t HW, void* stream) {
    SS_CHECK_ARG(acc && counts && mask && F >= 0 && HW >= 0, "fg_mask_frames: bad arguments");
    if (F == 0 || HW == 0) return STEMSEG_OK;
    hipLaunchKernelGGL(fg_mask_frames_kernel, dim3(grid_for((int64_t)F * HW, 256, 4096)), dim3(256), 0, as_stream(stream), acc, counts, thr, mask, F,
                       (long long)HW);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_fg_gather(const float* emb, const float* bw, const float* seed, const uint8_t* fg, int32_t E, int32_t Ev,
                                     int32_t T, int64_t HW, float* emb_out, float* bw_out, float* seed_out, int32_t* voxel_index,
                                     int64_t* frame_offsets, void* scratch, void* stream) {
    SS_CHECK_ARG(emb && bw && seed && fg && emb_out && bw_out && seed_out && voxel_index && frame_offsets && scratch, "fg_gather: null pointer");
    SS_CHECK_ARG(E >= 1 && E <= STEMSEG_MAX_EMB_DIMS && Ev >= 0 && Ev <= E && T >= 1 && HW >= 1, "fg_gather: bad dims");
    const long long V = (long long)T * HW;
    SS_CHECK_ARG(V < (1ll << 31), "fg_gather: more than 2^31 voxels");
    const int nb = (int)ceil_div(V, GA_BLOCK);
    // scratch: [nb] int counts | pad to 8 | [nb+1] long long offsets
    int* counts = reinterpret_cast<int*>(scratch);
    long long* offsets = reinterpret_cast<long long*>(reinterpret_cast<char*>(scratch) + round_up((int64_t)nb * 4, 8));
    hipStream_t s = as_stream(stream);
    void* ev = profile_begin(45, (double)V * (1.0 + 4.0 * (E + Ev + 1)), s);     // mask + head channels read once (writes <= that)
    hipLaunchKernelGGL(gather_count_kernel, dim3(nb), dim3(256), 0, s, fg, V, counts);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(gather_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)counts, offsets, nb);
    SS_LAUNCH_CHECK();
    GatherParams p;
    p.emb = emb; p.bw = bw; p.seed = seed; p.fg = fg; p.E = E; p.Ev = Ev; p.T = T; p.HW = HW; p.V = V;
    p.emb_out = emb_out; p.bw_out = bw_out; p.seed_out = seed_out; p.vox = voxel_index;
    p.frame_offsets = reinterpret_cast<long long*>(frame_offsets); p.block_offsets = offsets;
    hipLaunchKernelGGL(gather_scatter_kernel, dim3(nb), dim3(256), 0, s, p);
    profile_end(ev, s);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" size_t stemseg_hip_cluster_workspace_bytes(int64_t n_max) {
    if (n_max < 0) return 0;
    return (size_t)round_up((int64_t)n_max * 4, 256) + 2 * CL_MAX_BLOCKS * sizeof(ClusterPartial) + 256;
}

static int cl_fill(ClusterKParams& p, const StemsegClusterItem& it, int32_t E, int32_t Ev, const StemsegClusterParams* params, int points_per_thread) {
    SS_CHECK_ARG(it.n_max > 0 && it.n_max < (1ll << 31), "cluster: n_max out of range");
    SS_CHECK_ARG(it.meta_dev && it.emb && (it.bw || Ev == 0) && it.seed && it.labels && it.workspace, "cluster: null pointer");
    SS_CHECK_ARG(it.ws_bytes >= stemseg_hip_cluster_workspace_bytes(it.n_max), "cluster: workspace too small");
    p.emb = it.emb; p.bw = it.bw; p.seed = it.seed; p.n_dev = reinterpret_cast<const long long*>(it.n_points_dev); p.n_max = it.n_max;
    p.E = E; p.Ev = Ev; p.n_free = params->n_free_dims;
    for (int e = 0; e < STEMSEG_MAX_EMB_DIMS; ++e) p.free_bw[e] = params->free_dim_bandwidths[e];
    p.primary = params->primary_prob_thresh; p.secondary = params->secondary_prob_thresh; p.min_seed = params->min_seediness_prob;
    p.max_instances = params->max_instances; p.label_start = it.label_start;
    char* w = reinterpret_cast<char*>(it.workspace);
    p.round_of = reinterpret_cast<int*>(w);
    w += round_up((int64_t)it.n_max * 4, 256);
    p.partials = reinterpret_cast<ClusterPartial*>(w);
    w += 2 * CL_MAX_BLOCKS * sizeof(ClusterPartial);
    p.state = reinterpret_cast<ClusterState*>(w);
    p.meta = it.meta_dev; p.labels = reinterpret_cast<long long*>(it.labels); p.masks = it.opt_masks; p.probs = it.opt_probs;
    // one point per thread up to 262 144 points for a lone set (a round's cost is latency); two when several sets share the
    // launches, so that all their workgroups are resident together (8 x 256 threads per CU).  The partials' reduction is a
    // max with a lowest-index tie-break and integer counts: the result does not depend on how the points are cut into blocks.
    p.nblk = grid_for(it.n_max, CL_THREADS * points_per_thread, CL_MAX_BLOCKS);
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_cluster_batch(const StemsegClusterItem* items, int32_t n_items, int32_t E, int32_t Ev, const StemsegClusterParams* params,
                                         void* stream) {
    SS_CHECK_ARG(params && (items || n_items == 0) && n_items >= 0, "cluster_batch: null params / items");
    SS_CHECK_ARG(E >= 1 && E <= STEMSEG_MAX_EMB_DIMS && Ev >= 0 && Ev + params->n_free_dims == E,
                 "cluster: E=%d must equal Ev=%d + n_free_dims=%d (<= %d)", E, Ev, params->n_free_dims, STEMSEG_MAX_EMB_DIMS);
    SS_CHECK_ARG(params->max_instances >= 1 && params->max_instances <= STEMSEG_MAX_INSTANCES, "cluster: max_instances out of range");
    static_assert(sizeof(StemsegClusterMeta) % 4 == 0 && sizeof(ClusterState) % 4 == 0, "word-zeroed in the round -1 launch");
    hipStream_t s = as_stream(stream);
    for (int i0 = 0; i0 < n_items; i0 += CL_MAX_BATCH) {
        ClusterBatch b;
        int n_live = 0;
        for (int i = i0; i < n_items && i < i0 + CL_MAX_BATCH; ++i) n_live += items[i].n_max > 0 ? 1 : 0;
        const int ppt = n_live >= 3 ? 2 : 1;
        int n = 0, nblk = 1;
        long long n_largest = 1;
        double work = 0.0;
        for (int i = i0; i < n_items && i < i0 + CL_MAX_BATCH; ++i) {
            SS_CHECK_ARG(items[i].n_max >= 0 && items[i].meta_dev, "cluster: bad item %d", i);
            if (items[i].n_max == 0) {       // clusterers.py:62-69: nothing to cluster, empty record
                SS_HIP(hipMemsetAsync(items[i].meta_dev, 0, sizeof(StemsegClusterMeta), s));
                continue;
            }
            const int rc = cl_fill(b.p[n], items[i], E, Ev, params, ppt);
            if (rc) return rc;
            nblk = std::max(nblk, b.p[n].nblk);
            n_largest = std::max<long long>(n_largest, items[i].n_max);
            // compulsory bytes: every point's inputs read once, its int64 label written once (36 B / point for E+Ev = 6)
            work += (double)items[i].n_max * (4.0 * (E + Ev + 1) + 8.0);
            ++n;
        }
        if (n == 0) continue;
        for (int k = n; k < CL_MAX_BATCH; ++k) b.p[k] = b.p[0];      // (never selected: blockIdx.y < n)
        void* ev = profile_begin(46, work, s);
        for (int round = -1; round < params->max_instances; ++round) {
            hipLaunchKernelGGL(cluster_round_kernel, dim3(nblk, n), dim3(CL_THREADS), 0, s, b, round);
            SS_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(cluster_final_kernel, dim3(grid_for(n_largest, CL_THREADS, 2048), n), dim3(CL_THREADS), 0, s, b);
        profile_end(ev, s);
        SS_LAUNCH_CHECK();
    }
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_cluster(const float* emb, const float* bw, const float* seed, int64_t n_max, const int64_t* n_points_dev,
                                   int32_t E, int32_t Ev, const StemsegClusterParams* params, int64_t label_start, int64_t* labels,
                                   StemsegClusterMeta* meta_dev, uint8_t* opt_masks, float* opt_probs, void* workspace, size_t ws_bytes,
                                   void* stream) {
    SS_CHECK_ARG(params && meta_dev, "cluster: null params/meta");
    SS_CHECK_ARG(n_max >= 0 && n_max < (1ll << 31), "cluster: n_max out of range");
    StemsegClusterItem it;
    it.emb = emb; it.bw = bw; it.seed = seed; it.n_max = n_max; it.n_points_dev = n_points_dev; it.label_start = label_start; it.labels = labels;
    it.meta_dev = meta_dev; it.opt_masks = opt_masks; it.opt_probs = opt_probs; it.workspace = workspace; it.ws_bytes = ws_bytes;
    return stemseg_hip_cluster_batch(&it, 1, E, Ev, params, stream);
}

extern "C" int stemseg_hip_overlap_counts(const int64_t* labels_a, const int64_t* labels_b, int64_t n, const int32_t* lut_a,
                                          int32_t lut_a_len, const int32_t* lut_b, int32_t lut_b_len, int32_t Ka, int32_t Kb,
                                          int64_t* inter, int64_t* cnt_a, int64_t* cnt_b, void* stream) {
    SS_CHECK_ARG(Ka >= 0 && Kb >= 0 && n >= 0 && lut_a_len >= 0 && lut_b_len >= 0, "overlap_counts: negative size");
    SS_CHECK_ARG((int64_t)Ka * Kb < (1ll << 31), "overlap_counts: Ka * Kb = %lld does not fit the index type", (long long)Ka * Kb);
    hipStream_t s = as_stream(stream);
    if (Ka * Kb) SS_HIP(hipMemsetAsync(inter, 0, sizeof(int64_t) * Ka * Kb, s));
    if (Ka) SS_HIP(hipMemsetAsync(cnt_a, 0, sizeof(int64_t) * Ka, s));
    if (Kb) SS_HIP(hipMemsetAsync(cnt_b, 0, sizeof(int64_t) * Kb, s));
    if (n == 0 || (Ka == 0 && Kb == 0)) return STEMSEG_OK;
    SS_CHECK_ARG(labels_a && labels_b && lut_a && lut_b && inter && cnt_a && cnt_b, "overlap_counts: null pointer");
    const long long nh = (long long)Ka * Kb + Ka + Kb;
    if (nh <= 12288) {              // the usual case (<= max_instances ids per side): per-workgroup LDS histogram
        hipLaunchKernelGGL(overlap_counts_kernel, dim3(grid_for(n, 256 * 8, 512)), dim3(256), sizeof(unsigned int) * nh, s,
                           reinterpret_cast<const long long*>(labels_a), reinterpret_cast<const long long*>(labels_b), (long long)n,
                           lut_a, lut_a_len, lut_b, lut_b_len, Ka, Kb, reinterpret_cast<unsigned long long*>(inter),
                           reinterpret_cast<unsigned long long*>(cnt_a), reinterpret_cast<unsigned long long*>(cnt_b));
    } else {                        // any number of labels: global atomics (contention is spread over the large table)
        hipLaunchKernelGGL(overlap_counts_global_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s,
                           reinterpret_cast<const long long*>(labels_a), reinterpret_cast<const long long*>(labels_b), (long long)n,
                           lut_a, lut_a_len, lut_b, lut_b_len, Kb, reinterpret_cast<unsigned long long*>(inter),
                           reinterpret_cast<unsigned long long*>(cnt_a), reinterpret_cast<unsigned long long*>(cnt_b));
    }
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_label_presence(const int64_t* labels, int64_t n, uint8_t* present, int32_t cap, int64_t* max_plus_1,
                                          int32_t accumulate, void* stream) {
    SS_CHECK_ARG(n >= 0 && cap >= 0 && max_plus_1 && present, "label_presence: bad arguments (present: cap + 1 bytes)");
    hipStream_t s = as_stream(stream);
    if (!accumulate) {
        SS_HIP(hipMemsetAsync(present, 0, (size_t)cap + 1, s));
        SS_HIP(hipMemsetAsync(max_plus_1, 0, sizeof(int64_t), s));
    }
    if (n == 0) return STEMSEG_OK;
    SS_CHECK_ARG(labels, "label_presence: null labels");
    hipLaunchKernelGGL(label_presence_kernel, dim3(grid_for(n, 256 * 4, 1024)), dim3(256), 0, s,
                       reinterpret_cast<const long long*>(labels), (long long)n, present, cap,
                       reinterpret_cast<unsigned long long*>(max_plus_1));
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_relabel(int64_t* labels, int64_t n, const int64_t* map, int32_t map_len, void* stream) {
    SS_CHECK_ARG(n >= 0 && map_len >= 0, "relabel: bad arguments");
    if (n == 0 || map_len == 0) return STEMSEG_OK;
    SS_CHECK_ARG(labels && map, "relabel: null pointer");
    hipLaunchKernelGGL(relabel_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<long long*>(labels), (long long)n, reinterpret_cast<const long long*>(map), map_len);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_fg_compact(const uint8_t* fg, int32_t T, int64_t HW, int32_t* voxel_index, int64_t* frame_offsets, void* scratch,
                                      void* stream) {
    SS_CHECK_ARG(fg && voxel_index && frame_offsets && scratch && T >= 1 && HW >= 1, "fg_compact: bad arguments");
    const long long V = (long long)T * HW;
    SS_CHECK_ARG(V < (1ll << 31), "fg_compact: more than 2^31 voxels");
    const int nb = (int)ceil_div(V, GA_BLOCK);
    int* counts = reinterpret_cast<int*>(scratch);
    long long* offsets = reinterpret_cast<long long*>(reinterpret_cast<char*>(scratch) + round_up((int64_t)nb * 4, 8));
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(gather_count_kernel, dim3(nb), dim3(256), 0, s, fg, V, counts);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(gather_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)counts, offsets, nb);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(nb), dim3(256), 0, s, fg, V, (long long)HW, T, (const long long*)offsets, voxel_index,
                       reinterpret_cast<long long*>(frame_offsets));
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_labels_to_codes(const int64_t* labels, const int32_t* voxel_index, const int64_t* n_points_dev, int64_t n_max,
                                           int64_t label_start, uint8_t* codes, int64_t V, void* stream) {
    SS_CHECK_ARG(codes && V >= 0 && n_max >= 0 && n_max <= V, "labels_to_codes: bad arguments");
    hipStream_t s = as_stream(stream);
    if (V) SS_HIP(hipMemsetAsync(codes, 0, (size_t)V, s));
    if (n_max == 0) return STEMSEG_OK;
    SS_CHECK_ARG(labels && voxel_index, "labels_to_codes: null pointer");
    hipLaunchKernelGGL(labels_to_codes_kernel, dim3(grid_for(n_max, 256, 2048)), dim3(256), 0, s, reinterpret_cast<const long long*>(labels),
                       voxel_index, reinterpret_cast<const long long*>(n_points_dev), (long long)n_max, (long long)label_start, codes);
    SS_LAUNCH_CHECK();
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_pair_tables(const uint8_t* codes, const int32_t* plane_a, const int32_t* plane_b, int32_t n_items, int64_t HW,
                                       int32_t B, int32_t* tables, void* stream) {
    SS_CHECK_ARG(n_items >= 0 && HW >= 0 && B >= 3 && B <= STEMSEG_MAX_INSTANCES + 2, "pair_tables: bad arguments (B = max_instances + 2)");
    if (n_items == 0) return STEMSEG_OK;
    SS_CHECK_ARG(codes && plane_a && plane_b && tables, "pair_tables: null pointer");
    hipStream_t s = as_stream(stream);
    SS_HIP(hipMemsetAsync(tables, 0, sizeof(int32_t) * (size_t)n_items * B * B, s));
    if (HW == 0) return STEMSEG_OK;
    for (int32_t i0 = 0; i0 < n_items; i0 += 65535) {                 // (gridDim.y <= 65535: long sequences take several launches)
        const int32_t ni = std::min<int32_t>(65535, n_items - i0);
        hipLaunchKernelGGL(pair_tables_kernel, dim3(grid_for(HW, 256 * 8, 64), (unsigned)ni), dim3(256), sizeof(unsigned int) * B * B, s, codes,
                           plane_a + i0, plane_b + i0, (long long)HW, B, reinterpret_cast<unsigned int*>(tables) + (size_t)i0 * B * B);
        SS_LAUNCH_CHECK();
    }
    return STEMSEG_OK;
}

extern "C" int stemseg_hip_codes_to_labels(const uint8_t* codes, const int32_t* voxel_index, const int64_t* items, int32_t n_items,
                                           int64_t max_count, const int64_t* lut, int64_t HW, int32_t B, int64_t* out, void* stream) {
    SS_CHECK_ARG(n_items >= 0 && max_count >= 0 && HW >= 0 && B >= 3 && B <= STEMSEG_MAX_INSTANCES + 2, "codes_to_labels: bad arguments");
    if (n_items == 0 || max_count == 0) return STEMSEG_OK;
    SS_CHECK_ARG(codes && voxel_index && items && lut && out, "codes_to_labels: null pointer");
    for (int32_t i0 = 0; i0 < n_items; i0 += 65535) {
        const int32_t ni = std::min<int32_t>(65535, n_items - i0);
        hipLaunchKernelGGL(codes_to_labels_kernel, dim3(grid_for(max_count, 256 * 4, 256), (unsigned)ni), dim3(256), 0, as_stream(stream), codes,
                           voxel_index, reinterpret_cast<const long long*>(items) + (size_t)i0 * 5, reinterpret_cast<const long long*>(lut) + (size_t)i0 * B,
                           (long long)HW, B, reinterpret_cast<long long*>(out));
        SS_LAUNCH_CHECK();
    }
    return STEMSEG_OK;
}

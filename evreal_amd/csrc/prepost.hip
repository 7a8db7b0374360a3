// Pre/post-processing around the network on gfx950.
//
//   evr_event_tensor_normalize   eval.py:398-410  (normalize_event_tensor), per window
//   evr_percentile_normalize     eval.py:380-395  (post_process_normalization) +
//                                utils/eval_utils.py:15-35 (np.percentile, 'linear' method)
//
// Both are HBM/L2-bound elementwise passes around small reductions; fp32 with one rounding per
// operation (built with -ffp-contract=off) so the elementwise arithmetic matches numpy/torch.
#include "common.h"
#include <cstdlib>

namespace {

// ---------------------------------------------------------------- event-tensor normalization
constexpr int NRM_BLOCKS = 32;   // partial reductions per window

__global__ __launch_bounds__(256) void nrm_partial_kernel(const float* __restrict__ vox, double* __restrict__ partials,
                                                           int64_t vol) {
    __shared__ double sh[3][4];
    const int w = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* v = vox + (int64_t)w * vol;
    const int64_t per = (vol + NRM_BLOCKS - 1) / NRM_BLOCKS;
    const int64_t b0 = per * blk, b1 = min(vol, b0 + per);
    double s1 = 0, s2 = 0, nz = 0;
    for (int64_t i = b0 + tid; i < b1; i += 256) {
        const float a = v[i];
        s1 += (double)a; s2 += (double)a * (double)a; nz += (a != 0.f) ? 1.0 : 0.0;
    }
    s1 = evr_wave_sum(s1); s2 = evr_wave_sum(s2); nz = evr_wave_sum(nz);
    if (lane == 0) { sh[0][wave] = s1; sh[1][wave] = s2; sh[2][wave] = nz; }
    __syncthreads();
    if (tid < 3) partials[((int64_t)w * NRM_BLOCKS + blk) * 3 + tid] = ((sh[tid][0] + sh[tid][1]) + sh[tid][2]) + sh[tid][3];
}

// stats layout: [n][n_part][3] doubles {sum, sumsq, nnz}; partials summed in fixed order
__global__ __launch_bounds__(256) void nrm_apply_kernel(float* __restrict__ vox, const double* __restrict__ stats,
                                                         int n_part, int64_t vol) {
    const int w = blockIdx.y;
    double s1 = 0, s2 = 0, nz = 0;
    for (int k = 0; k < n_part; ++k) {
        const double* p = stats + ((int64_t)w * n_part + k) * 3;
        s1 += p[0]; s2 += p[1]; nz += p[2];
    }
    if (nz <= 0.0) return;                          // identity when there are no non-zeros
    // eval.py:402-407 in fp32: sums are fp32 tensors, nnz an integer tensor
    const float nf = (float)nz;
    const float mean = (float)s1 / nf;
    const float ex2 = (float)s2 / nf;
    const float m2 = mean * mean;
    float sd = sqrtf(ex2 - m2);
    if (sd == sd) sd = fmaxf(sd, 1e-6f);            // torch.max propagates NaN
    float* v = vox + (int64_t)w * vol;
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    const bool vec = ((vol & 3) == 0) && ((((uintptr_t)vox) & 15) == 0);
    for (int64_t i = i0; i < vol; i += stride) {
        if (vec) {
            float4 a = *(float4*)&v[i];
            float r[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float mask = (r[k] != 0.f) ? 1.f : 0.f;
                const float d = r[k] - mean;
                const float md = mask * d;
                r[k] = md / sd;
            }
            *(float4*)&v[i] = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            for (int k = 0; k < 4 && i + k < vol; ++k) {
                const float a = v[i + k];
                const float mask = (a != 0.f) ? 1.f : 0.f;
                const float d = a - mean;
                const float md = mask * d;
                v[i + k] = md / sd;
            }
        }
    }
}

// ---------------------------------------------------------------- percentile normalization
__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

constexpr int PCT_THREADS = 1024;
constexpr int PCT_MAXR = 1;    // (register-resident keys spill under hipcc's full unroll; keys are re-read from L2 instead)

struct PctShared {
    unsigned hist[4][256];    // one histogram per radix level, memoised across the rank selects
    unsigned level_prefix[4]; // prefix each memoised histogram was counted under
    unsigned level_valid[4];
    unsigned bcast[2];
    unsigned wsum[4];
    float lohi[2];
};

// wave-aggregated histogram add: in the first radix level almost every key shares its top byte (sign +
// exponent), and 64 LDS atomics on one address serialise; a few leader rounds collapse them into one add each.
__device__ __forceinline__ void hist_add(unsigned* hist, unsigned digit, bool active, bool aggregate) {
    if (aggregate) {
        const int lane = threadIdx.x & 63;
        unsigned long long rem = __ballot(active);
        for (int it = 0; it < 4 && rem; ++it) {
            const int leader = __ffsll((long long)rem) - 1;
            const unsigned d = (unsigned)__shfl((int)digit, leader, 64);
            const unsigned long long m = __ballot(active && digit == d);
            if (lane == leader) atomicAdd(&hist[d], (unsigned)__popcll(m));
            if (digit == d) active = false;
            rem &= ~m;
        }
    }
    if (active) atomicAdd(&hist[digit], 1u);
}

// k-th smallest key (0-based) by MSB-first 8-bit radix select over keys held in registers (REG) or re-read
// from global memory; histograms are memoised per level so ranks that share a prefix share the counting pass.
template <bool REG>
__device__ __forceinline__ unsigned radix_select(const unsigned (&keys)[PCT_MAXR], int cnt, const float* __restrict__ v, int n, int k,
                                 PctShared& sh) {
    unsigned prefix = 0, pmask = 0;
    int kk = k;
    for (int level = 0; level < 4; ++level) {
        const int shift = 24 - 8 * level;
        unsigned* hist = sh.hist[level];
        const bool reuse = sh.level_valid[level] && sh.level_prefix[level] == prefix;   // block-uniform (LDS)
        __syncthreads();
        if (!reuse) {
            if (threadIdx.x < 256) hist[threadIdx.x] = 0;
            __syncthreads();
            if (REG) {
#pragma unroll
                for (int j = 0; j < PCT_MAXR; ++j) {
                    const bool in = j < cnt;
                    const unsigned key = keys[j];
                    hist_add(hist, (key >> shift) & 255u, in && (key & pmask) == prefix, level == 0);
                }
            } else {
                // 8 independent loads in flight per thread: the pass is L2-latency-bound, not bandwidth-bound
                for (int i0 = 0; i0 < n; i0 += PCT_THREADS * 8) {
                    float x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * PCT_THREADS + threadIdx.x;
                        x[u] = (i < n) ? v[i] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * PCT_THREADS + threadIdx.x;
                        const unsigned key = f2key(x[u]);
                        hist_add(hist, (key >> shift) & 255u, i < n && (key & pmask) == prefix, level == 0);
                    }
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) { sh.level_prefix[level] = prefix; sh.level_valid[level] = 1; }
            // deeper memoised levels were counted under another prefix chain
            if (threadIdx.x == 0) for (int l = level + 1; l < 4; ++l) sh.level_valid[l] = 0;
        }
        // parallel search of the bin holding rank kk: 256 threads scan the histogram (a serial walk by one
        // thread cost ~10 us of dependent LDS reads per level)
        unsigned c = 0, incl = 0;
        if (threadIdx.x < 256) {
            c = hist[threadIdx.x];
            incl = c;
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = (unsigned)__shfl_up((int)incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (lane == 63) sh.wsum[threadIdx.x >> 6] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            unsigned base = 0;
            for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += sh.wsum[w];
            const unsigned excl = base + incl - c;
            if (c > 0 && (unsigned)kk >= excl && (unsigned)kk < excl + c) { sh.bcast[0] = threadIdx.x; sh.bcast[1] = excl; }
        }
        __syncthreads();
        prefix |= sh.bcast[0] << shift;
        pmask |= 255u << shift;
        kk -= (int)sh.bcast[1];
    }
    __syncthreads();
    return prefix;
}

// numpy: q = q/100 in the array dtype; virtual index (n-1)*q; method 'linear' (_get_indexes/_get_gamma)
__device__ __forceinline__ void pct_rank(int n, float q100, int& prev, int& next, float& gamma) {
    const float q = q100 / 100.0f;
    const float vi = (float)(n - 1) * q;
    prev = (int)floorf(vi); next = prev + 1;
    if (vi >= (float)(n - 1)) { prev = next = n - 1; gamma = vi - (-1.0f); }
    else if (vi < 0.f) { prev = next = 0; gamma = vi - 0.0f; }
    else gamma = vi - (float)prev;
}

// The four selects a frame needs are the two neighbours of each percentile's virtual index.  pair = 0: one work-group per
// (order statistic, image), all independent (4 x n work-groups; the single-work-group form of round 1 walked them one after
// the other on 64 of 256 CUs).  pair = 1 (default): one work-group per (percentile, image) -- see below.
__global__ __launch_bounds__(PCT_THREADS) void pct_select_kernel(const float* __restrict__ img, int n, float q_lo, float q_hi,
                                                                  float* __restrict__ vals, int pair) {
    __shared__ PctShared sh;
    const float* v = img + (int64_t)blockIdx.y * n;
    if (threadIdx.x < 4) sh.level_valid[threadIdx.x] = 0;
    __syncthreads();
    int prev, next; float gamma;
    const unsigned keys[PCT_MAXR] = {0u};
    if (pair) {
        // one work-group per (percentile, image): ranks prev and prev + 1 share their counting passes through the memoised
        // histograms (they differ in the last radix level at most), so the second select is four histogram scans
        const int which = blockIdx.x;
        pct_rank(n, which ? q_hi : q_lo, prev, next, gamma);
        const float a = key2f(radix_select<false>(keys, 0, v, n, prev, sh));
        const float b = (next == prev) ? a : key2f(radix_select<false>(keys, 0, v, n, next, sh));
        if (threadIdx.x == 0) { vals[(int64_t)blockIdx.y * 4 + 2 * which] = a; vals[(int64_t)blockIdx.y * 4 + 2 * which + 1] = b; }
        return;
    }
    const int sel = blockIdx.x;
    pct_rank(n, (sel & 2) ? q_hi : q_lo, prev, next, gamma);
    const float x = key2f(radix_select<false>(keys, 0, v, n, (sel & 1) ? next : prev, sh));
    if (threadIdx.x == 0) vals[(int64_t)blockIdx.y * 4 + sel] = x;
}

// ---- round 5: the same two order statistics in ONE pass over the image -----------------------------------------------------------
// radix_select above reads the image once per radix level (four passes, each eleven dependent L2 round trips of eight loads per
// thread: ~25 us per pass, 108 us per 64-frame launch).  Here a work-group (one per percentile and image, as above):
//   1. reads PCT_S evenly spaced SAMPLE keys into LDS and selects two of their order statistics around the target rank (4 sigma of
//      the sample rank's binomial spread + a margin) -- thresholds lo_t <= hi_t that bracket the wanted keys with near certainty;
//   2. makes ONE pass over the image (16-B loads, four in flight per thread): counts the keys below lo_t and compacts the keys in
//      [lo_t, hi_t] into LDS (wave-aggregated slot allocation: order is irrelevant for a selection);
//   3. CHECKS the bracket -- below <= prev and next < below + candidates, list not overflowed -- and then selects ranks prev - below
//      and next - below among the ~1000 candidates in LDS.  The answer is exact whenever the check passes; otherwise (a constant
//      image, a pathological distribution) the work-group falls back to the four-pass select.  Nothing is approximate.
constexpr int PCT_S = 4096;        // sample keys
constexpr int PCT_CAP = 8192;      // candidate keys kept in LDS (32 KB; shares its space with the sample)
struct PctFast {
    unsigned list[PCT_CAP];
    unsigned hist[256];
    unsigned bcast[2];
    unsigned wsum[16];
    unsigned count, below, overflow;
};
// k-th smallest (0-based) of list[0..m) in LDS: MSB-first 8-bit radix select, every thread of the work-group takes part
__device__ __forceinline__ unsigned lds_select(const unsigned* list, int m, int k, PctFast& sh) {
    unsigned prefix = 0, pmask = 0;
    int kk = k;
    for (int level = 0; level < 4; ++level) {
        const int shift = 24 - 8 * level;
        __syncthreads();
        if (threadIdx.x < 256) sh.hist[threadIdx.x] = 0;
        __syncthreads();
        const int mr = (m + PCT_THREADS - 1) / PCT_THREADS * PCT_THREADS;      // whole waves take every trip: hist_add ballots
        for (int i = threadIdx.x; i < mr; i += PCT_THREADS) {
            const bool in = i < m;
            const unsigned key = in ? list[i] : 0u;
            hist_add(sh.hist, (key >> shift) & 255u, in && (key & pmask) == prefix, level == 0);
        }
        __syncthreads();
        unsigned c = 0, incl = 0;
        if (threadIdx.x < 256) {
            c = sh.hist[threadIdx.x];
            incl = c;
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = (unsigned)__shfl_up((int)incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (lane == 63) sh.wsum[threadIdx.x >> 6] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            unsigned base = 0;
            for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += sh.wsum[w];
            const unsigned excl = base + incl - c;
            if (c > 0 && (unsigned)kk >= excl && (unsigned)kk < excl + c) { sh.bcast[0] = threadIdx.x; sh.bcast[1] = excl; }
        }
        __syncthreads();
        prefix |= sh.bcast[0] << shift;
        pmask |= 255u << shift;
        kk -= (int)sh.bcast[1];
    }
    __syncthreads();
    return prefix;
}

// n % 4 == 0 and 16-B aligned images (checked by the launcher).  vals[image][4] = {lo.prev, lo.next, hi.prev, hi.next}
__global__ __launch_bounds__(PCT_THREADS) void pct_select_fast_kernel(const float* __restrict__ img, int n, float q_lo, float q_hi,
                                                                       float* __restrict__ vals) {
    __shared__ PctFast sh;
    __shared__ PctShared shs;       // the fallback's histograms
    const float* v = img + (int64_t)blockIdx.y * n;
    const int which = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    int prev, next; float gamma;
    pct_rank(n, which ? q_hi : q_lo, prev, next, gamma);
    if (tid == 0) { sh.count = 0; sh.below = 0; sh.overflow = 0; }
    // ---- 1. sample -> thresholds
    const int S = n < PCT_S ? n : PCT_S;
    for (int j = tid; j < S; j += PCT_THREADS) sh.list[j] = f2key(v[(int)(((long long)j * n) / S)]);
    const float sr = (float)prev * (float)S / (float)n;                       // expected sample rank of the target
    const float var = sr * (1.0f - sr / (float)S);
    const int d = 24 + (int)(4.0f * sqrtf(var > 0.f ? var : 0.f));
    const int r_lo = (int)sr - d, r_hi = (int)sr + d + 1;
    __syncthreads();
    const unsigned lo_t = r_lo <= 0 ? 0u : lds_select(sh.list, S, r_lo, sh);
    const unsigned hi_t = r_hi >= S - 1 ? 0xFFFFFFFFu : lds_select(sh.list, S, r_hi, sh);
    __syncthreads();                                                          // the sample is dead: its space becomes the candidate list
    // ---- 2. one pass: count below, compact the bracket
    unsigned below = 0;
    const float4* v4 = (const float4*)v;
    const int n4 = n >> 2;
    for (int i0 = 0; i0 < n4; i0 += PCT_THREADS * 4) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * PCT_THREADS + tid;
            x[u] = (i < n4) ? v4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = i0 + u * PCT_THREADS + tid < n4;
            const float e[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned key = f2key(e[c]);
                below += (in && key < lo_t) ? 1u : 0u;
                const bool cand = in && key >= lo_t && key <= hi_t;
                const unsigned long long m = __ballot(cand);
                if (m) {                                                      // wave-uniform
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(&sh.count, (unsigned)__popcll(m));
                    base = (unsigned)__shfl((int)base, 0, 64);
                    const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                    if (cand) { if (pos < (unsigned)PCT_CAP) sh.list[pos] = key; else sh.overflow = 1u; }
                }
            }
        }
    }
    below = (unsigned)evr_wave_sum((float)below);                             // (< 2^24 per wave: exact in fp32)
    if (lane == 0) atomicAdd(&sh.below, below);
    __syncthreads();
    const int nb = (int)sh.below, nc = (int)sh.count;
    const bool ok = !sh.overflow && nb <= prev && next < nb + nc;             // block-uniform
    float a, b;
    if (ok) {
        a = key2f(lds_select(sh.list, nc, prev - nb, sh));
        b = (next == prev) ? a : key2f(lds_select(sh.list, nc, next - nb, sh));
    } else {
        if (tid < 4) shs.level_valid[tid] = 0;
        __syncthreads();
        const unsigned keys[PCT_MAXR] = {0u};
        a = key2f(radix_select<false>(keys, 0, v, n, prev, shs));
        b = (next == prev) ? a : key2f(radix_select<false>(keys, 0, v, n, next, shs));
    }
    if (tid == 0) { vals[(int64_t)blockIdx.y * 4 + 2 * which] = a; vals[(int64_t)blockIdx.y * 4 + 2 * which + 1] = b; }
}

__global__ __launch_bounds__(256) void pct_exp_kernel(float* __restrict__ img, int64_t total) {      // 'exprobust' (eval.py:391-393)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) img[i] = expf(img[i]);
}

__global__ __launch_bounds__(256) void pct_apply_kernel(float* __restrict__ img, int n, float q_lo, float q_hi,
                                                         const float* __restrict__ vals) {
    float* v = img + (int64_t)blockIdx.y * n;
    const float* val = vals + (int64_t)blockIdx.y * 4;
    float res[2];
#pragma unroll
    for (int which = 0; which < 2; ++which) {   // numpy _lerp
        int prev, next; float gamma;
        pct_rank(n, which ? q_hi : q_lo, prev, next, gamma);
        const float a = val[2 * which], b = val[2 * which + 1];
        const float diff = b - a;
        float r = a + diff * gamma;
        if (gamma >= 0.5f) r = b - diff * (1.0f - gamma);
        res[which] = r;
    }
    const float lo = res[0], range = res[1] - res[0];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { const float d = v[i] - lo; v[i] = d / range; }
}

}  // namespace

extern "C" int evr_event_tensor_normalize(float* vox, int n, int B, int H, int W, const double* stats,
                                          void* workspace, size_t workspace_bytes, evr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EVR_REQUIRE(n >= 0 && B >= 1 && H >= 1 && W >= 1, "evr_event_tensor_normalize: bad shape");
    if (n == 0) return EVR_OK;
    EVR_REQUIRE(vox != nullptr, "evr_event_tensor_normalize: null tensor");
    const int64_t vol = (int64_t)B * H * W;
    int n_part = 1;
    if (!stats) {
        const size_t need = (size_t)n * NRM_BLOCKS * 3 * sizeof(double);
        if (!workspace || workspace_bytes < need) {
            evr::set_error("evr_event_tensor_normalize: workspace %zu B < required %zu B", workspace_bytes, need);
            return EVR_ERR_WORKSPACE;
        }
        hipLaunchKernelGGL(nrm_partial_kernel, dim3(NRM_BLOCKS, n), dim3(256), 0, stream, vox, (double*)workspace, vol);
        EVR_LAUNCH_CHECK();
        stats = (const double*)workspace;
        n_part = NRM_BLOCKS;
    }
    int gx = (int)((vol / 4 + 255) / 256);
    if (gx > 128) gx = 128;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(nrm_apply_kernel, dim3(gx, n), dim3(256), 0, stream, vox, stats, n_part, vol);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

extern "C" size_t evr_percentile_normalize_workspace_bytes(int n, int H, int W) {
    (void)H; (void)W;
    return n > 0 ? (size_t)n * 4 * sizeof(float) : 0;      // the four order statistics of every image
}

extern "C" int evr_percentile_normalize(float* img, int n, int H, int W, float q_lo, float q_hi, int do_exp,
                                        void* workspace, size_t workspace_bytes, evr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EVR_REQUIRE(n >= 0 && H >= 1 && W >= 1, "evr_percentile_normalize: bad shape");
    EVR_REQUIRE(q_lo >= 0.f && q_hi <= 100.f && q_lo <= q_hi, "evr_percentile_normalize: percentiles must be in [0,100]");
    EVR_REQUIRE((int64_t)H * W < (1LL << 30), "evr_percentile_normalize: image too large");
    if (n == 0) return EVR_OK;
    EVR_REQUIRE(img != nullptr, "evr_percentile_normalize: null image");
    const size_t need = evr_percentile_normalize_workspace_bytes(n, H, W);
    if (!workspace || workspace_bytes < need) {
        evr::set_error("evr_percentile_normalize: workspace %zu B < required %zu B", workspace_bytes, need);
        return EVR_ERR_WORKSPACE;
    }
    const int px = H * W;
    if (do_exp) {
        const int64_t total = (int64_t)n * px;
        int64_t blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(pct_exp_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, img, total);
        EVR_LAUNCH_CHECK();
    }
    // two work-groups per image (one per percentile, both ranks each) instead of four (one per rank): half the image passes;
    // +0.3 .. +0.5 % on the headline in three A/B pairs on one box.  EVR_PCT_PAIR=0 restores the four-work-group form.
    static const int pair = [] { const char* e = getenv("EVR_PCT_PAIR"); return e ? atoi(e) : 1; }();
    // round 5: the one-pass sampled-bracket select (exact: the bracket is verified, the four-pass select is its fallback);
    // EVR_PCT_FAST=0 restores the four-pass form
    static const int fast = [] { const char* e = getenv("EVR_PCT_FAST"); return e ? atoi(e) : 1; }();
    if (fast && px % 4 == 0 && px >= 1024 && (((uintptr_t)img) & 15) == 0)
        hipLaunchKernelGGL(pct_select_fast_kernel, dim3(2, n), dim3(PCT_THREADS), 0, stream, img, px, q_lo, q_hi, (float*)workspace);
    else
        hipLaunchKernelGGL(pct_select_kernel, dim3(pair ? 2 : 4, n), dim3(PCT_THREADS), 0, stream, img, px, q_lo, q_hi, (float*)workspace, pair);
    EVR_LAUNCH_CHECK();
    int gx = (px + 256 * 8 - 1) / (256 * 8);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(pct_apply_kernel, dim3(gx, n), dim3(256), 0, stream, img, px, q_lo, q_hi, (const float*)workspace);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

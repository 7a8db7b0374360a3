// Events -> voxel grid on gfx950, bit-identical to the reference's CPU path.
//
// Reference: utils/event_utils.py:27-59 (events_to_voxel_torch), :4-24 (events_to_image_torch,
// sequential index_put_(accumulate=True)); called from dataset.py:205-216.
//
// The reference adds, per (bin, pixel) cell, the fp32 weights of the events that hit it IN EVENT
// ORDER.  Float atomics would break that order, so every cell is accumulated in LDS by a workgroup
// that sees the cell's events in time order.  The pixel plane is cut into G RANGES of `rows`
// consecutive sensor rows (B x rows x W cells = one workgroup's LDS: 4 rows = 27.7 KB at 346x260, B = 5).
//
//   K1 vox_split   several 512-thread workgroups per window, each taking SEGMENTS of 2048 consecutive
//                  events: one load round trip fetches coordinates, timestamps and polarities (12
//                  independent loads per lane), then every event is finished ONCE -- t_norm with the
//                  reference's exact fp32 operation order, its range and its cell offset -- and the
//                  segment is stably partitioned by range (ballot ranks inside a wave, per-wave LDS
//                  histograms, one prefix) into records -- 8 B {t_norm, offset | polarity byte << 24} in the
//                  raw form, 16 B {t_norm, p, offset} in the fp32 form.  A per-segment table of G+1
//                  offsets says where each range's records sit.
//   K2 vox_range   one 256-thread workgroup per (window, range); five fit a CU.  It reads its
//                  slice of every segment table (one round trip), then its records (a second one,
//                  ~N/G of them, contiguous per segment, in time order), and accumulates them one
//                  record per thread: threads that hit the same pixel are serialised
//                  lowest-thread-first through an LDS atomic-min ticket, so every cell sees its adds
//                  in event order.  Only the (at most two) bins with a non-zero weight are touched --
//                  adding the reference's +-0 products never changes a cell (cells start at +0).  The
//                  finished range streams out with 16-B NON-TEMPORAL stores (the grid is consumed much
//                  later and is larger than the caches: streamed past the L2 it no longer evicts the
//                  records its neighbours are about to read): every output cell is written exactly
//                  once, zero fill included, and nothing is ever read back from HBM.  Per-range
//                  {sum, sum of squares, nnz} feed eval.py:402-405.
//   K3 vox_stats   fixed-order reduction of the per-range partials -> stats[w][3] (deterministic).
//
// HBM-side traffic per window: the events once (13 N raw / 16 N fp32) + 4 B H W (output) -- the
// algorithmic bytes of SURVEY 8d -- plus 2 x 8 N (raw) / 2 x 16 N (fp32 form) of records that normally stay in the
// L2 / Infinity Cache.
//
// fp32 arithmetic is one IEEE rounding per op (this file is built with -ffp-contract=off; HIP's
// default correctly-rounded fp32 divide is kept), matching torch's CPU kernels.
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "common.h"

namespace {

constexpr int K1T = 512;            // threads of a split workgroup
constexpr int K1W = K1T / 64;       // its waves
constexpr int SEG = 2048;           // events per segment: 4 per thread, event (wave k, slot j, lane l) = k*256 + j*64 + l
constexpr int K2T = 256;            // threads of a range workgroup
constexpr int K2W = K2T / 64;
constexpr int NTAG = 512;         // ticket slots of a range workgroup (hashed by pixel)
constexpr int MAXSEG = 64;          // segment tables folded per pass of the range kernel
constexpr int ACC_KB_DEFAULT = 34;  // LDS for the B x rows x W cells of a range (5 rows of 346: four workgroups per CU)
constexpr int MAX_G = 4096;

struct VoxHeader {           // first 256 B of the workspace; must be zero before the workspace is used for the first time
    unsigned long long dropped_acc;    // out-of-sensor events counted by the running call (K1 adds, K2 folds and clears)
    unsigned long long dropped_last;   // ... of the last completed call (evr_voxelize_dropped reads this)
    unsigned long long dropped_total;  // ... of every call since the header was zeroed (evr_voxelize_dropped_total)
};

// torch.linspace(0, B-1, n)[i], ATen scalar formula (oracle/voxel.py:linspace_f32)
__device__ __forceinline__ float lin_tnorm(int i, int n, int B) {
    if (n == 1) return 0.0f;
    const float end = (float)(B - 1);
    const float step = end / (float)(n - 1);
    if (i < n / 2) {
        float m = step * (float)i;
        return 0.0f + m;
    }
    float m = step * (float)(n - i - 1);
    return end - m;
}

struct EventSrc {
    const float* x; const float* y; const float* t; const float* p;   // fp32 form
    const int16_t* xy; const double* ts; const uint8_t* pol;          // raw form
};

// first segment-table row of window w: floor(rec_base / SEG) + w  (>= the number of segments of all earlier windows)
__device__ __forceinline__ int64_t table_row0(int64_t rec_base_w, int w) { return rec_base_w / SEG + w; }

// ------------------------------------------------------------------------------------------------ K1
template <bool RAW>
__global__ __launch_bounds__(K1T) void vox_split_kernel(
    EventSrc src, const int64_t* __restrict__ win_begin, const int64_t* __restrict__ win_end,
    const int64_t* __restrict__ rec_base, float4* __restrict__ rec, int* __restrict__ table, VoxHeader* hdr,
    int S, int G, int rows, unsigned rows_magic, int B, int H, int W, int w0) {
    extern __shared__ int smem[];
    int* hist = smem;                  // [K1W][G]  per-wave counts, then exclusive offsets of the wave inside its range
    int* bbase = smem + K1W * G;       // [G + 1]   first record of each range inside the segment
    int* wsum = bbase + G + 1;         // [K1W]

    const int w = w0 + blockIdx.x / S, s0 = blockIdx.x % S;      // (w0: first window of this launch's chunk, voxelize_impl)
    const int tid = threadIdx.x, lane = tid & 63, k = tid >> 6;
    const int64_t a = win_begin[w];
    const int64_t ne = win_end[w] - a;
    const int n = ne > 0 ? (int)ne : 0;
    if ((int64_t)s0 * SEG >= n) return;
    const int64_t rb = rec_base[w];
    const int64_t row0 = table_row0(rb, w);

    float t0 = 0.f, dt = 0.f;
    double t0d = 0.0;
    if (RAW) {
        t0d = src.ts[a];
        dt = (float)(src.ts[a + n - 1] - t0d) - 0.0f;
    } else {
        t0 = src.t[a];
        dt = src.t[a + n - 1] - t0;
    }
    const bool lin = ((double)dt < 1e-9);
    const float bm1 = (float)(B - 1);
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int nbits = 0;
    while ((1 << nbits) < G) ++nbits;
    unsigned n_bad = 0;

    for (int sg = s0; (int64_t)sg * SEG < n; sg += S) {
        for (int i = tid; i < K1W * G; i += K1T) hist[i] = 0;

        // ---- one round trip: everything the segment's events need (clamped indices: every load is unconditional) ----
        int rel[4]; bool ok[4];
        int xi[4], yi[4]; float tn[4], pv[4]; unsigned pbyte[4] = {0u, 0u, 0u, 0u};
        if (RAW) {
            uint32_t wd[4]; double td[4]; uint8_t pb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rel[j] = sg * SEG + k * 256 + j * 64 + lane;
                ok[j] = rel[j] < n;
                const int64_t i = a + (ok[j] ? rel[j] : n - 1);
                const uint16_t* h = (const uint16_t*)src.xy + 2 * i;
                wd[j] = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
                td[j] = src.ts[i];
                pb[j] = src.pol[i];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xi[j] = (int16_t)(wd[j] & 0xFFFFu); yi[j] = (int16_t)(wd[j] >> 16);
                const float tf = (float)(td[j] - t0d);                   // dataset.py:56 (f64 subtract, cast)
                if (lin) tn[j] = lin_tnorm(rel[j], n, B);
                else { float q = (tf - 0.0f) / dt; tn[j] = q * bm1; }    // ts[0] is exactly 0 after the shift
                pv[j] = 0.f; pbyte[j] = pb[j];                          // p = 2 pol - 1 (dataset.py:227) is formed by the range kernel
            }
        } else {
            float fx[4], fy[4], ft[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rel[j] = sg * SEG + k * 256 + j * 64 + lane;
                ok[j] = rel[j] < n;
                const int64_t i = a + (ok[j] ? rel[j] : n - 1);
                fx[j] = src.x[i]; fy[j] = src.y[i]; ft[j] = src.t[i]; pv[j] = src.p[i];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (long long)f truncates toward zero (.long() in the reference): the result is in [0, W) exactly when -1 < f < W
                const bool in = (fx[j] > -1.0f) & (fx[j] < (float)W) & (fy[j] > -1.0f) & (fy[j] < (float)H);
                xi[j] = in ? (int)fx[j] : -1; yi[j] = in ? (int)fy[j] : -1;
                if (lin) tn[j] = lin_tnorm(rel[j], n, B);
                else { float d = ft[j] - t0; float q = d / dt; tn[j] = q * bm1; }
            }
        }
        __syncthreads();                       // hist is zero

        // ---- stable rank of every event inside (wave, range): order (slot j, lane) = event order ----
        int bkt[4], rank[4]; unsigned off[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool valid = ok[j] && (xi[j] >= 0) && (xi[j] < W) && (yi[j] >= 0) && (yi[j] < H);
            n_bad += ok[j] && !valid;
            const int b = valid ? (rows_magic ? (int)__umulhi((unsigned)yi[j], rows_magic) : yi[j]) : 0;     // yi / rows
            bkt[j] = valid ? b : -1;
            off[j] = valid ? (unsigned)((yi[j] - b * rows) * W + xi[j]) : 0u;
            unsigned long long peers = __ballot(valid);
            for (int bit = 0; bit < nbits; ++bit) {
                const bool s = (b >> bit) & 1;
                const unsigned long long m = __ballot(s);
                peers &= s ? m : ~m;
            }
            volatile int* vh = hist;           // other lanes of the wave update these counters between slots
            int base = 0;
            if (valid) base = vh[k * G + b];
            __builtin_amdgcn_wave_barrier();
            if (valid && (peers & lt_mask) == 0) vh[k * G + b] = base + __popcll(peers);        // lowest peer lane
            __builtin_amdgcn_wave_barrier();
            rank[j] = base + __popcll(peers & lt_mask);
        }
        __syncthreads();

        // ---- offsets: waves inside a range, then ranges inside the segment ----
        int carry = 0;                         // records of the ranges below this pass (uniform)
        for (int b0 = 0; b0 < G; b0 += K1T) {
            const int b = b0 + tid;
            int total = 0;
            if (b < G) {
#pragma unroll
                for (int q = 0; q < K1W; ++q) {
                    const int c = hist[q * G + b];
                    hist[q * G + b] = total;
                    total += c;
                }
            }
            int incl = total;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            if (lane == 63) wsum[k] = incl;
            __syncthreads();
            int wbase = carry, all = 0;
#pragma unroll
            for (int q = 0; q < K1W; ++q) {
                const int v = wsum[q];
                wbase += (q < k) ? v : 0;
                all += v;
            }
            if (b < G) bbase[b] = wbase + incl - total;
            carry += all;
            __syncthreads();                   // wsum is reused by the next pass; bbase complete after the last
        }
        if (tid == 0) bbase[G] = carry;
        __syncthreads();
        int* trow = table + (row0 + sg) * (int64_t)(G + 1);
        for (int b = tid; b <= G; b += K1T) trow[b] = bbase[b];

        // ---- scatter the records ----
        // raw form: 8-B records {t_norm, offset | polarity byte << 24} (offset < 2^15); fp32 form: 16-B {t_norm, p, offset, -}
        float4* srec = rec + rb + (int64_t)sg * SEG;
        float2* srec2 = (float2*)rec + rb + (int64_t)sg * SEG;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (bkt[j] >= 0) {
                const int pos = bbase[bkt[j]] + hist[k * G + bkt[j]] + rank[j];
                if (RAW) srec2[pos] = make_float2(tn[j], __uint_as_float(off[j] | (pbyte[j] << 24)));
                else srec[pos] = make_float4(tn[j], pv[j], __uint_as_float(off[j]), 0.f);
            }
        }
        __syncthreads();                       // before the next segment re-zeroes hist
    }
    if (n_bad) atomicAdd(&hdr->dropped_acc, (unsigned long long)n_bad);
}

// ------------------------------------------------------------------------------------------------ K2
// SCAN (round 6, timing experiment only -- EVR_VOX_SCANPROBE=1 / 2): VERDICT r5 proposed ONE kernel per (window, range) that scans the
// window's events itself (a wave-ballot range filter over the 4-byte coordinate words) instead of reading the records a split kernel
// sorted.  The probe adds exactly that scan to this kernel -- every work-group loads all coordinate words of its window, tests them
// against its rows and ballots; the result only lands in a spare LDS word -- so that (probe range kernel) bounds the fused kernel's
// time from below at one launch: it would still need this kernel's zero / ticket / flush phases and, per in-range event, a gather
// of {coordinates, timestamp, polarity} instead of one 8-byte record.  Numbers: DESIGN.md section 4.1.
template <bool RAW, int SCAN = 0>
__global__ __launch_bounds__(K2T, 3) void vox_range_kernel(
    const int64_t* __restrict__ win_begin, const int64_t* __restrict__ win_end, const int64_t* __restrict__ rec_base,
    const float4* __restrict__ rec, const int* __restrict__ table, float* __restrict__ out,
    double* __restrict__ partials, VoxHeader* hdr, int n_windows, int G, int rows, int Rp, int B, int H, int W, int vec_out, int xcd_map,
    unsigned rq_magic, int w0, int publish, const int16_t* __restrict__ scan_xy = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (n_windows: END of this launch's chunk [w0, n_windows); publish: the call's LAST range launch -- every split launch of the call
    // has finished by then, voxelize_impl orders them with events)
    if (publish && blockIdx.x == 0 && threadIdx.x == 0) {       // K1 has finished (kernel boundary): publish its count, re-arm the counter
        hdr->dropped_last = hdr->dropped_acc;
        hdr->dropped_total += hdr->dropped_acc;
        hdr->dropped_acc = 0;
    }
    float* acc = lds;                                          // [B][Rp]   Rp = rows * W rounded up to 4
    unsigned* tags = (unsigned*)(acc + (size_t)B * Rp);        // [NTAG] ticket slots (hashed by pixel)
    int* seg_pre = (int*)(tags + NTAG);                        // [MAXSEG + 1] records of this range before segment s
    int* seg_at = seg_pre + MAXSEG + 1;                        // [MAXSEG]     where they start inside the segment
    int* more = seg_at + MAXSEG;                               // [3] "another ticket round is needed" flags (+2 pad: `red` is 8-B aligned)
    double* red = (double*)(more + 5);                         // [K2W][3]  byte offset 4*(B*Rp + NTAG + 2*MAXSEG + 1 + 5): B*Rp % 4 == 0 -> a multiple of 8

    const int tid = threadIdx.x, lane = tid & 63, k = tid >> 6;
    int w, g;
    if (xcd_map) {      // blocks b, b+8, b+16, ... run on one XCD: give them the ranges of the same windows
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        w = w0 + (slot / G) * 8 + xcd; g = slot % G;
    } else {
        w = w0 + blockIdx.x / G; g = blockIdx.x % G;
    }
    if (w >= n_windows) return;
    const int64_t HW = (int64_t)H * W;
    const int y0 = g * rows;
    const int nrows = min(rows, H - y0);
    const int64_t ne = win_end[w] - win_begin[w];
    const int n = ne > 0 ? (int)ne : 0;
    const int nseg = (n + SEG - 1) / SEG;
    const int64_t rb = rec_base[w];
    const int* trow = table + table_row0(rb, w) * (int64_t)(G + 1) + g;

    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = tid * 4; j < B * Rp; j += K2T * 4) *(float4*)&acc[j] = z4;
    for (int j = tid; j < NTAG; j += K2T) tags[j] = 0xFFFFFFFFu;
    if (tid < 4) more[tid] = 0;
    volatile unsigned* tag = (volatile unsigned*)tags;
    volatile int* vmore = (volatile int*)more;
    int round = 0;                            // ticket rounds so far (selects the flag; uniform)
    if constexpr (SCAN != 0) {
        // wave k scans the k-th quarter of the window in 64-event strides, 16 loads in flight per lane (SCAN == 2: the y halves only)
        const int64_t a0 = win_begin[w];
        const int per = (n + K2W - 1) / K2W;
        const int lo = k * per, hi = min(n, lo + per);
        int cnt = 0;
        for (int e0 = lo; e0 < hi; e0 += 64 * 16) {
            unsigned wd[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int e = e0 + j * 64 + lane;
                const int64_t i = a0 + (e < hi ? e : hi - 1);
                if (SCAN == 2) wd[j] = (unsigned)((const uint16_t*)scan_xy)[2 * i + 1] << 16;
                else wd[j] = ((const unsigned*)scan_xy)[i];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int e = e0 + j * 64 + lane;
                const int yy = (int)(wd[j] >> 16), xx = (int)(wd[j] & 0xFFFFu);
                const bool in = e < hi && (unsigned)(yy - y0) < (unsigned)nrows && xx < W;
                cnt += __popcll(__ballot(in));
            }
        }
        if (lane == 0) vmore[4] = cnt;        // (a spare word: keeps the scan alive, read by nobody)
    }

    for (int sg0 = 0; sg0 < nseg; sg0 += MAXSEG) {
        const int ns = min(MAXSEG, nseg - sg0);
        __syncthreads();                      // LDS initialised / the previous group's seg_* no longer read
        // ---- where this range's records sit in each segment (first wave; one round trip) ----
        if (k == 0) {
            int c0 = 0, c1 = 0;
            if (lane < ns) {
                const int* t = trow + (int64_t)(sg0 + lane) * (G + 1);
                c0 = t[0]; c1 = t[1];
            }
            int incl = c1 - c0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            seg_pre[lane + 1] = incl;
            seg_at[lane] = c0;
            if (lane == 0) seg_pre[0] = 0;
        }
        __syncthreads();
        const int tot = seg_pre[ns];

        // ---- accumulate: one record per thread, K2T at a time; threads that hit the same pixel are serialised
        //      lowest-thread-first (= event order) through an LDS atomic-min ticket ----
        for (int c0 = 0; c0 < tot; c0 += K2T) {
            const int i = c0 + tid;
            bool pending = i < tot;
            float tn = 0.f, p = 0.f; unsigned pl = 0;
            if (pending) {
                int s = 0;                                  // segment of record i: seg_pre[s] <= i < seg_pre[s+1]
                for (int step = MAXSEG / 2; step > 0; step >>= 1)
                    if (s + step < ns && seg_pre[s + step] <= i) s += step;
                const int64_t ri = rb + (int64_t)(sg0 + s) * SEG + seg_at[s] + (i - seg_pre[s]);
                if (RAW) {
                    const float2 r = ((const float2*)rec)[ri];
                    const unsigned wd = __float_as_uint(r.y);
                    tn = r.x; pl = wd & 0xFFFFFFu;
                    p = (float)((double)(wd >> 24) * 2.0 - 1.0);             // dataset.py:227
                } else {
                    const float4 r = rec[ri];
                    tn = r.x; p = r.y; pl = __float_as_uint(r.z);
                }
            }
            const unsigned h = pl & (NTAG - 1);
            for (;; ++round) {
                const int f = round % 3;
                if (pending) atomicMin((unsigned*)&tag[h], (unsigned)tid);
                if (tid == 0) vmore[(round + 1) % 3] = 0;
                __syncthreads();
                if (pending && tag[h] == (unsigned)tid) {
                    // event_utils.py:53-56, one rounding per op; bins with weight 0 add +-0: no effect, skipped
                    float bf = floorf(tn);
                    bf = fminf(fmaxf(bf, -2.0f), (float)B);
                    const int bb = (int)bf;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int b = bb + s2;
                        if (b >= 0 && b < B) {
                            const float d = tn - (float)b;
                            const float wgt = 1.0f - fabsf(d);
                            if (wgt > 0.0f) {
                                const float v = p * wgt;
                                volatile float* cell = &acc[b * Rp + pl];
                                const float sum = *cell + v;
                                *cell = sum;
                            }
                        }
                    }
                    tag[h] = 0xFFFFFFFFu;
                    pending = false;
                }
                if (__ballot(pending) && lane == 0) vmore[f] = 1;
                __syncthreads();
                if (!vmore[f]) { ++round; break; }
            }
        }
    }
    __syncthreads();

    // ---- stream the range out (each output cell written once) + its statistics ----
    // Per-thread partial sums in fp32 over a few dozen cells, combined in fp64 in a fixed order: deterministic, and far
    // inside the 2e-6 the reference's own thread-dependent torch.sum order leaves (tests/test_gpu_prepost.py).
    float f1 = 0.f, f2 = 0.f; int fz = 0;
    const int Rc = nrows * W;
    float* o = out + (int64_t)w * B * HW + (int64_t)y0 * W;
    int nzw = 0;                       // non-zero cells counted per wave (uniform), fast path
    if (vec_out && rq_magic && nrows == rows) {      // (a last range with fewer rows takes the general loop)
        // 16-B groups of all bins flattened (index i = group j of bin b = i / Rq by multiply-high), four per thread and
        // pass: the LDS reads of a pass are in flight together and its stores leave back to back.  The kernel issues VALU
        // instructions for 70 % of its cycles (profiles/r03_voxelizer_pmc.md), so the statistics ride on packed fp32
        // adds / FMAs and a ballot count instead of 20 scalar-per-lane operations per group.
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef float f2v __attribute__((ext_vector_type(2)));
        const unsigned Rq = (unsigned)Rc >> 2, total = (unsigned)B * Rq;
        f2v s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
        for (unsigned base = 0; base < total; base += 4 * K2T) {
            f4v v[4]; unsigned go[4]; bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned i = base + u * K2T + tid;
                ok[u] = i < total;
                const unsigned ii = ok[u] ? i : 0u;
                const unsigned bb = __umulhi(ii, rq_magic), j4 = (ii - bb * Rq) * 4u;
                go[u] = bb * (unsigned)HW + j4;
                v[u] = *(const f4v*)&acc[bb * (unsigned)Rp + j4];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ok[u]) {
                    // non-temporal: the grid is consumed much later (and is larger than the caches); streamed past the L2, it
                    // no longer evicts the records and segment tables the range work-groups are about to read
                    __builtin_nontemporal_store(v[u], (f4v*)&o[go[u]]);
                    const f2v lo = {v[u].x, v[u].y}, hi = {v[u].z, v[u].w};
                    s1p += lo; s1p += hi;
                    s2p = __builtin_elementwise_fma(lo, lo, s2p); s2p = __builtin_elementwise_fma(hi, hi, s2p);
                }
                nzw += __popcll(__ballot(ok[u] && v[u].x != 0.f)) + __popcll(__ballot(ok[u] && v[u].y != 0.f)) +
                       __popcll(__ballot(ok[u] && v[u].z != 0.f)) + __popcll(__ballot(ok[u] && v[u].w != 0.f));
            }
        }
        f1 = s1p.x + s1p.y; f2 = s2p.x + s2p.y;
    } else {
        for (int b = 0; b < B; ++b) {
            for (int j = tid * 4; j < Rc; j += K2T * 4) {
                const float4 v = *(const float4*)&acc[b * Rp + j];
                if (vec_out && j + 4 <= Rc) {
                    *(float4*)&o[(int64_t)b * HW + j] = v;
                } else {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
                    for (int q = 0; q < 4; ++q)
                        if (j + q < Rc) o[(int64_t)b * HW + j + q] = vv[q];
                }
                // cells beyond the range end stay zero in LDS, so they do not disturb the statistics
                f1 += (v.x + v.y) + (v.z + v.w);
                f2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                fz += (v.x != 0.f) + (v.y != 0.f) + (v.z != 0.f) + (v.w != 0.f);
            }
        }
    }
    if (partials) {
        double s1 = evr_wave_sum((double)f1), s2 = evr_wave_sum((double)f2), nz = evr_wave_sum((double)fz) + (double)nzw;
        if (lane == 0) { red[k * 3 + 0] = s1; red[k * 3 + 1] = s2; red[k * 3 + 2] = nz; }
        __syncthreads();
        if (tid < 3) {
            double t = 0.0;
            for (int q = 0; q < K2W; ++q) t += red[q * 3 + tid];      // fixed order: deterministic
            partials[((int64_t)w * G + g) * 3 + tid] = t;
        }
    }
}

__global__ __launch_bounds__(256) void vox_stats_kernel(const double* __restrict__ partials,
                                                         double* __restrict__ stats, int n_parts) {
    __shared__ double sh[3][4];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int t = tid; t < n_parts; t += 256) {
        const double* p = partials + ((int64_t)w * n_parts + t) * 3;
        a0 += p[0]; a1 += p[1]; a2 += p[2];
    }
    a0 = evr_wave_sum(a0); a1 = evr_wave_sum(a1); a2 = evr_wave_sum(a2);
    if (lane == 0) { sh[0][wave] = a0; sh[1][wave] = a1; sh[2][wave] = a2; }
    __syncthreads();
    if (tid < 3) stats[w * 3 + tid] = ((sh[tid][0] + sh[tid][1]) + sh[tid][2]) + sh[tid][3];
}

// ------------------------------------------------------------------------------------------------ host
struct VoxPlan {
    int rows, G, Rp;        // sensor rows per range, ranges per window, cells per bin in LDS (rows * W rounded up to 4)
    size_t lds1, lds2, off_rec, off_table, off_partials, total;
};

int acc_kb() {
    const char* e = getenv("EVR_VOX_ACC_KB");      // tuning knob: LDS for the cells -> rows per range -> workgroups per CU
    const int v = e ? atoi(e) : ACC_KB_DEFAULT;
    return v < 8 ? 8 : (v > 128 ? 128 : v);
}

bool make_plan(int64_t n_events_total, int n_windows, int B, int H, int W, VoxPlan& p) {
    const int64_t budget = (int64_t)acc_kb() * 1024;
    int rows = (int)(budget / (4LL * B * W));
    if (rows < 1) {                                 // a single row of B bins must fit even when the knob is small
        if (4LL * B * W > 128 * 1024) return false;
        rows = 1;
    }
    if (rows > H) rows = H;
    int G = (H + rows - 1) / rows;
    rows = (H + G - 1) / G;                         // balance the ranges
    // 16-B stores need every range to start on a multiple of 4 cells: prefer a row count that gives one
    if (((int64_t)rows * W) % 4 != 0 && rows < H) {
        for (int r = rows; r >= 1 && r > rows - 4; --r)
            if (((int64_t)r * W) % 4 == 0) { rows = r; break; }
    }
    G = (H + rows - 1) / rows;
    if (G > MAX_G) return false;
    p.rows = rows; p.G = G;
    p.Rp = (int)(((int64_t)rows * W + 3) & ~3LL);
    p.lds1 = (size_t)(K1W * G + G + 1 + K1W) * sizeof(int);
    p.lds2 = (size_t)B * p.Rp * 4 + (size_t)NTAG * 4 + (size_t)(2 * MAXSEG + 1 + 5) * 4 + K2W * 3 * 8 + 8;
    size_t off = 256;
    p.off_rec = off; off += evr::align_up((size_t)(n_events_total > 0 ? n_events_total : 1) * sizeof(float4), 256);
    const size_t n_rows = (size_t)(n_events_total / SEG) + n_windows + 2;
    p.off_table = off; off += evr::align_up(n_rows * (G + 1) * sizeof(int), 256);
    p.off_partials = off; off += evr::align_up((size_t)n_windows * G * 3 * sizeof(double), 256);
    p.total = off;
    return true;
}

// internal second stream of the chunked form (voxelize_impl), one per device, created on first use and kept for the process
constexpr int MAX_CHUNKS = 16;
struct SideStream { hipStream_t s = nullptr; hipEvent_t fork = nullptr; hipEvent_t done[MAX_CHUNKS] = {};
                    std::mutex mu; };      // (one enqueue at a time per device, whichever form of the call -- ADVICE r5: a `static` mutex inside the
                                           //  templated caller gave the raw and the fp32 instantiation one each over the SAME stream and events)
SideStream* side_stream(int dev) {
    static SideStream pool[64];
    static std::atomic<int> state[64];      // 0 none, 1 being created, 2 ready, 3 failed
    if (dev < 0 || dev >= 64) return nullptr;
    int st = state[dev].load(std::memory_order_acquire);
    if (st == 2) return &pool[dev];
    if (st == 3) return nullptr;
    int expect = 0;
    if (!state[dev].compare_exchange_strong(expect, 1)) {      // another thread is creating it: this call runs unchunked
        return state[dev].load(std::memory_order_acquire) == 2 ? &pool[dev] : nullptr;
    }
    SideStream& p = pool[dev];
    bool ok = hipStreamCreateWithFlags(&p.s, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < MAX_CHUNKS; ++i) ok = hipEventCreateWithFlags(&p.done[i], hipEventDisableTiming) == hipSuccess;
    state[dev].store(ok ? 2 : 3, std::memory_order_release);
    return ok ? &p : nullptr;
}

int split_groups_env() {
    const char* e = getenv("EVR_VOX_SPLIT");       // tuning knob: split workgroups per window
    return e ? atoi(e) : 0;
}

template <bool RAW>
int voxelize_impl(const EventSrc& src, const int64_t* win_begin, const int64_t* win_end, const int64_t* rec_base,
                  int n_windows, int64_t n_events_total,
                  int B, int H, int W, float* out, double* stats, void* workspace, size_t ws_bytes,
                  hipStream_t stream) {
    EVR_REQUIRE(n_windows >= 0 && B >= 1 && H >= 1 && W >= 1, "evr_voxelize: bad shape n_windows=%d B=%d H=%d W=%d", n_windows, B, H, W);
    EVR_REQUIRE(H < 32768 && W < 32768 && (int64_t)H * W < (1LL << 30), "evr_voxelize: sensor %dx%d too large", W, H);
    EVR_REQUIRE(n_events_total >= 0 && n_events_total < (1LL << 31), "evr_voxelize: n_events_total out of range");
    if (n_windows == 0) return EVR_OK;
    EVR_REQUIRE(win_begin && win_end && rec_base && out && workspace, "evr_voxelize: null pointer");
    VoxPlan pl;
    EVR_REQUIRE(make_plan(n_events_total, n_windows, B, H, W, pl), "evr_voxelize: B=%d bins x %d columns do not fit the LDS cell budget", B, W);
    if (ws_bytes < pl.total) {
        evr::set_error("evr_voxelize: workspace %zu B < required %zu B", ws_bytes, pl.total);
        return EVR_ERR_WORKSPACE;
    }
    EVR_REQUIRE(pl.lds2 <= 160 * 1024 && pl.lds1 <= 160 * 1024, "evr_voxelize: B=%d needs %zu B of LDS (> 160 KiB)", B, pl.lds2);
    char* ws = (char*)workspace;
    VoxHeader* hdr = (VoxHeader*)ws;
    float4* rec = (float4*)(ws + pl.off_rec);
    int* table = (int*)(ws + pl.off_table);
    double* partials = stats ? (double*)(ws + pl.off_partials) : nullptr;

    // the LDS-size attribute is per device: remember it per (device, kernel form)
    static std::atomic<unsigned> attr_done[64];
    int dev = 0;
    EVR_HIP(hipGetDevice(&dev));
    const unsigned bit = RAW ? 2u : 1u;
    if (dev < 0 || dev >= 64 || !(attr_done[dev].load(std::memory_order_relaxed) & bit)) {
        EVR_HIP(hipFuncSetAttribute((const void*)vox_split_kernel<RAW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EVR_HIP(hipFuncSetAttribute((const void*)vox_range_kernel<RAW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (RAW) {
            EVR_HIP(hipFuncSetAttribute((const void*)vox_range_kernel<RAW, RAW ? 1 : 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            EVR_HIP(hipFuncSetAttribute((const void*)vox_range_kernel<RAW, RAW ? 2 : 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        if (dev >= 0 && dev < 64) attr_done[dev].fetch_or(bit, std::memory_order_relaxed);
    }
    // split workgroups per window: enough for the average window in one pass, at most 32 (longer windows loop)
    int S = split_groups_env();
    if (S <= 0) {
        const int64_t avg = n_events_total / n_windows;
        S = (int)((avg + SEG - 1) / SEG);
        S = S < 1 ? 1 : (S > 32 ? 32 : S);
    }
    // floor(y / rows) = umulhi(y, magic) for y < 2^16 (rows == 1: magic would not fit 32 bits -> 0 = identity)
    const unsigned rows_magic = pl.rows == 1 ? 0u : (unsigned)((0x100000000ULL + pl.rows - 1) / pl.rows);
    const int64_t HW = (int64_t)H * W;
    const int vec_out = (HW % 4 == 0) && (((int64_t)pl.rows * W) % 4 == 0 || pl.G == 1) && (((uintptr_t)out & 15) == 0);
    // flush index -> (bin, group) by multiply-high: groups per bin of a full range; exact for every index < B * Rq <= 2^15
    const unsigned rq_full = (unsigned)(((int64_t)pl.rows * W) >> 2);
    const unsigned rq_magic = (vec_out && rq_full >= 2 && (int64_t)B * HW < (1LL << 31)) ? (unsigned)((0x100000000ULL + rq_full - 1) / rq_full) : 0u;
    const int xcd_map = n_windows >= 8;
    EVR_REQUIRE((int64_t)((n_windows + 7) / 8) * 8 * pl.G < (1LL << 31), "evr_voxelize: %d windows x %d ranges exceed the grid", n_windows, pl.G);
    // The split kernel is bound by latency (one load round trip, ballot ranks, a prefix: ~25 % of a call at 346x260 where a window has
    // many events per output byte), the range kernel by its store bursts -- they use different parts of a CU.  A VERY large call is cut
    // into chunks of windows: the split of chunk c + 1 runs on an internal second stream UNDER the range launch of chunk c (events
    // order split c -> range c; the chunks' records, table rows and output slices are disjoint).  Measured (round 5, 346x260, 15k
    // events per window, us per call for 1 / 2 / 4 / 8 chunks): 64 windows 66 / 88 / 93 / 140, 512 windows 279 / 289 / 291 / 333,
    // 2048 windows 1171 / 1081 / 1099 / 1090 -- the cross-stream hand-offs cost more than the overlap returns until a chunk is
    // ~1000 windows, so: two chunks from 2048 windows on, one below.  EVR_VOX_CHUNKS=<n> forces a count.
    static const int chunks_env = getenv("EVR_VOX_CHUNKS") ? atoi(getenv("EVR_VOX_CHUNKS")) : 0;
    int C = chunks_env > 0 ? chunks_env : (n_windows >= 2048 ? 2 : 1);
    if (C > MAX_CHUNKS) C = MAX_CHUNKS;
    if (C > n_windows / 8) C = n_windows / 8 > 0 ? n_windows / 8 : 1;
    SideStream* ss = nullptr;
    if (C > 1) {
        ss = side_stream(dev);
        if (!ss) C = 1;
    }
    auto launch_split = [&](hipStream_t st, int w0, int w1) {
        hipLaunchKernelGGL(vox_split_kernel<RAW>, dim3((unsigned)(w1 - w0) * S), dim3(K1T), pl.lds1, st, src, win_begin,
                           win_end, rec_base, rec, table, hdr, S, pl.G, pl.rows, rows_magic, B, H, W, w0);
    };
    auto launch_range = [&](int w0, int w1, int publish) {
        const int nw = w1 - w0;
        const int64_t blocks = xcd_map ? (int64_t)((nw + 7) / 8) * 8 * pl.G : (int64_t)nw * pl.G;
        static const int scan_probe = getenv("EVR_VOX_SCANPROBE") ? atoi(getenv("EVR_VOX_SCANPROBE")) : 0;
        if (RAW && scan_probe == 1)
            hipLaunchKernelGGL((vox_range_kernel<RAW, RAW ? 1 : 0>), dim3((unsigned)blocks), dim3(K2T), pl.lds2, stream, win_begin, win_end, rec_base,
                               rec, table, out, partials, hdr, w1, pl.G, pl.rows, pl.Rp, B, H, W, vec_out, xcd_map, rq_magic, w0, publish, src.xy);
        else if (RAW && scan_probe == 2)
            hipLaunchKernelGGL((vox_range_kernel<RAW, RAW ? 2 : 0>), dim3((unsigned)blocks), dim3(K2T), pl.lds2, stream, win_begin, win_end, rec_base,
                               rec, table, out, partials, hdr, w1, pl.G, pl.rows, pl.Rp, B, H, W, vec_out, xcd_map, rq_magic, w0, publish, src.xy);
        else
        hipLaunchKernelGGL(vox_range_kernel<RAW>, dim3((unsigned)blocks), dim3(K2T), pl.lds2, stream, win_begin, win_end, rec_base,
                           rec, table, out, partials, hdr, w1, pl.G, pl.rows, pl.Rp, B, H, W, vec_out, xcd_map, rq_magic, w0, publish);
    };
    if (C <= 1) {
        launch_split(stream, 0, n_windows);
        EVR_LAUNCH_CHECK();
        launch_range(0, n_windows, 1);
        EVR_LAUNCH_CHECK();
    } else {
        std::lock_guard<std::mutex> lock(ss->mu);      // the side stream and its events are per device, shared by every caller
        // the side stream joins behind the caller's earlier work (a previous call's range launch may still read this workspace)
        EVR_HIP(hipEventRecord(ss->fork, stream));
        EVR_HIP(hipStreamWaitEvent(ss->s, ss->fork, 0));
        const int per = ((n_windows + C - 1) / C + 7) / 8 * 8;      // chunk sizes in multiples of 8 windows (the XCD map of the range kernel)
        int c = 0;
        for (int w0 = 0; w0 < n_windows; w0 += per, ++c) {
            const int w1 = w0 + per < n_windows ? w0 + per : n_windows;
            launch_split(ss->s, w0, w1);
            EVR_LAUNCH_CHECK();
            EVR_HIP(hipEventRecord(ss->done[c], ss->s));
            EVR_HIP(hipStreamWaitEvent(stream, ss->done[c], 0));
            launch_range(w0, w1, w1 == n_windows ? 1 : 0);
            EVR_LAUNCH_CHECK();
        }
    }
    if (stats) {
        hipLaunchKernelGGL(vox_stats_kernel, dim3(n_windows), dim3(256), 0, stream, partials, stats, pl.G);
        EVR_LAUNCH_CHECK();
    }
    return EVR_OK;
}

}  // namespace

extern "C" size_t evr_voxelize_workspace_bytes(int64_t n_events_total, int n_windows, int B, int H, int W) {
    if (n_windows < 0 || H < 1 || W < 1 || B < 1 || n_events_total < 0) return 0;
    VoxPlan pl;
    if (!make_plan(n_events_total, n_windows, B, H, W, pl)) return 0;
    return pl.total;
}

extern "C" int evr_voxelize(const float* x, const float* y, const float* t, const float* p,
                            const int64_t* win_offsets, int n_windows, int64_t n_events_total, int B, int H,
                            int W, float* out, double* stats, void* workspace, size_t workspace_bytes,
                            evr_stream_t stream) {
    EVR_REQUIRE(n_events_total == 0 || (x && y && t && p), "evr_voxelize: null event arrays");
    EventSrc s{};
    s.x = x; s.y = y; s.t = t; s.p = p;
    return voxelize_impl<false>(s, win_offsets, win_offsets ? win_offsets + 1 : nullptr, win_offsets, n_windows,
                                n_events_total, B, H, W, out, stats, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int evr_voxelize_raw(const int16_t* xy, const double* ts, const uint8_t* pol,
                                const int64_t* win_offsets, int n_windows, int64_t n_events_total, int B,
                                int H, int W, float* out, double* stats, void* workspace,
                                size_t workspace_bytes, evr_stream_t stream) {
    EVR_REQUIRE(n_events_total == 0 || (xy && ts && pol), "evr_voxelize_raw: null event arrays");
    EventSrc s{};
    s.xy = xy; s.ts = ts; s.pol = pol;
    return voxelize_impl<true>(s, win_offsets, win_offsets ? win_offsets + 1 : nullptr, win_offsets, n_windows,
                               n_events_total, B, H, W, out, stats, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int evr_voxelize_raw_windows(const int16_t* xy, const double* ts, const uint8_t* pol,
                                        const int64_t* win_begin, const int64_t* win_end, const int64_t* rec_base,
                                        int n_windows, int64_t n_window_events, int B, int H, int W, float* out,
                                        double* stats, void* workspace, size_t workspace_bytes, evr_stream_t stream) {
    EVR_REQUIRE(n_window_events == 0 || (xy && ts && pol), "evr_voxelize_raw_windows: null event arrays");
    EventSrc s{};
    s.xy = xy; s.ts = ts; s.pol = pol;
    return voxelize_impl<true>(s, win_begin, win_end, rec_base, n_windows, n_window_events, B, H, W, out, stats,
                               workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int evr_voxelize_dropped(const void* workspace, int64_t* n_dropped_host, evr_stream_t stream) {
    EVR_REQUIRE(workspace && n_dropped_host, "evr_voxelize_dropped: null pointer");
    unsigned long long v = 0;
    EVR_HIP(hipMemcpyAsync(&v, (const char*)workspace + offsetof(VoxHeader, dropped_last), sizeof(v), hipMemcpyDeviceToHost,
                           (hipStream_t)stream));
    EVR_HIP(hipStreamSynchronize((hipStream_t)stream));
    *n_dropped_host = (int64_t)v;
    return EVR_OK;
}

extern "C" int evr_voxelize_dropped_total(const void* workspace, int64_t* n_dropped_host, evr_stream_t stream) {
    EVR_REQUIRE(workspace && n_dropped_host, "evr_voxelize_dropped_total: null pointer");
    unsigned long long v = 0;
    EVR_HIP(hipMemcpyAsync(&v, (const char*)workspace + offsetof(VoxHeader, dropped_total), sizeof(v), hipMemcpyDeviceToHost,
                           (hipStream_t)stream));
    EVR_HIP(hipStreamSynchronize((hipStream_t)stream));
    *n_dropped_host = (int64_t)v;
    return EVR_OK;
}

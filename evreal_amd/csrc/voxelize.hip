// Events -> voxel grid on gfx950, bit-identical to the reference's CPU path.
//
// Reference: utils/event_utils.py:27-59 (events_to_voxel_torch), :4-24 (events_to_image_torch,
// sequential index_put_(accumulate=True)); called from dataset.py:205-216.
//
// The reference adds, per (bin, pixel) cell, the fp32 weights of the events that hit it IN EVENT
// ORDER.  Float atomics would break that order, so every cell is accumulated in LDS by exactly one
// wave that sees the cell's events in time order.  ONE kernel does the whole job (round 1 used a
// bucket-sort kernel + a tile kernel with 16-B records in between):
//
//   vox_fused   The pixel plane is cut into G ranges of R consecutive flattened pixels; one 512-thread
//               workgroup owns one (window, range): its B x R cells live in LDS (45 KiB at B = 5,
//               R = 2304 -> two workgroups per CU).
//               1. SCAN: the workgroup reads the coordinates of ALL the window's events, 16 B (4 events)
//                  per lane per load, eight loads in flight per lane -- one memory round trip for 16k
//                  events.  The G workgroups of a window re-read the same 60 KB from the L2 of ONE XCD
//                  (blockIdx -> (window, range) keeps a window's ranges on one XCD), never from HBM.
//               2. COMPACT: events whose pixel falls in the range (~N/G) are written to an LDS list in
//                  event order (ballot ranks inside a wave, a prefix over the 8 waves' counts).
//               3. GATHER: one thread per list entry fetches that event's timestamp and polarity and
//                  computes t_norm and the weight sign with the reference's exact fp32 operation order.
//               4. ACCUMULATE: wave k owns the pixels with (pixel & 7) == k.  Every wave walks the list 64
//                  entries at a time; lanes that hit the same pixel inside a batch are serialised
//                  lowest-lane-first through an LDS atomic-min ticket, so every cell sees its adds in
//                  time order.  Only the (at most two) bins with a non-zero weight are touched: adding
//                  the reference's +-0 products never changes a cell (cells start at +0).
//               5. STREAM OUT: the range's cells go to HBM with 16-B stores -- every output cell is
//                  written exactly once, zero fill included, nothing is read back -- together with the
//                  range's {sum, sum of squares, nnz} in fp64 for eval.py:402-405.
//               A list holds 1024 events; a scan iteration whose range receives more (bursts, hot
//               pixels) is replayed in 16 sub-passes of <= 1024 events each -- slower, same result.
//   vox_stats   fixed-order reduction of the per-range partials -> stats[w][3] (deterministic).
//
// HBM traffic per window: the events once (13 N raw / 16 N fp32) + 4 B H W (output) -- the
// algorithmic bytes of SURVEY 8d.
//
// fp32 arithmetic is one IEEE rounding per op (this file is built with -ffp-contract=off; HIP's
// default correctly-rounded fp32 divide is kept), matching torch's CPU kernels.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int VT = 512;             // threads per workgroup
constexpr int VW = VT / 64;         // waves
constexpr int U = 8;                // 4-event loads in flight per lane
constexpr int EPW = U * 256;        // events per wave per scan iteration
constexpr int EPI = VW * EPW;       // events per workgroup per scan iteration (16384)
constexpr int CAP = 1024;           // list capacity (events of one pass that fall in the range)
constexpr int NTAG = 128;           // ticket slots per wave (hashed by pixel)
constexpr int ACC_KB_DEFAULT = 45;  // LDS for the B x R cells

struct VoxHeader {           // first 256 B of the workspace
    unsigned long long dropped;
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

// coordinates of 4 consecutive events
template <bool RAW> struct Quad;
template <> struct Quad<true> { uint4 q; };
template <> struct Quad<false> { float4 x, y; };

template <bool RAW>
__device__ __forceinline__ void load_quad(const EventSrc& s, int64_t i4, int64_t a, int64_t end, bool vec, Quad<RAW>& o) {
    if (RAW) {
        uint4& q = ((Quad<true>&)o).q;
        q = make_uint4(0, 0, 0, 0);
        if (i4 >= end) return;
        const uint32_t* w = (const uint32_t*)s.xy;         // one event = one 32-bit word (x | y << 16)
        if (vec && i4 + 4 <= end) {
            q = *(const uint4*)(w + i4);
        } else {
            if (i4 + 0 >= a && i4 + 0 < end) q.x = (uint32_t)(uint16_t)s.xy[2 * (i4 + 0)] | ((uint32_t)(uint16_t)s.xy[2 * (i4 + 0) + 1] << 16);
            if (i4 + 1 >= a && i4 + 1 < end) q.y = (uint32_t)(uint16_t)s.xy[2 * (i4 + 1)] | ((uint32_t)(uint16_t)s.xy[2 * (i4 + 1) + 1] << 16);
            if (i4 + 2 >= a && i4 + 2 < end) q.z = (uint32_t)(uint16_t)s.xy[2 * (i4 + 2)] | ((uint32_t)(uint16_t)s.xy[2 * (i4 + 2) + 1] << 16);
            if (i4 + 3 >= a && i4 + 3 < end) q.w = (uint32_t)(uint16_t)s.xy[2 * (i4 + 3)] | ((uint32_t)(uint16_t)s.xy[2 * (i4 + 3) + 1] << 16);
        }
    } else {
        Quad<false>& f = (Quad<false>&)o;
        f.x = make_float4(-1.f, -1.f, -1.f, -1.f); f.y = f.x;
        if (i4 >= end) return;
        if (vec && i4 + 4 <= end) {
            f.x = *(const float4*)(s.x + i4); f.y = *(const float4*)(s.y + i4);
        } else {
            if (i4 + 0 >= a && i4 + 0 < end) { f.x.x = s.x[i4 + 0]; f.y.x = s.y[i4 + 0]; }
            if (i4 + 1 >= a && i4 + 1 < end) { f.x.y = s.x[i4 + 1]; f.y.y = s.y[i4 + 1]; }
            if (i4 + 2 >= a && i4 + 2 < end) { f.x.z = s.x[i4 + 2]; f.y.z = s.y[i4 + 2]; }
            if (i4 + 3 >= a && i4 + 3 < end) { f.x.w = s.x[i4 + 3]; f.y.w = s.y[i4 + 3]; }
        }
    }
}

// flattened pixel of event j of the quad, or -1 when it lies outside the sensor
template <bool RAW>
__device__ __forceinline__ int quad_pixel(const Quad<RAW>& o, int j, int W, int H) {
    long long xi, yi;
    if (RAW) {
        const uint4& q = ((const Quad<true>&)o).q;
        const uint32_t w = j == 0 ? q.x : j == 1 ? q.y : j == 2 ? q.z : q.w;
        xi = (int16_t)(w & 0xFFFFu); yi = (int16_t)(w >> 16);
    } else {
        const Quad<false>& f = (const Quad<false>&)o;
        const float fx = j == 0 ? f.x.x : j == 1 ? f.x.y : j == 2 ? f.x.z : f.x.w;
        const float fy = j == 0 ? f.y.x : j == 1 ? f.y.y : j == 2 ? f.y.z : f.y.w;
        xi = (long long)fx; yi = (long long)fy;            // .long(): truncation toward zero
    }
    const bool valid = (xi >= 0) && (xi < W) && (yi >= 0) && (yi < H);
    return valid ? (int)(yi * W + xi) : -1;
}

__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool RAW>
__global__ __launch_bounds__(VT, 4) void vox_fused_kernel(
    EventSrc src, const int64_t* __restrict__ win_begin, const int64_t* __restrict__ win_end,
    float* __restrict__ out, double* __restrict__ partials, VoxHeader* hdr,
    int n_windows, int G, int R, int B, int H, int W, int vec_in, int vec_out, int xcd_map) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* acc = lds;                                          // [B][R]
    unsigned* list_e = (unsigned*)(acc + (size_t)B * R);       // [CAP] event index in the window, then t_norm bits
    float* list_p = (float*)(list_e + CAP);                    // [CAP] polarity weight
    unsigned* list_pix = (unsigned*)(list_p + CAP);            // [CAP] pixel - range start
    unsigned* tags = list_pix + CAP;                           // [VW][NTAG]
    int* wave_cnt = (int*)(tags + VW * NTAG);                  // [VW]
    double* red = (double*)(wave_cnt + VW);                    // [VW][3]   (8-B aligned: everything before is a multiple of 8 B)

    const int tid = threadIdx.x, lane = tid & 63, k = tid >> 6;
    int w, g;
    if (xcd_map) {      // blocks b, b+8, b+16, ... run on one XCD: give them the ranges of the same windows
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        w = (slot / G) * 8 + xcd; g = slot % G;
    } else {
        w = blockIdx.x / G; g = blockIdx.x % G;
    }
    if (w >= n_windows) return;
    const int64_t HW = (int64_t)H * W;
    const int pix_lo = g * R;
    const int pix_hi = (int)min((int64_t)pix_lo + R, HW);

    const int64_t a = win_begin[w];
    const int64_t ne = win_end[w] - a;
    const int n = ne > 0 ? (int)ne : 0;
    const int64_t end = a + n;
    const int64_t a_al = a & ~(int64_t)3;

    // first scan iteration's loads go out before anything else
    Quad<RAW> quad[U];
    const bool vin = vec_in != 0;
#pragma unroll
    for (int c = 0; c < U; ++c)
        load_quad<RAW>(src, a_al + (int64_t)k * EPW + c * 256 + lane * 4, a, end, vin, quad[c]);

    float t0 = 0.f, dt = 0.f;
    double t0d = 0.0;
    if (n > 0) {
        if (RAW) {
            t0d = src.ts[a];
            dt = (float)(src.ts[end - 1] - t0d) - 0.0f;
        } else {
            t0 = src.t[a];
            dt = src.t[end - 1] - t0;
        }
    }
    const bool lin = ((double)dt < 1e-9);
    const float bm1 = (float)(B - 1);

    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = tid * 4; j < B * R; j += VT * 4) *(float4*)&acc[j] = z4;
    volatile unsigned* tag = (volatile unsigned*)(tags + k * NTAG);
    tag[lane] = 0xFFFFFFFFu; tag[lane + 64] = 0xFFFFFFFFu;

    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned n_bad = 0;

    for (int64_t it0 = a_al; it0 < end; it0 += EPI) {
        if (it0 != a_al) {
#pragma unroll
            for (int c = 0; c < U; ++c)
                load_quad<RAW>(src, it0 + (int64_t)k * EPW + c * 256 + lane * 4, a, end, vin, quad[c]);
        }
        // which of this lane's 4 x U events fall in the range
        unsigned mask = 0;
#pragma unroll
        for (int c = 0; c < U; ++c) {
            const int64_t i4 = it0 + (int64_t)k * EPW + c * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t i = i4 + j;
                if (i >= a && i < end) {
                    const int pix = quad_pixel<RAW>(quad[c], j, W, H);
                    if (pix < 0) n_bad += (g == 0);
                    else if (pix >= pix_lo && pix < pix_hi) mask |= 1u << (4 * c + j);
                }
            }
        }
        int wcnt = wave_sum_int(__popc(mask));
        if (lane == 0) wave_cnt[k] = wcnt;
        __syncthreads();
        int tot_it = 0;
#pragma unroll
        for (int q = 0; q < VW; ++q) tot_it += wave_cnt[q];
        const int npass = (tot_it <= CAP) ? 1 : 2 * U;

        for (int pass = 0; pass < npass; ++pass) {
            unsigned sel = mask;
            if (npass > 1) {        // sub-pass = one 4-event load slot of one half of the waves: <= 4 x 256 events
                const int pc = pass >> 1, ph = pass & 1;
                sel = ((k >> 2) == ph) ? (mask & (0xFu << (4 * pc))) : 0u;
                wcnt = wave_sum_int(__popc(sel));
                __syncthreads();                       // the previous sub-pass is done with the list and the counts
                if (lane == 0) wave_cnt[k] = wcnt;
                __syncthreads();
            }
            int base = 0, tot = 0;
#pragma unroll
            for (int q = 0; q < VW; ++q) {
                const int v = wave_cnt[q];
                base += (q < k) ? v : 0;
                tot += v;
            }
            if (tot == 0) continue;

            // ---- compact: list entries in event order ----
            int run = base;
#pragma unroll
            for (int c = 0; c < U; ++c) {
                const unsigned bits = (sel >> (4 * c)) & 0xFu;
                if (!__ballot(bits != 0)) continue;
                const unsigned long long b0 = __ballot(bits & 1u), b1 = __ballot(bits & 2u), b2 = __ballot(bits & 4u),
                                         b3 = __ballot(bits & 8u);
                const int pre = __popcll(b0 & lt_mask) + __popcll(b1 & lt_mask) + __popcll(b2 & lt_mask) + __popcll(b3 & lt_mask);
                const int64_t i4 = it0 + (int64_t)k * EPW + c * 256 + lane * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (bits & (1u << j)) {
                        const int pos = run + pre + __popc(bits & ((1u << j) - 1u));
                        list_e[pos] = (unsigned)(i4 + j - a);
                        list_pix[pos] = (unsigned)(quad_pixel<RAW>(quad[c], j, W, H) - pix_lo);
                    }
                }
                run += __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
            }
            __syncthreads();

            // ---- gather: timestamp and polarity of the listed events -> t_norm, weight ----
            for (int i = tid; i < tot; i += VT) {
                const int e = (int)list_e[i];
                float tn, pv;
                if (RAW) {
                    const float tf = (float)(src.ts[a + e] - t0d);       // dataset.py:56 (f64 subtract, cast)
                    if (lin) tn = lin_tnorm(e, n, B);
                    else { float q = (tf - 0.0f) / dt; tn = q * bm1; }   // ts[0] is exactly 0 after the shift
                    pv = (float)((double)src.pol[a + e] * 2.0 - 1.0);    // dataset.py:227
                } else {
                    if (lin) tn = lin_tnorm(e, n, B);
                    else { float d = src.t[a + e] - t0; float q = d / dt; tn = q * bm1; }
                    pv = src.p[a + e];
                }
                list_e[i] = __float_as_uint(tn);
                list_p[i] = pv;
            }
            __syncthreads();

            // ---- accumulate: wave k owns the pixels with (pixel & 7) == k ----
            for (int b0i = 0; b0i < tot; b0i += 64) {
                const int i = b0i + lane;
                const unsigned pl = (i < tot) ? list_pix[i] : 0xFFFFFFFFu;
                bool pending = (i < tot) && ((int)(pl & 7u) == k);
                if (!__ballot(pending)) continue;
                float tn = 0.f, p = 0.f;
                if (pending) { tn = __uint_as_float(list_e[i]); p = list_p[i]; }
                const unsigned h = (pl >> 3) & (NTAG - 1);
                while (__ballot(pending)) {
                    if (pending) atomicMin((unsigned*)&tag[h], (unsigned)lane);
                    __builtin_amdgcn_wave_barrier();
                    if (pending && tag[h] == (unsigned)lane) {
                        // event_utils.py:53-56, one rounding per op; bins with weight 0 add +-0: no effect, skipped
                        float bf = floorf(tn);
                        bf = fminf(fmaxf(bf, -2.0f), (float)B);
                        const int bb = (int)bf;
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const int b = bb + s;
                            if (b >= 0 && b < B) {
                                const float d = tn - (float)b;
                                const float wgt = 1.0f - fabsf(d);
                                if (wgt > 0.0f) {
                                    const float v = p * wgt;
                                    volatile float* cell = &acc[b * R + pl];
                                    const float sum = *cell + v;
                                    *cell = sum;
                                }
                            }
                        }
                        tag[h] = 0xFFFFFFFFu;
                        pending = false;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        __syncthreads();     // list, counts (and, after the last iteration, the cells) are settled
    }
    if (n_bad) atomicAdd(&hdr->dropped, (unsigned long long)n_bad);

    // ---- stream the range out (each output cell written once) + its statistics ----
    double s1 = 0.0, s2 = 0.0, nz = 0.0;
    const int Rc = pix_hi - pix_lo;
    float* o = out + (int64_t)w * B * HW + pix_lo;
    for (int b = 0; b < B; ++b) {
        for (int j = tid * 4; j < Rc; j += VT * 4) {
            const float4 v = *(const float4*)&acc[b * R + j];
            if (vec_out && j + 4 <= Rc) {
                *(float4*)&o[(int64_t)b * HW + j] = v;
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
                for (int q = 0; q < 4; ++q)
                    if (j + q < Rc) o[(int64_t)b * HW + j + q] = vv[q];
            }
            // cells beyond the range end stay zero in LDS, so they do not disturb the statistics
            if (partials && (v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f)) {
                s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
                s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
                nz += (double)((v.x != 0.f) + (v.y != 0.f) + (v.z != 0.f) + (v.w != 0.f));
            }
        }
    }
    if (partials) {
        s1 = evr_wave_sum(s1); s2 = evr_wave_sum(s2); nz = evr_wave_sum(nz);
        if (lane == 0) { red[k * 3 + 0] = s1; red[k * 3 + 1] = s2; red[k * 3 + 2] = nz; }
        __syncthreads();
        if (tid < 3) {
            double t = 0.0;
            for (int q = 0; q < VW; ++q) t += red[q * 3 + tid];      // fixed order: deterministic
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

struct VoxPlan {
    int R, G;
    size_t lds, off_partials, total;
};

int acc_kb() {
    const char* e = getenv("EVR_VOX_ACC_KB");      // tuning knob: LDS for the cells -> range size -> workgroups per CU
    const int v = e ? atoi(e) : ACC_KB_DEFAULT;
    return v < 8 ? 8 : (v > 140 ? 140 : v);
}

bool make_plan(int n_windows, int B, int H, int W, VoxPlan& p) {
    const int64_t HW = (int64_t)H * W;
    int64_t rmax = ((int64_t)acc_kb() * 1024 / (4LL * B)) & ~63LL;
    if (rmax < 64) return false;
    const int64_t G = (HW + rmax - 1) / rmax;
    int64_t R = (HW + G - 1) / G;
    R = (R + 63) & ~63LL;
    p.R = (int)R; p.G = (int)G;
    p.lds = (size_t)B * R * 4 + (size_t)CAP * 12 + (size_t)VW * NTAG * 4 + VW * 4 + VW * 3 * 8;
    size_t off = 256;
    p.off_partials = off; off += evr::align_up((size_t)n_windows * p.G * 3 * sizeof(double), 256);
    p.total = off;
    return true;
}

template <bool RAW>
int voxelize_impl(const EventSrc& src, const int64_t* win_begin, const int64_t* win_end,
                  int n_windows, int64_t n_events_total,
                  int B, int H, int W, float* out, double* stats, void* workspace, size_t ws_bytes,
                  hipStream_t stream) {
    EVR_REQUIRE(n_windows >= 0 && B >= 1 && H >= 1 && W >= 1, "evr_voxelize: bad shape n_windows=%d B=%d H=%d W=%d", n_windows, B, H, W);
    EVR_REQUIRE((int64_t)H * W < (1LL << 30), "evr_voxelize: sensor %dx%d too large", W, H);
    EVR_REQUIRE(n_events_total >= 0 && n_events_total < (1LL << 31), "evr_voxelize: n_events_total out of range");
    if (n_windows == 0) return EVR_OK;
    EVR_REQUIRE(win_begin && win_end && out && workspace, "evr_voxelize: null pointer");
    VoxPlan pl;
    EVR_REQUIRE(make_plan(n_windows, B, H, W, pl), "evr_voxelize: B=%d bins do not fit the LDS cell budget", B);
    if (ws_bytes < pl.total) {
        evr::set_error("evr_voxelize: workspace %zu B < required %zu B", ws_bytes, pl.total);
        return EVR_ERR_WORKSPACE;
    }
    EVR_REQUIRE(pl.lds <= 160 * 1024, "evr_voxelize: B=%d needs %zu B of LDS (> 160 KiB)", B, pl.lds);
    char* ws = (char*)workspace;
    VoxHeader* hdr = (VoxHeader*)ws;
    double* partials = stats ? (double*)(ws + pl.off_partials) : nullptr;

    EVR_HIP(hipMemsetAsync(hdr, 0, sizeof(VoxHeader), stream));
    // the attribute is per device: set it on every call (cheap) rather than caching it process-wide
    EVR_HIP(hipFuncSetAttribute((const void*)vox_fused_kernel<RAW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int64_t HW = (int64_t)H * W;
    const int vec_out = (HW % 4 == 0) && (((uintptr_t)out & 15) == 0);
    const int vec_in = RAW ? (((uintptr_t)src.xy & 15) == 0) : ((((uintptr_t)src.x | (uintptr_t)src.y) & 15) == 0);
    const int xcd_map = n_windows >= 8;
    const int64_t blocks = xcd_map ? (int64_t)((n_windows + 7) / 8) * 8 * pl.G : (int64_t)n_windows * pl.G;
    EVR_REQUIRE(blocks < (1LL << 31), "evr_voxelize: %d windows x %d ranges exceed the grid", n_windows, pl.G);
    hipLaunchKernelGGL(vox_fused_kernel<RAW>, dim3((unsigned)blocks), dim3(VT), pl.lds, stream, src, win_begin, win_end,
                       out, partials, hdr, n_windows, pl.G, pl.R, B, H, W, vec_in, vec_out, xcd_map);
    EVR_LAUNCH_CHECK();
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
    if (!make_plan(n_windows, B, H, W, pl)) return 0;
    return pl.total;
}

extern "C" int evr_voxelize(const float* x, const float* y, const float* t, const float* p,
                            const int64_t* win_offsets, int n_windows, int64_t n_events_total, int B, int H,
                            int W, float* out, double* stats, void* workspace, size_t workspace_bytes,
                            evr_stream_t stream) {
    EVR_REQUIRE(n_events_total == 0 || (x && y && t && p), "evr_voxelize: null event arrays");
    EventSrc s{};
    s.x = x; s.y = y; s.t = t; s.p = p;
    return voxelize_impl<false>(s, win_offsets, win_offsets ? win_offsets + 1 : nullptr, n_windows,
                                n_events_total, B, H, W, out, stats, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int evr_voxelize_raw(const int16_t* xy, const double* ts, const uint8_t* pol,
                                const int64_t* win_offsets, int n_windows, int64_t n_events_total, int B,
                                int H, int W, float* out, double* stats, void* workspace,
                                size_t workspace_bytes, evr_stream_t stream) {
    EVR_REQUIRE(n_events_total == 0 || (xy && ts && pol), "evr_voxelize_raw: null event arrays");
    EventSrc s{};
    s.xy = xy; s.ts = ts; s.pol = pol;
    return voxelize_impl<true>(s, win_offsets, win_offsets ? win_offsets + 1 : nullptr, n_windows,
                               n_events_total, B, H, W, out, stats, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int evr_voxelize_raw_windows(const int16_t* xy, const double* ts, const uint8_t* pol,
                                        const int64_t* win_begin, const int64_t* win_end, const int64_t* rec_base,
                                        int n_windows, int64_t n_window_events, int B, int H, int W, float* out,
                                        double* stats, void* workspace, size_t workspace_bytes, evr_stream_t stream) {
    EVR_REQUIRE(n_window_events == 0 || (xy && ts && pol), "evr_voxelize_raw_windows: null event arrays");
    (void)rec_base;      // round 1's record layout; the fused kernel keeps no per-event records in the workspace
    EventSrc s{};
    s.xy = xy; s.ts = ts; s.pol = pol;
    return voxelize_impl<true>(s, win_begin, win_end, n_windows, n_window_events, B, H, W, out, stats,
                               workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int evr_voxelize_dropped(const void* workspace, int64_t* n_dropped_host, evr_stream_t stream) {
    EVR_REQUIRE(workspace && n_dropped_host, "evr_voxelize_dropped: null pointer");
    unsigned long long v = 0;
    EVR_HIP(hipMemcpyAsync(&v, workspace, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
    EVR_HIP(hipStreamSynchronize((hipStream_t)stream));
    *n_dropped_host = (int64_t)v;
    return EVR_OK;
}

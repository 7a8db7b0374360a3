// Events -> voxel grid on gfx950, bit-identical to the reference's CPU path.
//
// Reference: utils/event_utils.py:27-59 (events_to_voxel_torch), :4-24 (events_to_image_torch,
// sequential index_put_(accumulate=True)); called from dataset.py:205-216.
//
// The reference adds, per (bin, pixel) cell, the fp32 weights of the events that hit it IN EVENT
// ORDER.  Float atomics would break that order, so the work is organised as a stable bucket sort
// followed by in-order accumulation in LDS:
//
//   K1 vox_bucket   one 1024-thread workgroup per window.  The pixel plane is cut into tiles of T
//                   consecutive flattened pixels.  A stable multi-split by tile id (per-wave LDS
//                   histograms, scan, ballot-ranked scatter) writes one 16-B record per event
//                   {pixel-in-tile, t_norm, p} into the workspace, time order preserved per tile.
//   K2 vox_tile     one wave per (window, tile).  The tile's B x T cells live in LDS.  Events are
//                   taken 64 at a time; lanes hitting the same pixel inside a batch are serialised
//                   lowest-lane-first with an LDS atomic-min ticket, so every cell sees its adds in
//                   time order.  The finished tile is streamed out with 16-B stores: every output
//                   cell is written exactly once, zero fill included, and nothing is ever read
//                   back from HBM.  Per-tile {sum, sumsq, nnz} partials feed eval.py:402-405.
//   K3 vox_stats    fixed-order reduction of the partials -> stats[w][3] (deterministic).
//
// HBM traffic per window: 16 N (events) + 2 x 16 N (records, normally L2/MALL resident) +
// 4 B H W (output).  Algorithmic bytes: 16 N + 4 B H W.
//
// fp32 arithmetic is one IEEE rounding per op (this file is built with -ffp-contract=off; HIP's
// default correctly-rounded fp32 divide is kept), matching torch's CPU kernels.
#include "common.h"

namespace {

constexpr int K1_THREADS = 1024;
constexpr int K1_WAVES = K1_THREADS / 64;
constexpr int K2_WAVES = 4;
constexpr int MAX_TILES = 1024;

struct VoxHeader {           // first 256 B of the workspace
    unsigned long long dropped;
};

__host__ __device__ inline int tile_pixels(int64_t HW) {
    // smallest multiple of 256 with ceil(HW/T) <= MAX_TILES
    int64_t m = (HW + 256LL * MAX_TILES - 1) / (256LL * MAX_TILES);
    if (m < 1) m = 1;
    return (int)(256 * m);
}

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

template <bool RAW>
__device__ __forceinline__ long long event_pixel(const EventSrc& s, int64_t g, int W, int H, bool& valid) {
    long long xi, yi;
    if (RAW) {
        xi = s.xy[2 * g]; yi = s.xy[2 * g + 1];
    } else {
        xi = (long long)s.x[g]; yi = (long long)s.y[g];   // .long(): truncation toward zero
    }
    valid = (xi >= 0) && (xi < W) && (yi >= 0) && (yi < H);
    return yi * W + xi;
}

template <bool RAW>
__global__ __launch_bounds__(K1_THREADS) void vox_bucket_kernel(
    EventSrc src, const int64_t* __restrict__ win_begin, const int64_t* __restrict__ win_end,
    const int64_t* __restrict__ rec_base, float4* __restrict__ rec,
    int* __restrict__ tile_offsets, VoxHeader* hdr, int n_tiles, int T, int B, int H, int W) {
    extern __shared__ int smem[];
    int* cnt = smem;                           // [K1_WAVES][n_tiles]
    int* tstart = smem + K1_WAVES * n_tiles;   // [n_tiles]
    int* wsum = tstart + n_tiles;              // [K1_WAVES]

    const int w = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t a = win_begin[w];
    const int64_t ne = win_end[w] - a;
    const int n = ne > 0 ? (int)ne : 0;
    float4* wrec = rec + rec_base[w];   // this window's records (windows may overlap in the event stream)

    for (int i = tid; i < K1_WAVES * n_tiles; i += K1_THREADS) cnt[i] = 0;
    __syncthreads();

    const int chunk = ((n + K1_THREADS - 1) / K1_THREADS) * 64;   // per wave, multiple of 64
    const int beg = wave * chunk;
    const int end = min(n, beg + chunk);

    // phase a: per-wave tile histograms
    unsigned dropped = 0;
    for (int i = beg + lane; i < end; i += 64) {
        bool valid;
        long long pix = event_pixel<RAW>(src, a + i, W, H, valid);
        if (valid) atomicAdd(&cnt[wave * n_tiles + (int)(pix / T)], 1);
        else ++dropped;
    }
    if (dropped) atomicAdd(&hdr->dropped, (unsigned long long)dropped);
    __syncthreads();

    // phase b: per-tile exclusive prefix over waves, then exclusive scan over tiles
    int total = 0;
    if (tid < n_tiles) {
        for (int wv = 0; wv < K1_WAVES; ++wv) {
            int c = cnt[wv * n_tiles + tid];
            cnt[wv * n_tiles + tid] = total;
            total += c;
        }
    }
    int incl = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wave_base = 0;
    for (int wv = 0; wv < wave; ++wv) wave_base += wsum[wv];
    const int excl = wave_base + incl - total;
    if (tid < n_tiles) {
        tstart[tid] = excl;
        tile_offsets[(int64_t)w * (n_tiles + 1) + tid] = excl;
    }
    if (tid == n_tiles - 1) tile_offsets[(int64_t)w * (n_tiles + 1) + n_tiles] = excl + total;
    __syncthreads();

    // phase c: stable scatter (time order kept inside every tile)
    int nbits = 0;
    while ((1 << nbits) < n_tiles) ++nbits;
    float t0 = 0.f, dt = 0.f;
    double t0d = 0.0;
    if (n > 0) {
        if (RAW) {
            t0d = src.ts[a];
            dt = (float)(src.ts[a + n - 1] - t0d) - 0.0f;
        } else {
            t0 = src.t[a];
            dt = src.t[a + n - 1] - t0;
        }
    }
    const bool lin = ((double)dt < 1e-9);
    const float bm1 = (float)(B - 1);
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    for (int batch = beg; batch < end; batch += 64) {
        const int i = batch + lane;
        bool valid = false;
        long long pix = 0;
        if (i < end) pix = event_pixel<RAW>(src, a + i, W, H, valid);
        const int tile = valid ? (int)(pix / T) : 0;
        unsigned long long peers = __ballot(valid);
        for (int bit = 0; bit < nbits; ++bit) {
            const bool s = (tile >> bit) & 1;
            const unsigned long long m = __ballot(s);
            peers &= s ? m : ~m;
        }
        if (valid) {
            const int rank = __popcll(peers & lt_mask);
            const int slot = wave * n_tiles + tile;
            const int dest = tstart[tile] + cnt[slot] + rank;
            float tn, pv;
            if (RAW) {
                const float tf = (float)(src.ts[a + i] - t0d);   // dataset.py:56 (f64 subtract, cast)
                if (lin) tn = lin_tnorm(i, n, B);
                else { float q = (tf - 0.0f) / dt; tn = q * bm1; }  // ts[0] is exactly 0 after the shift
                pv = (float)((double)src.pol[a + i] * 2.0 - 1.0);    // dataset.py:227
            } else {
                if (lin) tn = lin_tnorm(i, n, B);
                else { float d = src.t[a + i] - t0; float q = d / dt; tn = q * bm1; }
                pv = src.p[a + i];
            }
            wrec[dest] = make_float4(__int_as_float((int)(pix - (long long)tile * T)), tn, pv, 0.f);
            const bool last = (peers >> lane) == 1ull;   // no higher peer lane
            if (last) cnt[slot] += rank + 1;
        }
    }
}

__global__ __launch_bounds__(64 * K2_WAVES) void vox_tile_kernel(
    const float4* __restrict__ rec, const int* __restrict__ tile_offsets,
    const int64_t* __restrict__ rec_base, float* __restrict__ out, double* __restrict__ partials,
    int n_windows, int n_tiles, int T, int B, int64_t HW, int vec_ok) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * K2_WAVES + wave;
    if (gw >= (int64_t)n_windows * n_tiles) return;   // no block-level sync below
    const int w = (int)(gw / n_tiles), tile = (int)(gw % n_tiles);

    float* acc = lds + (size_t)wave * (B + 1) * T;                 // [B][T]
    volatile unsigned* tag = (volatile unsigned*)(acc + (size_t)B * T);   // [T]

    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = lane * 4; j < B * T; j += 256) *(float4*)&acc[j] = z4;
    for (int j = lane; j < T; j += 64) tag[j] = 0xFFFFFFFFu;
    __builtin_amdgcn_wave_barrier();

    const int* to = tile_offsets + (int64_t)w * (n_tiles + 1);
    const int e0 = to[tile], e1 = to[tile + 1];
    const float4* r = rec + rec_base[w];

    for (int batch = e0; batch < e1; batch += 64) {
        const int i = batch + lane;
        bool pending = i < e1;
        unsigned pix = 0; float tn = 0.f, p = 0.f;
        if (pending) {
            const float4 e = r[i];
            pix = (unsigned)__float_as_int(e.x); tn = e.y; p = e.z;
        }
        while (__ballot(pending)) {
            if (pending) atomicMin((unsigned*)&tag[pix], (unsigned)lane);
            __builtin_amdgcn_wave_barrier();
            if (pending && tag[pix] == (unsigned)lane) {
                // event_utils.py:53-56, one rounding per op
                for (int b = 0; b < B; ++b) {
                    const float d = tn - (float)b;
                    float wgt = 1.0f - fabsf(d);
                    wgt = (wgt > 0.0f) ? wgt : 0.0f;
                    const float v = p * wgt;
                    volatile float* cell = &acc[b * T + pix];
                    const float s = *cell + v;
                    *cell = s;
                }
                tag[pix] = 0xFFFFFFFFu;
                pending = false;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_wave_barrier();

    // stream the tile out (each output cell written once) + per-tile statistics
    double s1 = 0.0, s2 = 0.0, nz = 0.0;
    const int64_t pix0 = (int64_t)tile * T;
    float* o = out + (int64_t)w * B * HW;
    for (int b = 0; b < B; ++b) {
        for (int j = lane * 4; j < T; j += 256) {
            const float4 v = *(const float4*)&acc[b * T + j];
            const int64_t g = pix0 + j;
            if (g + 3 < HW && vec_ok) {
                *(float4*)&o[(int64_t)b * HW + g] = v;
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
                for (int k = 0; k < 4; ++k)
                    if (g + k < HW) o[(int64_t)b * HW + g + k] = vv[k];
            }
            // cells beyond HW stay zero in LDS, so they do not disturb the statistics
            s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
            nz += (double)((v.x != 0.f) + (v.y != 0.f) + (v.z != 0.f) + (v.w != 0.f));
        }
    }
    if (partials) {
        s1 = evr_wave_sum(s1); s2 = evr_wave_sum(s2); nz = evr_wave_sum(nz);
        if (lane == 0) {
            double* pp = partials + gw * 3;
            pp[0] = s1; pp[1] = s2; pp[2] = nz;
        }
    }
}

__global__ __launch_bounds__(256) void vox_stats_kernel(const double* __restrict__ partials,
                                                         double* __restrict__ stats, int n_tiles) {
    __shared__ double sh[3][4];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int t = tid; t < n_tiles; t += 256) {
        const double* p = partials + ((int64_t)w * n_tiles + t) * 3;
        a0 += p[0]; a1 += p[1]; a2 += p[2];
    }
    a0 = evr_wave_sum(a0); a1 = evr_wave_sum(a1); a2 = evr_wave_sum(a2);
    if (lane == 0) { sh[0][wave] = a0; sh[1][wave] = a1; sh[2][wave] = a2; }
    __syncthreads();
    if (tid < 3) stats[w * 3 + tid] = ((sh[tid][0] + sh[tid][1]) + sh[tid][2]) + sh[tid][3];
}

struct VoxPlan {
    int T, n_tiles;
    size_t off_rec, off_tiles, off_partials, total;
};

VoxPlan make_plan(int64_t n_events_total, int n_windows, int H, int W) {
    VoxPlan p;
    const int64_t HW = (int64_t)H * W;
    p.T = tile_pixels(HW);
    p.n_tiles = (int)((HW + p.T - 1) / p.T);
    size_t off = 256;
    p.off_rec = off; off += evr::align_up((size_t)(n_events_total > 0 ? n_events_total : 1) * sizeof(float4), 256);
    p.off_tiles = off; off += evr::align_up((size_t)n_windows * (p.n_tiles + 1) * sizeof(int), 256);
    p.off_partials = off; off += evr::align_up((size_t)n_windows * p.n_tiles * 3 * sizeof(double), 256);
    p.total = off;
    return p;
}

template <bool RAW>
int voxelize_impl(const EventSrc& src, const int64_t* win_begin, const int64_t* win_end, const int64_t* rec_base,
                  int n_windows, int64_t n_events_total,
                  int B, int H, int W, float* out, double* stats, void* workspace, size_t ws_bytes,
                  hipStream_t stream) {
    EVR_REQUIRE(n_windows >= 0 && B >= 1 && H >= 1 && W >= 1, "evr_voxelize: bad shape n_windows=%d B=%d H=%d W=%d", n_windows, B, H, W);
    EVR_REQUIRE((int64_t)H * W <= 256LL * MAX_TILES * 64, "evr_voxelize: sensor %dx%d too large", W, H);
    EVR_REQUIRE(n_events_total >= 0 && n_events_total < (1LL << 31), "evr_voxelize: n_events_total out of range");
    if (n_windows == 0) return EVR_OK;
    EVR_REQUIRE(win_begin && win_end && rec_base && out && workspace, "evr_voxelize: null pointer");
    const VoxPlan pl = make_plan(n_events_total, n_windows, H, W);
    if (ws_bytes < pl.total) {
        evr::set_error("evr_voxelize: workspace %zu B < required %zu B", ws_bytes, pl.total);
        return EVR_ERR_WORKSPACE;
    }
    const size_t lds2 = (size_t)K2_WAVES * (B + 1) * pl.T * sizeof(float);
    EVR_REQUIRE(lds2 <= 160 * 1024, "evr_voxelize: B=%d with tile %d needs %zu B of LDS (> 160 KiB)", B, pl.T, lds2);
    char* ws = (char*)workspace;
    VoxHeader* hdr = (VoxHeader*)ws;
    float4* rec = (float4*)(ws + pl.off_rec);
    int* tile_offsets = (int*)(ws + pl.off_tiles);
    double* partials = stats ? (double*)(ws + pl.off_partials) : nullptr;

    EVR_HIP(hipMemsetAsync(hdr, 0, sizeof(VoxHeader), stream));
    const size_t lds1 = (size_t)(K1_WAVES * pl.n_tiles + pl.n_tiles + K1_WAVES) * sizeof(int);
    static bool attr_done[2] = {false, false};
    if (!attr_done[RAW]) {
        EVR_HIP(hipFuncSetAttribute((const void*)vox_bucket_kernel<RAW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EVR_HIP(hipFuncSetAttribute((const void*)vox_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[RAW] = true;
    }
    hipLaunchKernelGGL(vox_bucket_kernel<RAW>, dim3(n_windows), dim3(K1_THREADS), lds1, stream, src, win_begin, win_end,
                       rec_base, rec, tile_offsets, hdr, pl.n_tiles, pl.T, B, H, W);
    EVR_LAUNCH_CHECK();
    const int64_t n_waves = (int64_t)n_windows * pl.n_tiles;
    const int64_t HW = (int64_t)H * W;
    const int vec_ok = (HW % 4 == 0) && (((uintptr_t)out & 15) == 0);
    hipLaunchKernelGGL(vox_tile_kernel, dim3((unsigned)((n_waves + K2_WAVES - 1) / K2_WAVES)), dim3(64 * K2_WAVES),
                       lds2, stream, rec, tile_offsets, rec_base, out, partials, n_windows, pl.n_tiles, pl.T, B,
                       HW, vec_ok);
    EVR_LAUNCH_CHECK();
    if (stats) {
        hipLaunchKernelGGL(vox_stats_kernel, dim3(n_windows), dim3(256), 0, stream, partials, stats, pl.n_tiles);
        EVR_LAUNCH_CHECK();
    }
    return EVR_OK;
}

}  // namespace

extern "C" size_t evr_voxelize_workspace_bytes(int64_t n_events_total, int n_windows, int B, int H, int W) {
    (void)B;
    if (n_windows < 0 || H < 1 || W < 1 || n_events_total < 0) return 0;
    return make_plan(n_events_total, n_windows, H, W).total;
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
    EVR_HIP(hipMemcpyAsync(&v, workspace, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
    EVR_HIP(hipStreamSynchronize((hipStream_t)stream));
    *n_dropped_host = (int64_t)v;
    return EVR_OK;
}

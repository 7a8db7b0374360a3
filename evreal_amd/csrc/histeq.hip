// Histogram-equalisation modes of the evaluation tracker (utils/eval_metrics.py:326-350 of the reference), applied to the
// clipped reconstruction and reference frame before the metrics:
//
//   'global'  skimage.exposure.equalize_hist(img)                      256-bin histogram over [min, max], CDF, np.interp
//   'local'   skimage.filters.rank.equalize(img_as_ubyte(img), disk(55))  per-pixel rank inside a radius-55 disk
//   'clahe'   cv2.createCLAHE(2.0, (8, 8)).apply(img_as_ubyte(img))     clipped tile histograms, bilinear blend of the tile LUTs
//
// PARITY UNPINNED: scikit-image and OpenCV are neither in the reference tree nor installed here; the kernels restate the
// published algorithms (skimage exposure.py / rank generic_cy.pyx, OpenCV imgproc/src/clahe.cpp) and are checked against
// the numpy restatement in oracle/histeq.py.  Every shipped eval config uses 'none'.
// One rounding per fp32 operation (built with -ffp-contract=off) so the binning follows numpy's float32 arithmetic.
#include "common.h"

namespace {

constexpr int HE_T = 1024;

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// np.linspace(mn, mx, 257, dtype=float32)[i]: arange(i) * step + start, last element = stop
__device__ __forceinline__ float he_edge(int i, float mn, float mx, float step) {
    if (i >= 256) return mx;
    const float m = (float)i * step;
    return m + mn;
}

// ---------------------------------------------------------------- 'global'
// One workgroup per image.  In place on img [n, HW] (values already clipped to [0, 1]).
__global__ __launch_bounds__(HE_T) void he_global_kernel(float* __restrict__ img, int HW) {
    __shared__ float red[2][HE_T / 64];
    __shared__ int hist[256];
    __shared__ float cdf[256], centers[256];
    __shared__ float s_mn, s_mx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* v = img + (int64_t)blockIdx.x * HW;

    float mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < HW; i += HE_T) { const float a = v[i]; mn = fminf(mn, a); mx = fmaxf(mx, a); }
    mn = wave_min_f(mn); mx = wave_max_f(mx);
    if (lane == 0) { red[0][wave] = mn; red[1][wave] = mx; }
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    if (tid == 0) {
        float a = red[0][0], b = red[1][0];
        for (int q = 1; q < HE_T / 64; ++q) { a = fminf(a, red[0][q]); b = fmaxf(b, red[1][q]); }
        if (a == b) { a = a - 0.5f; b = b + 0.5f; }      // np.histogram widens a degenerate range
        s_mn = a; s_mx = b;
    }
    __syncthreads();
    mn = s_mn; mx = s_mx;
    const float denom = mx - mn;
    const float step = denom / 256.0f;
    // np.histogram, uniform bins: index from the scaled value, then corrected against the float32 edges
    for (int i = tid; i < HW; i += HE_T) {
        const float a = v[i];
        const float f = ((a - mn) / denom) * 256.0f;
        int idx = (int)f;
        if (idx >= 256) idx = 255;
        if (idx < 0) idx = 0;
        if (a < he_edge(idx, mn, mx, step)) --idx;
        else if (idx != 255 && a >= he_edge(idx + 1, mn, mx, step)) ++idx;
        idx = min(max(idx, 0), 255);
        atomicAdd(&hist[idx], 1);
    }
    __syncthreads();
    if (tid < 256) {
        long long c = 0;
        for (int q = 0; q <= tid; ++q) c += hist[q];     // hist.cumsum()
        cdf[tid] = (float)((double)c / (double)HW);      // cdf / float(cdf[-1]) in float64, stored float32
        centers[tid] = (he_edge(tid, mn, mx, step) + he_edge(tid + 1, mn, mx, step)) / 2.0f;
    }
    __syncthreads();
    // np.interp(x, centers, cdf) in float64
    for (int i = tid; i < HW; i += HE_T) {
        const double x = (double)v[i];
        double r;
        if (x > (double)centers[255]) r = (double)cdf[255];
        else if (x < (double)centers[0]) r = (double)cdf[0];
        else {
            int lo = 0, hi = 255;                        // largest j with centers[j] <= x
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((double)centers[mid] <= x) lo = mid; else hi = mid - 1; }
            const int j = lo;
            if (j == 255 || (double)centers[j] == x) r = (double)cdf[j];
            else {
                const double slope = ((double)cdf[j + 1] - (double)cdf[j]) / ((double)centers[j + 1] - (double)centers[j]);
                r = slope * (x - (double)centers[j]) + (double)cdf[j];
            }
        }
        v[i] = (float)r;
    }
}

// img_as_ubyte: rint(x * 255) clipped (x is float32 in [0, 1]); img_as_float32: u8 * float32(1/255)
__device__ __forceinline__ int to_u8(float x) {
    float y = rintf(x * 255.0f);
    y = fminf(fmaxf(y, 0.f), 255.f);
    return (int)y;
}
__global__ __launch_bounds__(256) void he_to_u8_kernel(const float* __restrict__ img, unsigned char* __restrict__ u8, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) u8[i] = (unsigned char)to_u8(img[i]);
}

// ---------------------------------------------------------------- 'local'  (rank.equalize, disk(R))
__global__ __launch_bounds__(256) void he_local_kernel(const unsigned char* __restrict__ u8, float* __restrict__ out, int H, int W, int R) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const unsigned char* im = u8 + (int64_t)blockIdx.z * H * W;
    const int g = im[y * W + x];
    int pop = 0, sum = 0;
    const int y_lo = max(y - R, 0), y_hi = min(y + R, H - 1);
    for (int yy = y_lo; yy <= y_hi; ++yy) {
        const int dy = yy - y;
        const int half = (int)floorf(sqrtf((float)(R * R - dy * dy)));     // dx^2 + dy^2 <= R^2
        const int x_lo = max(x - half, 0), x_hi = min(x + half, W - 1);
        pop += x_hi - x_lo + 1;
        const unsigned char* row = im + yy * W;
        for (int xx = x_lo; xx <= x_hi; ++xx) sum += (row[xx] <= g);
    }
    // <uint8>(((n_bins - 1) * sum) / pop): double division, truncation
    const int o = pop ? (int)((255.0 * (double)sum) / (double)pop) : 0;
    out[(int64_t)blockIdx.z * H * W + y * W + x] = (float)o * (float)(1.0 / 255.0);
}

// ---------------------------------------------------------------- 'clahe'
// LUTs: one 256-thread workgroup per (image, tile).  The source is extended by BORDER_REFLECT_101 to whole tiles.
__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
    return p;
}
__device__ __forceinline__ int cv_round_u8(float v) {     // saturate_cast<uchar>(float): cvRound (half to even), clamp
    const float r = rintf(v);
    return (int)fminf(fmaxf(r, 0.f), 255.f);
}
__global__ __launch_bounds__(256) void he_clahe_lut_kernel(const unsigned char* __restrict__ u8, unsigned char* __restrict__ lut,
                                                            int H, int W, int tw, int th, int tiles, int clip_limit, float lut_scale) {
    __shared__ int hist[256];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, tx = tile % tiles, ty = tile / tiles;
    const unsigned char* im = u8 + (int64_t)blockIdx.y * H * W;
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < tw * th; i += 256) {
        const int yy = reflect101(ty * th + i / tw, H), xx = reflect101(tx * tw + i % tw, W);
        atomicAdd(&hist[im[yy * W + xx]], 1);
    }
    __syncthreads();
    if (tid == 0) {       // the clip / redistribution is a short serial pass over 256 bins
        if (clip_limit > 0) {
            int clipped = 0;
            for (int i = 0; i < 256; ++i) if (hist[i] > clip_limit) { clipped += hist[i] - clip_limit; hist[i] = clip_limit; }
            const int batch = clipped / 256;
            int residual = clipped - batch * 256;
            for (int i = 0; i < 256; ++i) hist[i] += batch;
            if (residual != 0) {
                const int rstep = max(256 / residual, 1);
                for (int i = 0; i < 256 && residual > 0; i += rstep, --residual) hist[i]++;
            }
        }
        int sum = 0;
        unsigned char* l = lut + ((int64_t)blockIdx.y * tiles * tiles + tile) * 256;
        for (int i = 0; i < 256; ++i) { sum += hist[i]; l[i] = (unsigned char)cv_round_u8((float)sum * lut_scale); }
    }
}
__global__ __launch_bounds__(256) void he_clahe_apply_kernel(const unsigned char* __restrict__ u8, const unsigned char* __restrict__ lut,
                                                              float* __restrict__ out, int H, int W, int tw, int th, int tiles) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i % W);
    const unsigned char* im = u8 + (int64_t)blockIdx.y * H * W;
    const unsigned char* l = lut + (int64_t)blockIdx.y * tiles * tiles * 256;
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    const float tyf = (float)y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
    ty1 = max(ty1, 0); ty2 = min(ty2, tiles - 1);
    const float txf = (float)x * inv_tw - 0.5f;
    int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
    const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
    tx1 = max(tx1, 0); tx2 = min(tx2, tiles - 1);
    const int v = im[i];
    const float a = (float)l[(ty1 * tiles + tx1) * 256 + v], b = (float)l[(ty1 * tiles + tx2) * 256 + v];
    const float c = (float)l[(ty2 * tiles + tx1) * 256 + v], d = (float)l[(ty2 * tiles + tx2) * 256 + v];
    const float res = (a * xa1 + b * xa) * ya1 + (c * xa1 + d * xa) * ya;
    out[(int64_t)blockIdx.y * H * W + i] = (float)cv_round_u8(res) * (float)(1.0 / 255.0);
}

}  // namespace

extern "C" size_t evr_hist_equalize_workspace_bytes(int n, int H, int W, int mode) {
    if (n < 1 || H < 1 || W < 1) return 0;
    if (mode == EVR_HISTEQ_GLOBAL) return 0;
    return (size_t)n * H * W + (size_t)n * 64 * 256 + 256;       // uint8 copy (+ 8x8 tile LUTs for CLAHE)
}

extern "C" int evr_hist_equalize(float* img, int n, int H, int W, int mode, void* workspace, size_t workspace_bytes,
                                 evr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EVR_REQUIRE(img && n >= 1 && H >= 1 && W >= 1, "evr_hist_equalize: bad argument");
    const int64_t HW = (int64_t)H * W;
    EVR_REQUIRE(HW < (1LL << 30), "evr_hist_equalize: image too large");
    if (mode == EVR_HISTEQ_GLOBAL) {
        hipLaunchKernelGGL(he_global_kernel, dim3(n), dim3(HE_T), 0, stream, img, (int)HW);
        EVR_LAUNCH_CHECK();
        return EVR_OK;
    }
    EVR_REQUIRE(mode == EVR_HISTEQ_LOCAL || mode == EVR_HISTEQ_CLAHE, "evr_hist_equalize: unknown mode %d", mode);
    const size_t need = evr_hist_equalize_workspace_bytes(n, H, W, mode);
    if (!workspace || workspace_bytes < need) {
        evr::set_error("evr_hist_equalize: workspace %zu B < required %zu B", workspace_bytes, need);
        return EVR_ERR_WORKSPACE;
    }
    unsigned char* u8 = (unsigned char*)workspace;
    const int64_t total = (int64_t)n * HW;
    hipLaunchKernelGGL(he_to_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, img, u8, total);
    EVR_LAUNCH_CHECK();
    if (mode == EVR_HISTEQ_LOCAL) {
        hipLaunchKernelGGL(he_local_kernel, dim3((W + 15) / 16, (H + 15) / 16, n), dim3(256), 0, stream, u8, img, H, W, 55);
        EVR_LAUNCH_CHECK();
        return EVR_OK;
    }
    // CLAHE, clipLimit 2.0, 8x8 tiles (utils/eval_metrics.py:341)
    const int tiles = 8;
    const int We = (W % tiles == 0) ? W : W + (tiles - W % tiles), He = (H % tiles == 0) ? H : H + (tiles - H % tiles);
    const int tw = We / tiles, th = He / tiles;
    const int area = tw * th;
    int clip_limit = (int)(2.0 * area / 256);
    if (clip_limit < 1) clip_limit = 1;
    const float lut_scale = (float)(255) / (float)area;
    unsigned char* lut = u8 + (((size_t)n * HW + 255) & ~(size_t)255);
    hipLaunchKernelGGL(he_clahe_lut_kernel, dim3(tiles * tiles, n), dim3(256), 0, stream, u8, lut, H, W, tw, th, tiles, clip_limit, lut_scale);
    EVR_LAUNCH_CHECK();
    hipLaunchKernelGGL(he_clahe_apply_kernel, dim3((unsigned)((HW + 255) / 256), n), dim3(256), 0, stream, u8, lut, img, H, W, tw, th, tiles);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

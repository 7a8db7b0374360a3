// Per-frame MSE and SSIM on gfx950.
//
// Reference call sites: utils/eval_metrics.py:253-255 (clip to [0,1]), :82-84 (MseMetric ->
// skimage.metrics.mean_squared_error), :95-97 (SsimMetric -> skimage.metrics.structural_similarity
// with gaussian_weights=True, sigma=1.5, use_sample_covariance=False, data_range=1.0).
// scikit-image is not part of the reference tree ("parity unpinned"): the arithmetic
// below follows scikit-image >= 0.19 for fp32 inputs:
//   mse  : fp32 (ref-img), fp32 square, mean accumulated in fp64
//   ssim : 11-tap Gaussian (sigma 1.5, truncate 3.5), scipy.ndimage.gaussian_filter semantics:
//          axis 0 then axis 1, 'reflect' borders, fp64 accumulation in scipy's symmetric order
//          x0*w0 + sum_j (x[-j]+x[j])*w[j], each pass rounded to fp32; then the fp32 SSIM map
//          S = ((2 ux uy + C1)(2 vxy + C2)) / ((ux^2+uy^2+C1)(vx+vy+C2)), mean (fp64) of S with a
//          5-pixel border cropped.
// One kernel reads each input pixel from HBM once (plus tile halos, served by L2): tiles of
// 16 x 64 outputs, inputs + halo staged in LDS, both filter passes on chip, per-tile fp64 partial
// sums; a second tiny kernel adds the partials in fixed order (deterministic).
// Algorithmic bytes per frame: 2*4*H*W.
#include "common.h"
#include <cmath>

namespace {

constexpr int R = 5;              // Gaussian radius: int(3.5*1.5+0.5)
constexpr int TH = 16, TW = 64;   // output tile
constexpr int IH = TH + 2 * R, IW = TW + 2 * R;
constexpr int VW = (IW + 2 + 3) & ~3;   // row length of the vertical-pass maps: a multiple of 4 floats, 16 floats readable from every run start

struct Gauss { double w[R + 1]; };   // w[0] centre, w[j] = weight at distance j

__device__ __forceinline__ int reflect(int i, int n) {
    // scipy 'reflect' (d c b a | a b c d | d c b a); handles images narrower than the radius
    while (i < 0 || i >= n) {
        if (i < 0) i = -i - 1;
        if (i >= n) i = 2 * n - 1 - i;
    }
    return i;
}

__global__ __launch_bounds__(256) void metrics_tile_kernel(const float* __restrict__ img, const float* __restrict__ ref,
                                                            int H, int W, int clip, unsigned which, Gauss g,
                                                            double* __restrict__ partials, int tiles_x, int tiles_y) {
    __shared__ float sx[IH][IW + 1], sy[IH][IW + 1];
    __shared__ __attribute__((aligned(16))) float v[5][TH][VW];     // after the vertical pass: ux, uy, uxx, uyy, uxy (rows 16-B aligned)
    __shared__ double red[2][4];
    const int f = blockIdx.z, ty = blockIdx.y, tx = blockIdx.x, tid = threadIdx.x;
    const float* X = ref + (int64_t)f * H * W;   // skimage argument order: (ref, img) -> im1 = ref
    const float* Y = img + (int64_t)f * H * W;
    const int y0 = ty * TH, x0 = tx * TW;

    for (int i = tid; i < IH * IW; i += 256) {
        const int r = i / IW, c = i % IW;
        const int yy = reflect(y0 + r - R, H), xx = reflect(x0 + c - R, W);
        float a = X[(int64_t)yy * W + xx], b = Y[(int64_t)yy * W + xx];
        if (clip) { a = fminf(fmaxf(a, 0.f), 1.f); b = fminf(fmaxf(b, 0.f), 1.f); }
        sx[r][c] = a; sy[r][c] = b;
    }
    __syncthreads();

    double se = 0.0, ss = 0.0;
    if (which & 1u) {   // squared error over the tile interior (every pixel belongs to one tile)
        for (int i = tid; i < TH * TW; i += 256) {
            const int r = i / TW, c = i % TW;
            if (y0 + r < H && x0 + c < W) {
                const float d = sx[r + R][c + R] - sy[r + R][c + R];
                const float d2 = d * d;
                se += (double)d2;
            }
        }
    }
    if (which & 2u) {
        // Both passes keep the scipy operation order per output value; what changed (round 3) is who computes what: a thread
        // owns FOUR adjacent outputs along the filter axis and slides over one register window of 4 + 2R inputs, so an
        // output costs 3.5 (pass 1: 28 reads per 4 outputs, two inputs) / 1 (pass 2: 16-B reads) LDS instructions instead of 22 / 55.
        // pass 1: along axis 0 (rows), for every column of the haloed tile
        for (int i = tid; i < (TH / 4) * IW; i += 256) {
            const int c = i % IW, r0 = 4 * (i / IW);
            float wx[4 + 2 * R], wy[4 + 2 * R];
#pragma unroll
            for (int m = 0; m < 4 + 2 * R; ++m) { wx[m] = sx[r0 + m][c]; wy[m] = sy[r0 + m][c]; }
            // map by map: the window's 14 values are formed (fp32) and widened ONCE, then shared by the four outputs -- the
            // kernel is bound by the fp64 pipe, and 55 of an output's 110 fp64-pipe operations were these conversions
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                double wd[4 + 2 * R];
#pragma unroll
                for (int m = 0; m < 4 + 2 * R; ++m) {
                    const float x = wx[m], y = wy[m];
                    const float e = k == 0 ? x : k == 1 ? y : k == 2 ? x * x : k == 3 ? y * y : x * y;
                    wd[m] = (double)e;
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    double a = wd[rr + R] * g.w[0];
#pragma unroll
                    for (int j = R; j >= 1; --j) a += (wd[rr + R - j] + wd[rr + R + j]) * g.w[j];   // scipy: jj = -size1 .. -1
                    v[k][r0 + rr][c] = (float)a;
                }
            }
        }
        __syncthreads();
        // pass 2: along axis 1 (columns) + SSIM map; thread = (row, run of four columns): TH x TW / 4 = 256 runs
        static_assert(TH * TW / 4 == 256 && TW % 4 == 0 && TW + 2 * R + 2 <= VW, "one run of four outputs per thread");
        const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);
        {
            const int r = tid / (TW / 4), c0 = 4 * (tid % (TW / 4));
            const int gy = y0 + r;
            float u[4][5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                double win[16];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f4v t = *(const f4v*)&v[k][r][c0 + 4 * m];
                    win[4 * m] = (double)t[0]; win[4 * m + 1] = (double)t[1]; win[4 * m + 2] = (double)t[2]; win[4 * m + 3] = (double)t[3];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double a = win[q + R] * g.w[0];
#pragma unroll
                    for (int j = R; j >= 1; --j) a += (win[q + R - j] + win[q + R + j]) * g.w[j];
                    u[q][k] = (float)a;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gx = x0 + c0 + q;
                if (gy < R || gy >= H - R || gx < R || gx >= W - R) continue;   // crop(S, 5)
                const float ux = u[q][0], uy = u[q][1], uxx = u[q][2], uyy = u[q][3], uxy = u[q][4];
                const float vx = uxx - ux * ux, vy = uyy - uy * uy, vxy = uxy - ux * uy;
                const float A1 = (2.f * ux) * uy + C1;
                const float A2 = 2.f * vxy + C2;
                const float B1 = (ux * ux + uy * uy) + C1;
                const float B2 = (vx + vy) + C2;
                const float D = B1 * B2;
                const float S = (A1 * A2) / D;
                ss += (double)S;
            }
        }
    }
    se = evr_wave_sum(se); ss = evr_wave_sum(ss);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) { red[0][wave] = se; red[1][wave] = ss; }
    __syncthreads();
    if (tid < 2) {
        const int64_t t = ((int64_t)f * tiles_y + ty) * tiles_x + tx;
        partials[t * 2 + tid] = ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3];
    }
}

__global__ __launch_bounds__(256) void metrics_reduce_kernel(const double* __restrict__ partials, double* __restrict__ out,
                                                              int n_tiles, double inv_mse, double inv_ssim, unsigned which) {
    __shared__ double red[2][4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double a = 0, b = 0;
    for (int t = tid; t < n_tiles; t += 256) {
        a += partials[((int64_t)f * n_tiles + t) * 2];
        b += partials[((int64_t)f * n_tiles + t) * 2 + 1];
    }
    a = evr_wave_sum(a); b = evr_wave_sum(b);
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (tid == 0) {
        const double sa = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        const double sb = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        out[f * 2 + 0] = (which & 1u) ? sa * inv_mse : 0.0;
        out[f * 2 + 1] = (which & 2u) ? sb * inv_ssim : 0.0;
    }
}

}  // namespace

extern "C" size_t evr_metrics_workspace_bytes(int n, int H, int W) {
    if (n < 0 || H < 1 || W < 1) return 0;
    const size_t tiles = (size_t)((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    return (size_t)n * tiles * 2 * sizeof(double) + 256;
}

extern "C" int evr_metrics(const float* img, const float* ref, int n, int H, int W, unsigned which, int clip,
                           double* out, void* workspace, size_t workspace_bytes, evr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EVR_REQUIRE(n >= 0 && H >= 1 && W >= 1, "evr_metrics: bad shape");
    if (n == 0) return EVR_OK;
    EVR_REQUIRE(img && ref && out, "evr_metrics: null pointer");
    EVR_REQUIRE(!(which & 2u) || (H > 2 * R && W > 2 * R), "evr_metrics: SSIM needs images larger than 11x11 (win_size)");
    const size_t need = evr_metrics_workspace_bytes(n, H, W);
    if (!workspace || workspace_bytes < need) {
        evr::set_error("evr_metrics: workspace %zu B < required %zu B", workspace_bytes, need);
        return EVR_ERR_WORKSPACE;
    }
    // scipy.ndimage._gaussian_kernel1d(sigma=1.5, order=0, radius=5) in fp64
    Gauss g;
    {
        const double sigma2 = 1.5 * 1.5;
        double phi[2 * R + 1], sum = 0.0;
        for (int i = -R; i <= R; ++i) { phi[i + R] = std::exp(-0.5 / sigma2 * (double)(i * i)); }
        for (int i = 0; i < 2 * R + 1; ++i) sum += phi[i];
        for (int j = 0; j <= R; ++j) g.w[j] = phi[R + j] / sum;
    }
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    double* partials = (double*)workspace;
    hipLaunchKernelGGL(metrics_tile_kernel, dim3(tiles_x, tiles_y, n), dim3(256), 0, stream, img, ref, H, W, clip, which,
                       g, partials, tiles_x, tiles_y);
    EVR_LAUNCH_CHECK();
    const double inv_mse = 1.0 / ((double)H * W);
    const double inv_ssim = (which & 2u) ? 1.0 / ((double)(H - 2 * R) * (W - 2 * R)) : 0.0;
    hipLaunchKernelGGL(metrics_reduce_kernel, dim3(n), dim3(256), 0, stream, partials, out, tiles_x * tiles_y, inv_mse,
                       inv_ssim, which);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

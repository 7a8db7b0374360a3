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
    __shared__ float v[5][TH][IW + 1];     // after the vertical pass: ux, uy, uxx, uyy, uxy
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
        // pass 1: along axis 0 (rows), for every column of the haloed tile
        for (int i = tid; i < TH * IW; i += 256) {
            const int r = i / IW, c = i % IW;
            double a[5];
            {
                const float x = sx[r + R][c], y = sy[r + R][c];
                const float xx = x * x, yy = y * y, xy = x * y;
                a[0] = x * g.w[0]; a[1] = y * g.w[0]; a[2] = xx * g.w[0]; a[3] = yy * g.w[0]; a[4] = xy * g.w[0];
            }
            for (int j = R; j >= 1; --j) {   // scipy: jj = -size1 .. -1
                const float xa = sx[r + R - j][c], xb = sx[r + R + j][c];
                const float ya = sy[r + R - j][c], yb = sy[r + R + j][c];
                const float xxa = xa * xa, xxb = xb * xb, yya = ya * ya, yyb = yb * yb, xya = xa * ya, xyb = xb * yb;
                a[0] += ((double)xa + (double)xb) * g.w[j];
                a[1] += ((double)ya + (double)yb) * g.w[j];
                a[2] += ((double)xxa + (double)xxb) * g.w[j];
                a[3] += ((double)yya + (double)yyb) * g.w[j];
                a[4] += ((double)xya + (double)xyb) * g.w[j];
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k][r][c] = (float)a[k];
        }
        __syncthreads();
        // pass 2: along axis 1 (columns) + SSIM map
        const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);
        for (int i = tid; i < TH * TW; i += 256) {
            const int r = i / TW, c = i % TW;
            const int gy = y0 + r, gx = x0 + c;
            if (gy < R || gy >= H - R || gx < R || gx >= W - R) continue;   // crop(S, 5)
            float u[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                double a = (double)v[k][r][c + R] * g.w[0];
                for (int j = R; j >= 1; --j) a += ((double)v[k][r][c + R - j] + (double)v[k][r][c + R + j]) * g.w[j];
                u[k] = (float)a;
            }
            const float ux = u[0], uy = u[1], uxx = u[2], uyy = u[3], uxy = u[4];
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

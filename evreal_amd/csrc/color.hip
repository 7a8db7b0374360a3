// Colour reconstruction helpers (ColorNet, model/model.py:46-105 of the reference; merge in utils/color_utils.py:53-88).
//
//   evr_bayer_split   event tensor [n,B,H,W] -> the four half-resolution Bayer sub-lattices R,G,B,W
//                     (model.py:54-57: R = [0::2,0::2], G = [0::2,1::2], B = [1::2,1::2], W = [1::2,0::2]) stacked as
//                     [4n,B,H/2,W/2] (sequence-major: R,G,B,W of sequence 0, then sequence 1, ...), so the recurrent
//                     network advances all four colour streams of all sequences in one batched step.
//   evr_color_merge   the five reconstructions -> one BGR uint8 frame: per-channel clip(img*255) -> uint8 (truncation,
//                     model.py:101), bilinear x2 of the four colour planes, the 1-pixel Bayer origin shifts with edge
//                     replication, G/W averaging, BGR -> CIE Lab, L replaced by the full-resolution gray reconstruction,
//                     Lab -> BGR (color_utils.py:20-88).
// PARITY: the split and the uint8 planes are pinned against the reference (tests/golden/colornet_seq.npz).  The merge
// follows OpenCV's documented formulas in floating point; OpenCV's own 8-bit fixed-point resize / Lab tables are not
// available offline (cv2 is absent), so merged pixels may differ from the reference by a few LSB -- UNPINNED.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void bayer_split_kernel(const float* __restrict__ vox, float* __restrict__ out, int n, int B,
                                                           int H, int W) {
    const int h2 = H / 2, w2 = W / 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n * 4 * B * h2 * w2;
    if (i >= total) return;
    const int x = (int)(i % w2); int64_t p = i / w2;
    const int y = (int)(p % h2); p /= h2;
    const int b = (int)(p % B); p /= B;
    const int ch = (int)(p % 4);
    const int s = (int)(p / 4);
    // R (0,0)  G (0,1)  B (1,1)  W (1,0)   as (row offset, column offset)
    const int oy = (ch >= 2) ? 1 : 0, ox = (ch == 1 || ch == 2) ? 1 : 0;
    out[i] = vox[(((int64_t)s * B + b) * H + 2 * y + oy) * W + 2 * x + ox];
}

__device__ __forceinline__ float q8(float v) {   // np.clip(img * 255, 0, 255).astype(np.uint8): truncation
    v = v * 255.f;
    v = fminf(fmaxf(v, 0.f), 255.f);
    return floorf(v);
}

// value of colour plane `ch` after cv2.resize(x2, INTER_LINEAR) and shift_image(dx, dy), at full-res pixel (y, x)
__device__ float plane_at(const float* __restrict__ pl, int h2, int w2, int y, int x, int dx, int dy) {
    const int H = 2 * h2, W = 2 * w2;
    // shift_image: np.roll by (dy, dx) then replicate the row/column just inside the wrapped border
    int sy = y - dy, sx = x - dx;
    if (dy > 0 && y < dy) sy = 0;          // X[:dy] = X[dy] (which holds source row 0 after the roll)
    if (dx > 0 && x < dx) sx = 0;
    sy = min(max(sy, 0), H - 1); sx = min(max(sx, 0), W - 1);
    // bilinear x2 (half-pixel centres, edge clamp) of the uint8 plane, result rounded to uint8
    float fy = (sy + 0.5f) * 0.5f - 0.5f, fx = (sx + 0.5f) * 0.5f - 0.5f;
    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    float ly = fy - y0, lx = fx - x0;
    if (y0 < 0) { y0 = 0; ly = 0.f; }
    if (x0 < 0) { x0 = 0; lx = 0.f; }
    if (y0 >= h2 - 1) { y0 = h2 - 1; ly = 0.f; }
    if (x0 >= w2 - 1) { x0 = w2 - 1; lx = 0.f; }
    const int y1 = min(y0 + 1, h2 - 1), x1 = min(x0 + 1, w2 - 1);
    const float a = q8(pl[y0 * w2 + x0]), b = q8(pl[y0 * w2 + x1]), c = q8(pl[y1 * w2 + x0]), d = q8(pl[y1 * w2 + x1]);
    const float v = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * c + lx * d);
    return floorf(v + 0.5f);
}

__device__ __forceinline__ float srgb_to_lin(float c) { return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float lin_to_srgb(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.f / 2.4f) - 0.055f; }
__device__ __forceinline__ float lab_f(float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.f / 116.f; }
__device__ __forceinline__ float lab_finv(float t) { const float t3 = t * t * t; return t3 > 0.008856f ? t3 : (t - 16.f / 116.f) / 7.787f; }

// planes: [n][4][h2][w2] float (R,G,B,W streams, network output), gray: [n][H][W] float; out: [n][H][W][3] uint8 BGR
__global__ __launch_bounds__(256) void color_merge_kernel(const float* __restrict__ planes, const float* __restrict__ gray,
                                                           unsigned char* __restrict__ out, int n, int h2, int w2) {
    const int H = 2 * h2, W = 2 * w2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * H * W) return;
    const int x = (int)(i % W); int64_t p = i / W;
    const int y = (int)(p % H);
    const int s = (int)(p / H);
    const float* pl = planes + (int64_t)s * 4 * h2 * w2;
    const float R = plane_at(pl, h2, w2, y, x, 0, 0);
    const float G = plane_at(pl + (int64_t)h2 * w2, h2, w2, y, x, 1, 0);
    const float Bc = plane_at(pl + 2 * (int64_t)h2 * w2, h2, w2, y, x, 1, 1);
    const float Wc = plane_at(pl + 3 * (int64_t)h2 * w2, h2, w2, y, x, 0, 1);
    const float Gm = rintf(0.5f * G + 0.5f * Wc);           // cv2.addWeighted(..., dtype=CV_8U): round half to even
    // BGR (uint8) -> Lab (8-bit convention: L*255/100, a+128, b+128), D65, sRGB gamma
    const float r = srgb_to_lin(R / 255.f), g = srgb_to_lin(Gm / 255.f), b = srgb_to_lin(Bc / 255.f);
    float X = (0.412453f * r + 0.357580f * g + 0.180423f * b) / 0.950456f;
    float Z = (0.019334f * r + 0.119193f * g + 0.950227f * b) / 1.088754f;
    float Y = 0.212671f * r + 0.715160f * g + 0.072169f * b;
    const float fxv = lab_f(X), fyv = lab_f(Y), fzv = lab_f(Z);
    float a8 = rintf(500.f * (fxv - fyv) + 128.f), b8 = rintf(200.f * (fyv - fzv) + 128.f);
    a8 = fminf(fmaxf(a8, 0.f), 255.f); b8 = fminf(fmaxf(b8, 0.f), 255.f);
    const float L8 = q8(gray[i]);                            // lab[:, :, 0] = grayscale_highres
    // Lab -> BGR
    const float L = L8 * 100.f / 255.f, aa = a8 - 128.f, bb = b8 - 128.f;
    const float fy2 = (L + 16.f) / 116.f, fx2 = fy2 + aa / 500.f, fz2 = fy2 - bb / 200.f;
    X = lab_finv(fx2) * 0.950456f; Y = lab_finv(fy2); Z = lab_finv(fz2) * 1.088754f;
    const float ro = 3.240479f * X - 1.537150f * Y - 0.498535f * Z;
    const float go = -0.969256f * X + 1.875991f * Y + 0.041556f * Z;
    const float bo = 0.055648f * X - 0.204043f * Y + 1.057311f * Z;
    auto to8 = [](float lin) { float v = rintf(lin_to_srgb(fminf(fmaxf(lin, 0.f), 1.f)) * 255.f); return (unsigned char)fminf(fmaxf(v, 0.f), 255.f); };
    out[i * 3 + 0] = to8(bo); out[i * 3 + 1] = to8(go); out[i * 3 + 2] = to8(ro);
}

}  // namespace

extern "C" int evr_bayer_split(const float* vox, int n, int B, int H, int W, float* out, evr_stream_t stream) {
    EVR_REQUIRE(vox && out && n >= 1 && B >= 1, "evr_bayer_split: bad argument");
    EVR_REQUIRE(H % 2 == 0 && W % 2 == 0, "evr_bayer_split: sensor %dx%d must have even sides", W, H);
    const int64_t total = (int64_t)n * 4 * B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(bayer_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vox, out, n, B, H, W);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

extern "C" int evr_color_merge(const float* planes, const float* gray, int n, int H, int W, unsigned char* bgr_out,
                               evr_stream_t stream) {
    EVR_REQUIRE(planes && gray && bgr_out && n >= 1, "evr_color_merge: bad argument");
    EVR_REQUIRE(H % 2 == 0 && W % 2 == 0 && H >= 4 && W >= 4, "evr_color_merge: sensor %dx%d must have even sides", W, H);
    const int64_t total = (int64_t)n * H * W;
    hipLaunchKernelGGL(color_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, planes, gray,
                       bgr_out, n, H / 2, W / 2);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

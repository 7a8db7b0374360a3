// SPADE-E2VID (model/spade_e2v.py of the reference) -- the pieces around its convolutions:
//
//   spade_pad        cropper.pad (utils/util.py:48) made explicit: Unet6.forward rewrites the first three channels of
//                    its (padded) input IN PLACE on the first frame (spade_e2v.py:146-150), padding included
//   spade_first      that first-frame rewrite per sequence: x[:, :3] -= min; if max > 0: x[:, :3] /= max; x_org = x[:, :3]
//   nearest_half     F.interpolate(segmap, size=x.size()[-2:], mode='nearest') for the half-resolution SPADE (:65)
//   spade_apply      SPADE.forward's last line (:70-72) + UpConvLayer3's ReLU (:106) [+ the next layer's skip sum]:
//                    out = relu(normalized * (1 + gamma) + beta) [+ skip]; `normalized` is the pixel-shuffled conv0
//                    output with the parameter-free BatchNorm folded into its weights (conv.hip writes it through the
//                    phase-major column groups: PixelShuffle(2) costs nothing)
//   spade_pred       conv_img(relu(x + head)) -> bn_img -> sigmoid = prev_recs (3 channels); image = their mean (:166-170)
#include "conv.h"
#include "packed.h"

namespace evr {

__global__ __launch_bounds__(256) void spade_pad_kernel(const float* __restrict__ vox, float* __restrict__ xpad, int n, int B,
                                                         int H, int W, int hp, int wp, int pad_top, int pad_left) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n * B * hp * wp;
    if (i >= total) return;
    const int x = (int)(i % wp); int64_t p = i / wp;
    const int y = (int)(p % hp); p /= hp;                 // p = n*B + b
    const int yy = y - pad_top, xx = x - pad_left;
    xpad[i] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? vox[(p * H + yy) * W + xx] : 0.f;
}

// one workgroup per sequence (the reference runs batch 1, so its tensor-wide min/max IS per sequence)
__global__ __launch_bounds__(1024) void spade_first_kernel(float* __restrict__ xpad, float* __restrict__ xorg, int B, int plane) {
    __shared__ float red[16];
    __shared__ float s_val;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* x = xpad + (int64_t)blockIdx.x * B * plane;    // channels 0..2 are the first 3 planes
    float* o = xorg + (int64_t)blockIdx.x * 3 * plane;
    const int tot = 3 * plane;
    float mn = INFINITY;
    for (int i = tid; i < tot; i += 1024) mn = fminf(mn, x[i]);
    for (int s = 32; s > 0; s >>= 1) mn = fminf(mn, __shfl_xor(mn, s, 64));
    if (lane == 0) red[wave] = mn;
    __syncthreads();
    if (tid == 0) { float m = red[0]; for (int q = 1; q < 16; ++q) m = fminf(m, red[q]); s_val = m; }
    __syncthreads();
    mn = s_val;
    float mx = -INFINITY;
    for (int i = tid; i < tot; i += 1024) { const float v = x[i] - mn; x[i] = v; mx = fmaxf(mx, v); }
    for (int s = 32; s > 0; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s, 64));
    __syncthreads();
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid == 0) { float m = red[0]; for (int q = 1; q < 16; ++q) m = fmaxf(m, red[q]); s_val = m; }
    __syncthreads();
    mx = s_val;
    for (int i = tid; i < tot; i += 1024) {
        float v = x[i];
        if (mx > 0.f) { v = v / mx; x[i] = v; }
        o[i] = v;
    }
}

__global__ __launch_bounds__(256) void nearest_half_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int h, int w) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ho = h / 2, wo = w / 2;
    if (i >= (int64_t)planes * ho * wo) return;
    const int x = (int)(i % wo); int64_t p = i / wo;
    const int y = (int)(p % ho); p /= ho;
    out[i] = in[(p * h + 2 * y) * w + 2 * x];          // nearest: src = floor(dst * in/out) = 2 * dst
}

// xn: PLAIN [pix, C]; gb: PLAIN [pix, 2C] = gamma | beta; skip: optional (PLAIN or PACKED); out: PLAIN or PACKED
__global__ __launch_bounds__(256) void spade_apply_kernel(const float* __restrict__ xn, const float* __restrict__ gb,
                                                           const float* __restrict__ skip, float* __restrict__ out, int64_t pix, int C,
                                                           int skip_packed, int out_packed) {
    const int c4n = C / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pix * c4n) return;
    const int c4 = (int)(i % c4n) * 4;
    const int64_t p = i / c4n;
    const float4 v = *(const float4*)(xn + p * C + c4);
    const float4 g = *(const float4*)(gb + p * 2 * C + c4), b = *(const float4*)(gb + p * 2 * C + C + c4);
    float4 r;
    r.x = fmaxf(__fadd_rn(__fmul_rn(v.x, __fadd_rn(1.0f, g.x)), b.x), 0.f);
    r.y = fmaxf(__fadd_rn(__fmul_rn(v.y, __fadd_rn(1.0f, g.y)), b.y), 0.f);
    r.z = fmaxf(__fadd_rn(__fmul_rn(v.z, __fadd_rn(1.0f, g.z)), b.z), 0.f);
    r.w = fmaxf(__fadd_rn(__fmul_rn(v.w, __fadd_rn(1.0f, g.w)), b.w), 0.f);
    if (skip) { const float4 s = ld4_any(skip + p * C, c4, skip_packed); r.x += s.x; r.y += s.y; r.z += s.z; r.w += s.w; }
    st4_any(out + p * C, c4, r, out_packed);
}

__global__ __launch_bounds__(256) void spade_pred_kernel(const SpadePredArgs a) {
    // 8 lanes per pixel, one float4 of the 32 channels each (as pred_kernel in conv.hip)
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = gid >> 3;
    const int sub = (int)(gid & 7);
    const int64_t total = (int64_t)a.n * a.hp * a.wp;
    const bool live = i < total;
    float acc[3] = {0.f, 0.f, 0.f};
    if (live) {
        float4 v = ld4_any(a.x + i * 32, sub * 4, a.x_packed);
        const float4 u = ld4_any(a.head + i * 32, sub * 4, a.head_packed);
        v.x = fmaxf(v.x + u.x, 0.f); v.y = fmaxf(v.y + u.y, 0.f); v.z = fmaxf(v.z + u.z, 0.f); v.w = fmaxf(v.w + u.w, 0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* w = a.wgt + k * 32 + sub * 4;
            acc[k] = fmaf(v.x, w[0], fmaf(v.y, w[1], fmaf(v.z, w[2], v.w * w[3])));
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { acc[k] += __shfl_xor(acc[k], 4, 64); acc[k] += __shfl_xor(acc[k], 2, 64); acc[k] += __shfl_xor(acc[k], 1, 64); }
    if (live && sub == 0) {
        const int n = (int)(i / ((int64_t)a.hp * a.wp));
        const int rem = (int)(i - (int64_t)n * a.hp * a.wp);
        const int y = rem / a.wp, x = rem - y * a.wp;
        float s[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            s[k] = 1.0f / (1.0f + expf(-(acc[k] + a.bias[k])));                    // bn_img folded into wgt / bias
            a.prev[(((int64_t)n * 3 + k) * a.hp + y) * a.wp + x] = s[k];
        }
        const int yy = y - a.iy0, xx = x - a.ix0;
        if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W)
            a.img[((int64_t)n * a.H + yy) * a.W + xx] = ((s[0] + s[1]) + s[2]) / 3.0f;  // x.mean(1)
    }
}

int launch_spade_pad(const float* vox, float* xpad, int n, int B, int H, int W, int hp, int wp, int pad_top, int pad_left, hipStream_t stream) {
    const int64_t total = (int64_t)n * B * hp * wp;
    hipLaunchKernelGGL(spade_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, vox, xpad, n, B, H, W, hp, wp, pad_top, pad_left);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}
int launch_spade_first(float* xpad, float* xorg, int n, int B, int hp, int wp, hipStream_t stream) {
    EVR_REQUIRE(B >= 3, "SPADE-E2VID takes its first segmentation map from 3 input channels (num_bins %d)", B);
    hipLaunchKernelGGL(spade_first_kernel, dim3(n), dim3(1024), 0, stream, xpad, xorg, B, hp * wp);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}
int launch_nearest_half(const float* in, float* out, int planes, int h, int w, hipStream_t stream) {
    const int64_t total = (int64_t)planes * (h / 2) * (w / 2);
    hipLaunchKernelGGL(nearest_half_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, out, planes, h, w);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}
int launch_spade_apply(const float* xn, const float* gb, const float* skip, float* out, int64_t pix, int C, int skip_packed, int out_packed, hipStream_t stream) {
    EVR_REQUIRE(C % 8 == 0, "spade_apply: %d channels", C);
    const int64_t total = pix * (C / 4);
    hipLaunchKernelGGL(spade_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, xn, gb, skip, out, pix, C, skip_packed, out_packed);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}
int launch_spade_pred(const SpadePredArgs& a, hipStream_t stream) {
    const int64_t total = (int64_t)a.n * a.hp * a.wp * 8;
    hipLaunchKernelGGL(spade_pred_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

}  // namespace evr

// ---------------------------------------------------------------------------------------------------
// InstanceNorm2d (no running statistics, no affine) of the ResidualBlocks built with norm='IN' (model/submodules.py:
// 160-162,171-179): per (image, channel) mean / biased variance over H x W, y = (x - mean) / sqrt(var + 1e-5), then
// relu, or + residual then relu, [+ the next decoder's fused skip].  x: PLAIN [n, hw, c]; one 256-thread workgroup per
// (image, 32 channels): lane % 32 = channel (a pixel's 32 channels are one 128-B line), 8 pixel stripes.
namespace evr {

__global__ __launch_bounds__(256) void instnorm_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ skip,
                                                        float* __restrict__ out, int hw, int c, int res_packed, int skip_packed, int out_packed) {
    __shared__ double s1[8][32], s2[8][32];
    __shared__ float s_mean[32], s_inv[32];
    const int tid = threadIdx.x, ch = tid & 31, stripe = tid >> 5;
    const int c0 = blockIdx.x * 32;
    const int64_t base = (int64_t)blockIdx.y * hw * c;
    double a = 0.0, b = 0.0;
    for (int i = stripe; i < hw; i += 8) {
        const double v = (double)x[base + (int64_t)i * c + c0 + ch];
        a += v; b += v * v;
    }
    s1[stripe][ch] = a; s2[stripe][ch] = b;
    __syncthreads();
    if (tid < 32) {
        double sa = 0.0, sb = 0.0;
        for (int q = 0; q < 8; ++q) { sa += s1[q][tid]; sb += s2[q][tid]; }
        const double mean = sa / hw;
        double var = sb / hw - mean * mean;            // biased variance (F.instance_norm / batch_norm training statistics)
        if (var < 0.0) var = 0.0;
        s_mean[tid] = (float)mean;
        s_inv[tid] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    // apply: a thread handles 4 consecutive channels of a pixel (16-B accesses; PACKED runs are 4 channels)
    const int c4 = (tid & 7) * 4, prow = tid >> 3;      // 8 runs per 32 channels, 32 pixel rows per pass
    for (int i = prow; i < hw; i += 32) {
        const int64_t o = base + (int64_t)i * c;
        float4 v = *(const float4*)(x + o + c0 + c4);
        v.x = (v.x - s_mean[c4]) * s_inv[c4]; v.y = (v.y - s_mean[c4 + 1]) * s_inv[c4 + 1];
        v.z = (v.z - s_mean[c4 + 2]) * s_inv[c4 + 2]; v.w = (v.w - s_mean[c4 + 3]) * s_inv[c4 + 3];
        if (res) { const float4 r = ld4_any(res + o, c0 + c4, res_packed); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (skip) { const float4 k = ld4_any(skip + o, c0 + c4, skip_packed); v.x += k.x; v.y += k.y; v.z += k.z; v.w += k.w; }
        st4_any(out + o, c0 + c4, v, out_packed);
    }
}

int launch_instnorm(const float* x, const float* res, const float* skip, float* out, int n, int hw, int c, int res_packed,
                    int skip_packed, int out_packed, hipStream_t stream) {
    EVR_REQUIRE(c % 32 == 0, "instance norm: %d channels (need a multiple of 32)", c);
    hipLaunchKernelGGL(instnorm_kernel, dim3(c / 32, n), dim3(256), 0, stream, x, res, skip, out, hw, c, res_packed, skip_packed, out_packed);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

}  // namespace evr

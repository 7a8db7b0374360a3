// LPIPS (AlexNet, v0.1) on gfx950 -- the 'lpips' metric of the reference (utils/eval_metrics.py:100-156:
// pyiqa.create_metric('lpips') on [n,3,H,W] batches built by cv2torch(img, num_ch=3), eval_utils.py:46-54).
//
// PARITY UNPINNED: pyiqa is neither in the reference tree nor installed, and its AlexNet + linear-head weights are
// downloaded at run time (unobtainable offline).  The arithmetic below follows the published algorithm (Zhang et
// al. 2018; richzhang/PerceptualSimilarity v0.1 as wrapped by pyiqa): x <- 2x-1; (x-shift)/scale per channel;
// AlexNet features after relu1..relu5; unit-normalise every pixel's feature vector (eps 1e-10); squared
// difference; non-negative 1x1 "lin" weights; spatial mean; sum over the five layers.  Scores therefore agree with
// the oracle restatement (oracle/lpips.py) on any weights, and with pyiqa only once its weights are supplied.
//
// Layers: conv1 (3->64, k11 s4 p2) is a direct VALU kernel on the gray input (the three input channels are affine
// in the same gray value); conv2..conv5 run on the convolution kernels of conv.hip (split arithmetic: conv2 on the implicit
// GEMM, conv3..conv5 on the band kernel; pool1/feat1.. are then PACKED tensors, conv.h); 3x3/2 max pools and the
// per-layer score are NHWC streaming kernels (one wave per pixel for the channel norms).
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <atomic>
#include "conv.h"
#include "packed.h"

using namespace evr;

namespace {

// ---- conv1: gray [n2,H,W] -> NHWC [n2,h1,w1,64], relu ------------------------------------------------------------
// The three input channels are the same gray value g pushed through per-channel affine maps
// x_c = ((2g-1) - shift_c)/scale_c = alpha_c*g + beta_c (and 0 where the conv pads), so
//   sum_c sum_t w[co][c][t] x_c[t] = sum_t inside[t] * (wg[t][co]*g[t] + wb[t][co]),
// with wg = sum_c w*alpha_c, wb = sum_c w*beta_c folded on the host: a ONE-channel 11x11 conv (121 taps instead of
// 363) plus, for output tiles whose receptive field leaves the image, the wb term under the inside mask; interior
// tiles take sum_t wb[t] from the bias.  Weights are wave-uniform (scalar loads); the gray tile and mask sit in LDS.
struct Conv1Args {
    const float* img; const float* ref;   // [n,H,W] each; batch index >= n reads ref
    int n, H, W, h1, w1, clip;
    const float* wg;                       // [121][64]
    const float* wb;                       // [121][64]
    const float* bias;                     // [64]
    const float* bias_in;                  // [64] bias + sum_t wb[t]  (interior tiles)
    float* out;
    const unsigned* wfrag;                 // matrix-core form: wg in MFMA-fragment order (conv1_pack_wfrag), bf16 hi / lo
    const float* wbsum;                    // [12][12][64] prefix sums of wb over (ky, kx): the padding term of a border pixel
};

__global__ __launch_bounds__(256) void lpips_conv1_kernel(const Conv1Args a) {
    constexpr int TS = 16, K = 11, S = 4, P = 2, IS = (TS - 1) * S + K;   // 71
    extern __shared__ float smem[];
    float* tile = smem;                 // [IS][IS] gray (0 outside the image)
    float* mask = smem + IS * IS;       // [IS][IS] 1 inside / 0 outside
    const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS, tid = threadIdx.x;
    const float* src = (b < a.n ? a.img + (int64_t)b * a.H * a.W : a.ref + (int64_t)(b - a.n) * a.H * a.W);
    const int y_lo = ty0 * S - P, x_lo = tx0 * S - P;
    const bool interior = y_lo >= 0 && x_lo >= 0 && y_lo + IS <= a.H && x_lo + IS <= a.W;   // block-uniform
    for (int i = tid; i < IS * IS; i += 256) {
        const int rr = i / IS, cc = i % IS;
        const int y = y_lo + rr, x = x_lo + cc;
        float g = 0.f, in = 0.f;
        if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) {
            g = src[(int64_t)y * a.W + x];
            if (a.clip) g = fminf(fmaxf(g, 0.f), 1.f);
            in = 1.f;
        }
        tile[i] = g; mask[i] = in;
    }
    __syncthreads();
    const int ly = tid / TS, lx = tid % TS, oy = ty0 + ly, ox = tx0 + lx;
    float acc[64];
    if (interior) {
#pragma unroll
        for (int co = 0; co < 64; ++co) acc[co] = a.bias_in[co];
        for (int ky = 0; ky < K; ++ky)
            for (int kx = 0; kx < K; ++kx) {
                const float v = tile[(ly * S + ky) * IS + lx * S + kx];
                const float* w = a.wg + (ky * K + kx) * 64;                 // wave-uniform -> scalar loads
#pragma unroll
                for (int co = 0; co < 64; ++co) acc[co] = fmaf(v, w[co], acc[co]);
            }
    } else {
#pragma unroll
        for (int co = 0; co < 64; ++co) acc[co] = a.bias[co];
        for (int ky = 0; ky < K; ++ky)
            for (int kx = 0; kx < K; ++kx) {
                const int ti = (ly * S + ky) * IS + lx * S + kx;
                const float v = tile[ti], mk = mask[ti];
                const float* w = a.wg + (ky * K + kx) * 64;
                const float* wbp = a.wb + (ky * K + kx) * 64;
#pragma unroll
                for (int co = 0; co < 64; ++co) acc[co] = fmaf(mk, wbp[co], fmaf(v, w[co], acc[co]));
            }
    }
    if (oy < a.h1 && ox < a.w1) {
        float* o = a.out + (((int64_t)b * a.h1 + oy) * a.w1 + ox) * 64;
#pragma unroll
        for (int co = 0; co < 64; co += 4)
            *(float4*)(o + co) = make_float4(fmaxf(acc[co], 0.f), fmaxf(acc[co + 1], 0.f), fmaxf(acc[co + 2], 0.f), fmaxf(acc[co + 3], 0.f));
    }
}

// conv1 on the matrix cores (split mode): GEMM M = pixels, N = 64, K = 121 taps padded to 128, three bf16 products
// (x = hi + lo, w = hi + lo) like the networks' head conv.  The gray tile sits in LDS; lane (pixel r, half h) gathers the 8
// taps k = 16s + 8h + j of slab s at compile-time offsets from its pixel's window origin; the weight fragments (32 KB, packed
// on the host) are copied to LDS once per block.  The padding term sum_t inside[t] * wb[t] is a rectangle sum over the
// window's in-image taps: 4 look-ups in the prefix table for border pixels, the folded bias for the others.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void lpips_conv1_mfma_kernel(const Conv1Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TS = 16, K = 11, S = 4, P = 2, IS = (TS - 1) * S + K;   // 71
    extern __shared__ float smem[];
    float* tile = smem;                                   // [IS][IS] gray (0 outside the image)
    const float4* wf = (const float4*)(smem + IS * IS + 3);   // [8 slabs][2 blocks][hi, lo][64 lanes] x 16 B (16-B aligned: 5041 + 3)
    const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS, tid = threadIdx.x;
    const float* src = (b < a.n ? a.img + (int64_t)b * a.H * a.W : a.ref + (int64_t)(b - a.n) * a.H * a.W);
    const int y_lo = ty0 * S - P, x_lo = tx0 * S - P;
    for (int i = tid; i < IS * IS; i += 256) {
        const int rr = i / IS, cc = i % IS;
        const int y = y_lo + rr, x = x_lo + cc;
        float g = 0.f;
        if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) {
            g = src[(int64_t)y * a.W + x];
            if (a.clip) g = fminf(fmaxf(g, 0.f), 1.f);
        }
        tile[i] = g;
    }
    {
        float4* dst = (float4*)(smem + IS * IS + 3);
        const float4* wsrc = (const float4*)a.wfrag;
        for (int i = tid; i < 8 * 2 * 2 * 64; i += 256) dst[i] = wsrc[i];
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nb][i] = 0.f;
    int base[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int ly = 2 * (2 * wv + mt) + (r >> 4), lx = r & 15;
        base[mt] = ly * S * IS + lx * S;
    }
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
        u32x4_t ah[2], al[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                constexpr int NT = K * K;
                const int k0 = 16 * s8 + j, k1 = k0 + 8;                              // this lane's tap for h = 0 / 1
                const int o0 = k0 < NT ? (k0 / K) * IS + k0 % K : 0, o1 = k1 < NT ? (k1 / K) * IS + k1 % K : 0;   // (padding taps: zero weights)
                e[j] = tile[base[mt] + (h ? o1 : o0)];
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                ah[mt][d] = cvt_pk_bf16(e[2 * d], e[2 * d + 1]);
                al[mt][d] = cvt_pk_bf16(e[2 * d] - __uint_as_float(ah[mt][d] << 16), e[2 * d + 1] - __uint_as_float(ah[mt][d] & 0xffff0000u));
            }
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const bf16x8 b_hi = __builtin_bit_cast(bf16x8, wf[((s8 * 2 + nb) * 2 + 0) * 64 + lane]);
            const bf16x8 b_lo = __builtin_bit_cast(bf16x8, wf[((s8 * 2 + nb) * 2 + 1) * 64 + lane]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const bf16x8 a_hi = __builtin_bit_cast(bf16x8, ah[mt]), a_lo = __builtin_bit_cast(bf16x8, al[mt]);
                acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_hi, a_lo, acc[mt][nb], 0, 0, 0);
                acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_lo, a_hi, acc[mt][nb], 0, 0, 0);
                acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_hi, a_hi, acc[mt][nb], 0, 0, 0);
            }
        }
    }
    // epilogue: lane = pixel r, registers = channels nb*32 + (i & 3) + 8 * (i >> 2) + 4h
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int ly = 2 * (2 * wv + mt) + (r >> 4), lx = r & 15, oy = ty0 + ly, ox = tx0 + lx;
        if (oy >= a.h1 || ox >= a.w1) continue;
        // in-image taps of this pixel's window: ky in [ky0, ky1], kx in [kx0, kx1]
        const int ky0 = max(0, P - oy * S), ky1 = min(K - 1, a.H - 1 - oy * S + P);
        const int kx0 = max(0, P - ox * S), kx1 = min(K - 1, a.W - 1 - ox * S + P);
        const bool inside = ky0 == 0 && kx0 == 0 && ky1 == K - 1 && kx1 == K - 1;
        float* o = a.out + (((int64_t)b * a.h1 + oy) * a.w1 + ox) * 64;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c4 = nb * 32 + 8 * q + 4 * h;
                float4 add;
                if (inside) add = *(const float4*)(a.bias_in + c4);
                else {
                    const float4 bb = *(const float4*)(a.bias + c4);
                    const float4 p11 = *(const float4*)(a.wbsum + ((ky1 + 1) * 12 + kx1 + 1) * 64 + c4), p01 = *(const float4*)(a.wbsum + (ky0 * 12 + kx1 + 1) * 64 + c4);
                    const float4 p10 = *(const float4*)(a.wbsum + ((ky1 + 1) * 12 + kx0) * 64 + c4), p00 = *(const float4*)(a.wbsum + (ky0 * 12 + kx0) * 64 + c4);
                    add = make_float4(bb.x + ((p11.x - p01.x) - (p10.x - p00.x)), bb.y + ((p11.y - p01.y) - (p10.y - p00.y)),
                                      bb.z + ((p11.z - p01.z) - (p10.z - p00.z)), bb.w + ((p11.w - p01.w) - (p10.w - p00.w)));
                }
                *(float4*)(o + c4) = make_float4(fmaxf(acc[mt][nb][4 * q] + add.x, 0.f), fmaxf(acc[mt][nb][4 * q + 1] + add.y, 0.f),
                                                 fmaxf(acc[mt][nb][4 * q + 2] + add.z, 0.f), fmaxf(acc[mt][nb][4 * q + 3] + add.w, 0.f));
            }
    }
#endif
}

// ---- max pool 3x3 stride 2 (no padding), NHWC --------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int h, int w,
                                                          int c, int ho, int wo, int in_packed, int out_packed) {
    const int c4n = c / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * ho * wo * c4n) return;
    const int c4 = (int)(i % c4n); int64_t p = i / c4n;
    const int ox = (int)(p % wo); p /= wo;
    const int oy = (int)(p % ho);
    const int b = (int)(p / ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            const float4 v = ld4_any(in + (((int64_t)b * h + 2 * oy + ky) * w + 2 * ox + kx) * c, c4 * 4, in_packed);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    st4_any(out + (((int64_t)b * ho + oy) * wo + ox) * c, c4 * 4, m, out_packed);
}

// ---- per-layer score ------------------------------------------------------------------------------------------------
// feat: NHWC [2n, h, w, C] (first n = img, last n = ref); out partial[n][blocks] of sum over pixels.
// A lane owns 4-channel runs (one 16-B access on PLAIN rows, 8 B + 4 B on PACKED ones): R = C / 4 runs per pixel.  With
// R = 16 (relu1, C = 64) a wave takes four pixels at once (16-lane reductions); otherwise one pixel per wave and up to two
// runs per lane (C <= 512).
struct ScoreArgs { const float* feat[5]; const float* lin[5]; int hw[5], C[5], packed[5]; };
// (one launch for the five layers -- grid.z = layer -- so their small grids share the chip instead of queueing one by one)
__global__ __launch_bounds__(256) void lpips_score_kernel(const ScoreArgs sa, int n, double* __restrict__ partials_all, int blocks_per_img, int layer0) {
    __shared__ double red[4];
    const int layer = layer0 + blockIdx.z;
    const float* __restrict__ feat = sa.feat[layer]; const float* __restrict__ lin = sa.lin[layer];
    const int hw = sa.hw[layer], C = sa.C[layer], packed = sa.packed[layer];
    double* __restrict__ partials = partials_all + (size_t)layer * n * blocks_per_img;
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* f0 = feat + (int64_t)b * hw * C;
    const float* f1 = feat + (int64_t)(b + n) * hw * C;
    const int R = C >> 2;
    const int P = (R <= 16) ? 4 : 1;                    // pixels per wave
    const int sub = (P == 4) ? (lane >> 4) : 0;         // the lane's pixel inside the wave
    const int run0 = (P == 4) ? (lane & 15) : lane;
    double acc = 0.0;
    for (int p0 = (blockIdx.x * 4 + wave) * P; p0 < hw; p0 += blocks_per_img * 4 * P) {
        const int p = p0 + sub;
        const bool pok = p < hw;
        float4 a[2], c[2];
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int run = run0 + 64 * k;
            a[k] = make_float4(0.f, 0.f, 0.f, 0.f); c[k] = a[k];
            if (pok && run < R && (k == 0 || P == 1)) {
                a[k] = ld4_any(f0 + (int64_t)p * C, run * 4, packed);
                c[k] = ld4_any(f1 + (int64_t)p * C, run * 4, packed);
            }
            s0 += (a[k].x * a[k].x + a[k].y * a[k].y) + (a[k].z * a[k].z + a[k].w * a[k].w);
            s1 += (c[k].x * c[k].x + c[k].y * c[k].y) + (c[k].z * c[k].z + c[k].w * c[k].w);
        }
        const int top = (P == 4) ? 8 : 32;               // reduce over the lanes of one pixel
        for (int o = top; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
        const float n0 = sqrtf(s0) + 1e-10f, n1 = sqrtf(s1) + 1e-10f;
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int run = run0 + 64 * k;
            if (pok && run < R && (k == 0 || P == 1)) {
                const float4 w = *(const float4*)(lin + run * 4);
                const float e0 = a[k].x / n0 - c[k].x / n1, e1 = a[k].y / n0 - c[k].y / n1;
                const float e2 = a[k].z / n0 - c[k].z / n1, e3 = a[k].w / n0 - c[k].w / n1;
                d = fmaf(w.x, e0 * e0, d); d = fmaf(w.y, e1 * e1, d); d = fmaf(w.z, e2 * e2, d); d = fmaf(w.w, e3 * e3, d);
            }
        }
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);      // all pixels of the wave together
        acc += (double)d;
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[(int64_t)b * blocks_per_img + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// one wave per image: lanes stride over a layer's partial sums, fixed-order tree (deterministic), layers added in order
// (round 5: one THREAD per image walked 5 x blocks_per_img dependent loads -- 19 us of a one-sequence frame's evaluation chain)
__global__ __launch_bounds__(64) void lpips_final_kernel(const double* __restrict__ partials, double* __restrict__ out, int blocks_per_img, int nlayers,
                                   int n, const int* __restrict__ hw) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= n) return;
    double total = 0.0;
    for (int l = 0; l < nlayers; ++l) {
        double s = 0.0;
        for (int k = lane; k < blocks_per_img; k += 64) s += partials[((int64_t)l * n + b) * blocks_per_img + k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        total += s / (double)hw[l];
    }
    if (lane == 0) out[b] = total;
}

// work-groups per image and layer of the score kernel: 32 at 64 images (2048 per launch); a small batch takes more of them -- at one
// image 32 work-groups walked relu1's 5440 pixels in eleven dependent rounds (16 us per layer)
constexpr int SCORE_BLOCKS_MAX = 256;
static int score_blocks(int n) { int b = 2048 / (n > 0 ? n : 1); return b < 32 ? 32 : (b > SCORE_BLOCKS_MAX ? SCORE_BLOCKS_MAX : b); }

struct Layer { int cin, cout, k, pad; int x3 = 0; int mx_e = 0; std::vector<float> w, b; float* d_w = nullptr; float* d_b = nullptr; float* d_lin = nullptr;
               float* d_wino = nullptr; };      // (exact-fp32 mode, 3x3 layers: Winograd-domain weights, wino.hip)

}  // namespace

struct evr_lpips {
    // conv1 (direct) + 4 igemm layers
    std::vector<float> wg, wb, b1, b1in, wfrag1, wbsum;
    float* d_wg = nullptr; float* d_wb = nullptr; float* d_b1 = nullptr; float* d_b1in = nullptr;
    float* d_wfrag1 = nullptr; float* d_wbsum = nullptr;     // matrix-core form of conv1 (split mode)
    float* d_lin[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    Layer L[4];
    // shape-dependent.  Buffers are sized for `cap` image pairs at H x W and only ever grow; what depends on the pair count n of a call
    // (the four ConvArgs with their tile choice) lives in a small cache of plans.  Round 6: the drop-in's chunks change n at every
    // sequence boundary (a short last chunk, then a full one), and the old one-plan form answered each change with a stream
    // synchronise + hipFree + hipMalloc of everything -- 16 device-wide stalls in an eight-sequence run.
    int n = 0, H = 0, W = 0, cap = 0;
    int h[5], w[5];              // feature map sizes of relu1..relu5
    std::vector<void*> allocs;
    float* feat[5] = {}; float* pool1 = nullptr; float* pool2 = nullptr;
    struct Plan { int n = 0; unsigned stamp = 0; ConvArgs args[4]; int wm[4], nb[4]; };
    static constexpr int NPLANS = 8;
    Plan plans[NPLANS]; unsigned clock = 0; Plan* cur = nullptr;
    ConvArgs* d_args = nullptr;      // [NPLANS][4]
    float* kws = nullptr; bool kws_tried = false;
    double* partials = nullptr; int* d_hw = nullptr;
    void release() {
        for (void* p : allocs) (void)hipFree(p);
        allocs.clear(); n = 0; cap = 0; d_args = nullptr; cur = nullptr; kws = nullptr; kws_tried = false;
        for (auto& pl : plans) pl.n = 0;
    }
    ~evr_lpips() {
        release();
        if (d_wg) (void)hipFree(d_wg); if (d_wb) (void)hipFree(d_wb); if (d_b1) (void)hipFree(d_b1); if (d_b1in) (void)hipFree(d_b1in);
        if (d_wfrag1) (void)hipFree(d_wfrag1); if (d_wbsum) (void)hipFree(d_wbsum);
        for (auto& p : d_lin) if (p) (void)hipFree(p);
        for (auto& l : L) { if (l.d_w) (void)hipFree(l.d_w); if (l.d_b) (void)hipFree(l.d_b); if (l.d_wino) (void)hipFree(l.d_wino); }
    }
};

namespace {
int up(const std::vector<float>& h, float** d) {
    EVR_HIP(hipMalloc((void**)d, h.size() * sizeof(float) + 64));
    EVR_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return EVR_OK;
}
template <typename T>
int dalloc(evr_lpips* m, T** p, size_t count) {
    EVR_HIP(hipMalloc((void**)p, count * sizeof(T) + 256));
    m->allocs.push_back((void*)*p);
    return EVR_OK;
}
}  // namespace

extern "C" int evr_lpips_create(const evr_tensor* tensors, int n_tensors, evr_lpips** out) {
    EVR_REQUIRE(tensors && out, "evr_lpips_create: null argument");
    std::map<std::string, const evr_tensor*> sd;
    for (int i = 0; i < n_tensors; ++i) if (tensors[i].name) sd[tensors[i].name] = &tensors[i];
    auto get = [&](const std::string& k, int64_t numel, const evr_tensor** t) -> int {
        auto it = sd.find(k);
        if (it == sd.end()) { set_error("LPIPS state_dict is missing '%s'", k.c_str()); return EVR_ERR_MISSING_TENSOR; }
        int64_t ne = 1; for (int d = 0; d < it->second->ndim; ++d) ne *= it->second->shape[d];
        if (ne != numel) { set_error("LPIPS tensor '%s' has %lld elements, expected %lld", k.c_str(), (long long)ne, (long long)numel); return EVR_ERR_INVALID; }
        *t = it->second; return EVR_OK;
    };
    evr_lpips* m = new evr_lpips();
    int rc = EVR_OK;
    const evr_tensor *w, *b;
    // torchvision alexnet.features indices 0,3,6,8,10 inside pyiqa's slices
    const char* names[5] = {"net.slice1.0", "net.slice2.3", "net.slice3.6", "net.slice4.8", "net.slice5.10"};
    const int cin[5] = {3, 64, 192, 384, 256}, cout[5] = {64, 192, 384, 256, 256}, ks[5] = {11, 5, 3, 3, 3};
    do {
        if ((rc = get(std::string(names[0]) + ".weight", 64 * 3 * 121, &w))) break;
        if ((rc = get(std::string(names[0]) + ".bias", 64, &b))) break;
        m->wg.assign((size_t)121 * 64, 0.f); m->wb.assign((size_t)121 * 64, 0.f);
        m->b1.assign(b->data_host, b->data_host + 64); m->b1in.assign(64, 0.f);
        {   // scaling layer of LPIPS folded into conv1: x_c = alpha_c * g + beta_c
            const double shift[3] = {-.030, -.088, -.188}, scale[3] = {.458, .448, .450};
            for (int co = 0; co < 64; ++co) {
                double bsum = 0.0;
                for (int t = 0; t < 121; ++t) {
                    double g = 0.0, bb = 0.0;
                    for (int c = 0; c < 3; ++c) {
                        const double wv = w->data_host[((size_t)co * 3 + c) * 121 + t];
                        g += wv * (2.0 / scale[c]); bb += wv * ((-1.0 - shift[c]) / scale[c]);
                    }
                    m->wg[(size_t)t * 64 + co] = (float)g; m->wb[(size_t)t * 64 + co] = (float)bb;
                    bsum += bb;
                }
                m->b1in[co] = (float)((double)m->b1[co] + bsum);
            }
        }
        if ((rc = up(m->wg, &m->d_wg))) break;
        if ((rc = up(m->wb, &m->d_wb))) break;
        if ((rc = up(m->b1, &m->d_b1))) break;
        if ((rc = up(m->b1in, &m->d_b1in))) break;
        if (use_split_mode()) {
            // weight fragments of lpips_conv1_mfma_kernel: [slab s][block nb][hi, lo][lane] x 4 dwords, lane (r, h) holding the
            // 8 taps k = 16s + 8h + j of output channel nb*32 + r as bf16 pairs (taps >= 121: zero)
            m->wfrag1.assign((size_t)8 * 2 * 2 * 64 * 4, 0.f);
            unsigned* wf = (unsigned*)m->wfrag1.data();
            for (int s8 = 0; s8 < 8; ++s8) for (int nb = 0; nb < 2; ++nb) for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 31, hh = lane >> 5;
                unsigned short hi[8], lo[8];
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s8 + 8 * hh + j;
                    const float wv = k < 121 ? m->wg[(size_t)k * 64 + nb * 32 + r] : 0.f;
                    hi[j] = bf16_rne(wv); lo[j] = bf16_rne(wv - bf16_to_f32(hi[j]));
                }
                for (int d = 0; d < 4; ++d) {
                    wf[((((size_t)s8 * 2 + nb) * 2 + 0) * 64 + lane) * 4 + d] = (unsigned)hi[2 * d] | ((unsigned)hi[2 * d + 1] << 16);
                    wf[((((size_t)s8 * 2 + nb) * 2 + 1) * 64 + lane) * 4 + d] = (unsigned)lo[2 * d] | ((unsigned)lo[2 * d + 1] << 16);
                }
            }
            // prefix sums of wb over the window: P[i][j][co] = sum_{ky < i, kx < j} wb[ky][kx][co] (fp64, rounded once)
            m->wbsum.assign((size_t)12 * 12 * 64, 0.f);
            for (int co = 0; co < 64; ++co) {
                double Pd[12][12] = {};
                for (int i = 1; i < 12; ++i) for (int j = 1; j < 12; ++j)
                    Pd[i][j] = Pd[i - 1][j] + Pd[i][j - 1] - Pd[i - 1][j - 1] + (double)m->wb[(size_t)((i - 1) * 11 + (j - 1)) * 64 + co];
                for (int i = 0; i < 12; ++i) for (int j = 0; j < 12; ++j) m->wbsum[(size_t)(i * 12 + j) * 64 + co] = (float)Pd[i][j];
            }
            if ((rc = up(m->wfrag1, &m->d_wfrag1))) break;
            if ((rc = up(m->wbsum, &m->d_wbsum))) break;
        }
        for (int l = 1; l < 5 && !rc; ++l) {
            Layer& L = m->L[l - 1];
            L.cin = cin[l]; L.cout = cout[l]; L.k = ks[l]; L.pad = ks[l] / 2;
            const int taps = L.k * L.k;
            if ((rc = get(std::string(names[l]) + ".weight", (int64_t)cout[l] * cin[l] * taps, &w))) break;
            if ((rc = get(std::string(names[l]) + ".bias", cout[l], &b))) break;
            L.w.assign((size_t)cout[l] * taps * cin[l], 0.f); L.b.assign(b->data_host, b->data_host + cout[l]);
            for (int co = 0; co < cout[l]; ++co) for (int ci = 0; ci < cin[l]; ++ci) for (int t = 0; t < taps; ++t)
                L.w[((size_t)co * taps + t) * cin[l] + ci] = w->data_host[((size_t)co * cin[l] + ci) * taps + t];
            L.x3 = arith_mode();          // conv2..conv5 run on the same implicit-GEMM family as the networks, in their mode
            if (L.x3 == 4) L.x3 = 2;      // (P6 tensors have no 4-channel writer -- the max pools, conv1's epilogue: LPIPS keeps the fp8 form)
            // exact-fp32 mode: conv3..conv5 (3x3 stride 1) also get their Winograd-domain weights (model.cpp finish_conv; EVR_WINO=0: never)
            if (L.x3 == 0 && L.k == 3 && wino_enabled() && L.cin % 16 == 0 && L.cout % 64 == 0) {
                std::vector<float> u;
                wino_pack_weights(L.w, L.cout, L.cin, 0, u);
                if ((rc = up(u, &L.d_wino))) break;
            }
            if (L.x3) L.mx_e = pack_weights_for(L.x3, L.w);
            if (L.x3 == 3) for (float& bv : L.b) bv = std::ldexp(bv, L.mx_e + H2_ACT_EXP);      // (model.cpp finish_conv)
            if ((rc = up(L.w, &L.d_w))) break;
            if ((rc = up(L.b, &L.d_b))) break;
        }
        if (rc) break;
        for (int l = 0; l < 5 && !rc; ++l) {
            if ((rc = get("lin" + std::to_string(l) + ".model.1.weight", cout[l], &w))) break;
            std::vector<float> lw(w->data_host, w->data_host + cout[l]);
            rc = up(lw, &m->d_lin[l]);
        }
    } while (0);
    if (rc) { delete m; return rc; }
    *out = m;
    return EVR_OK;
}

extern "C" int evr_lpips_destroy(evr_lpips* m) { delete m; return EVR_OK; }

// (re)allocate the feature buffers for `cap` image pairs of H x W: the only place that synchronises
static int lpips_buffers(evr_lpips* m, int cap, int H, int W, hipStream_t stream) {
    EVR_HIP(hipStreamSynchronize(stream));
    m->release();
    m->cap = cap; m->H = H; m->W = W;
    const int n2 = 2 * cap;
    m->h[0] = (H + 4 - 11) / 4 + 1; m->w[0] = (W + 4 - 11) / 4 + 1;
    const int hp1 = (m->h[0] - 3) / 2 + 1, wp1 = (m->w[0] - 3) / 2 + 1;
    m->h[1] = hp1; m->w[1] = wp1;
    const int hp2 = (hp1 - 3) / 2 + 1, wp2 = (wp1 - 3) / 2 + 1;
    for (int l = 2; l < 5; ++l) { m->h[l] = hp2; m->w[l] = wp2; }
    if (!(hp2 >= 1 && wp2 >= 1)) { m->cap = 0; m->H = m->W = 0; }
    EVR_REQUIRE(hp2 >= 1 && wp2 >= 1, "evr_lpips: image %dx%d too small for AlexNet", W, H);
    const int C[5] = {64, 192, 384, 256, 256};
    int rc;
    for (int l = 0; l < 5; ++l) if ((rc = dalloc(m, &m->feat[l], (size_t)n2 * m->h[l] * m->w[l] * C[l]))) return rc;
    if ((rc = dalloc(m, &m->pool1, (size_t)n2 * hp1 * wp1 * 64))) return rc;
    if ((rc = dalloc(m, &m->pool2, (size_t)n2 * hp2 * wp2 * 192))) return rc;
    if ((rc = dalloc(m, &m->partials, (size_t)5 * cap * SCORE_BLOCKS_MAX))) return rc;
    if ((rc = dalloc(m, &m->d_hw, 8))) return rc;
    if ((rc = dalloc(m, &m->d_args, (size_t)evr_lpips::NPLANS * 4))) return rc;
    int hw[5]; for (int l = 0; l < 5; ++l) hw[l] = m->h[l] * m->w[l];
    EVR_HIP(hipMemcpy(m->d_hw, hw, sizeof(hw), hipMemcpyHostToDevice));
    return EVR_OK;
}

// the plan of n image pairs inside the current buffers: cached, or built into the least recently used slot (its ConvArgs go to the
// device in stream order, behind whatever launch still reads the slot's previous contents)
static int lpips_plan(evr_lpips* m, int n, hipStream_t stream) {
    evr_lpips::Plan* pl = nullptr;
    for (auto& c : m->plans) if (c.n == n) pl = &c;
    if (pl) { pl->stamp = ++m->clock; m->cur = pl; m->n = n; return EVR_OK; }
    for (auto& c : m->plans) {
        if (c.n == 0) { pl = &c; break; }
        if (!pl || c.stamp < pl->stamp) pl = &c;
    }
    pl->n = 0;
    const int n2 = 2 * n;
    const int hp1 = m->h[1], wp1 = m->w[1], hp2 = m->h[2], wp2 = m->w[2];
    const float* ins[4] = {m->pool1, m->pool2, m->feat[2], m->feat[3]};
    const int hin[4] = {hp1, hp2, hp2, hp2}, win[4] = {wp1, wp2, wp2, wp2};
    for (int i = 0; i < 4; ++i) {
        Layer& L = m->L[i];
        ConvArgs& a = pl->args[i];
        memset(&a, 0, sizeof(a));
        a.in0 = ins[i]; a.c0 = L.cin; a.in_mode = IN_SINGLE; a.n = n2; a.hin = hin[i]; a.win = win[i];
        a.hm = hin[i]; a.wm = win[i]; a.stride = 1; a.os = 1; a.hout = hin[i]; a.wout = win[i];
        a.tp.ntaps = L.k * L.k; a.tp.ngroups = 1; a.tp.grp_cols = L.cout;
        for (int ky = 0; ky < L.k; ++ky) for (int kx = 0; kx < L.k; ++kx) a.tp.set_tap(ky * L.k + kx, ky - L.pad, kx - L.pad, 1);
        a.wgt = L.d_w; a.bias = L.d_b; a.cout = L.cout; a.n_valid = L.cout; a.out = m->feat[i + 1]; a.cout_total = L.cout;
        a.epi = EPI_BIAS_RELU; a.x3 = L.x3;
        a.wgt_wino = L.d_wino; set_wino_grid(a);
        {   // conv2 (5x5) on the band kernel with 8 KB of LDS padding (two blocks per CU instead of three); EVR_LPIPS_BAND5=<KB> sets the
            // padding, EVR_LPIPS_BAND5=-1 keeps the implicit GEMM (the default until the evaluation stream moved behind the residual
            // blocks: beside the ConvLSTM layers its LDS footprint cost more than the kernel gained; now +0.7 .. 1.0 %, profiles/r03_env_ab.txt)
            const char* e = getenv("EVR_LPIPS_BAND5");
            const int kb = e ? atoi(e) : 8;
            a.no_band5 = kb < 0 ? 1 : 0; a.band_lds_pad = kb < 0 ? 0 : kb * 1024;
        }
        a.acc_scale = (L.x3 == 3) ? std::ldexp(1.0f, -(L.mx_e + H2_ACT_EXP)) : 1.0f;
        a.in_packed = a.out_packed = L.x3 ? 1 : 0;      // pool1 / pool2 / feat1..feat4 are PACKED (H2 in mode 3) in the split modes
        a.mx_sa = 127 - MX_LO_EXP; a.mx_sb = 127 - L.mx_e; a.group_store = use_group_store(); set_fastdiv(a);
        pick_conv_tile(a, 32, &pl->wm[i], &pl->nb[i]);
    }
    {   // split-K partial sums of conv3..conv5 at small batches (conv.h KSPLIT_WS_BYTES): this handle's own -- it runs on the evaluation
        // stream beside a model's launches
        // (only when a conv of this plan can split -- a split arithmetic and <= 192 tiles of 128 pixels -- and a failed allocation
        // degrades to "never split" instead of failing the plan: ADVICE r5)
        bool may_split = false;
        for (int i = 1; i < 4; ++i) if (pl->args[i].x3 && (int64_t)n2 * pl->args[i].hm * pl->args[i].wm <= 192LL * 128) may_split = true;
        if (may_split && !m->kws && !m->kws_tried) {
            m->kws_tried = true;
            if (hipMalloc((void**)&m->kws, KSPLIT_WS_BYTES) != hipSuccess) { (void)hipGetLastError(); m->kws = nullptr; }
            else m->allocs.push_back((void*)m->kws);
        }
        for (int i = 0; i < 4; ++i) pl->args[i].ksplit_ws = may_split ? m->kws : nullptr;
    }
    EVR_HIP(hipMemcpyAsync(m->d_args + 4 * (pl - m->plans), pl->args, 4 * sizeof(ConvArgs), hipMemcpyHostToDevice, stream));
    pl->n = n; pl->stamp = ++m->clock; m->cur = pl; m->n = n;
    return EVR_OK;
}

extern "C" int evr_lpips_forward(evr_lpips* m, const float* img, const float* ref, int n, int H, int W, int clip, double* out,
                                 evr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EVR_REQUIRE(m && img && ref && out && n >= 1 && H >= 1 && W >= 1, "evr_lpips_forward: bad argument");
    int rc;
    if (m->H != H || m->W != W || n > m->cap) if ((rc = lpips_buffers(m, n, H, W, stream))) { m->release(); m->H = m->W = 0; return rc; }      // (a half-built set of buffers must not look like capacity)
    if (!m->cur || m->cur->n != n) if ((rc = lpips_plan(m, n, stream))) return rc;
    const evr_lpips::Plan& P = *m->cur;
    ConvArgs* const d_args = m->d_args + 4 * (m->cur - m->plans);
    const int n2 = 2 * n;
    Conv1Args c1{};
    c1.img = img; c1.ref = ref; c1.n = n; c1.H = H; c1.W = W; c1.h1 = m->h[0]; c1.w1 = m->w[0]; c1.clip = clip;
    c1.wg = m->d_wg; c1.wb = m->d_wb; c1.bias = m->d_b1; c1.bias_in = m->d_b1in; c1.out = m->feat[0];
    c1.wfrag = (const unsigned*)m->d_wfrag1; c1.wbsum = m->d_wbsum;
    static std::atomic<unsigned> attr_done[64];      // the LDS-size attribute is per device
    int dev = 0;
    EVR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_relaxed)) {
        EVR_HIP(hipFuncSetAttribute((const void*)lpips_conv1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EVR_HIP(hipFuncSetAttribute((const void*)lpips_conv1_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev >= 0 && dev < 64) attr_done[dev].store(1, std::memory_order_relaxed);
    }
    static const bool conv1_valu = getenv("EVR_LPIPS_CONV1_VALU") != nullptr;      // A/B switch: the direct VALU kernel in split mode too
    if (m->d_wfrag1 && !conv1_valu) {
        const size_t lds = (size_t)(71 * 71 + 3) * sizeof(float) + (size_t)8 * 2 * 2 * 64 * 16;
        hipLaunchKernelGGL(lpips_conv1_mfma_kernel, dim3((m->w[0] + 15) / 16, (m->h[0] + 15) / 16, n2), dim3(256), lds, stream, c1);
    } else {
        const size_t lds1 = (size_t)(2 * 71 * 71) * sizeof(float);
        hipLaunchKernelGGL(lpips_conv1_kernel, dim3((m->w[0] + 15) / 16, (m->h[0] + 15) / 16, n2), dim3(256), lds1, stream, c1);
    }
    EVR_LAUNCH_CHECK();
    const int pk = packed_fmt(m->L[0].x3);
    auto pool = [&](const float* in, float* o, int h, int w, int c, int ho, int wo, int in_pk) -> int {
        const int64_t total = (int64_t)n2 * ho * wo * (c / 4);
        hipLaunchKernelGGL(maxpool3s2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, o, n2, h, w, c, ho, wo, in_pk, pk);
        EVR_LAUNCH_CHECK();
        return EVR_OK;
    };
    // The layer scores read both images' features again: scored right behind the layer that wrote them (EVR_LPIPS_SCORE_SPLIT, default
    // 1) they come from the Infinity Cache -- 446 MB per 64-frame step that one launch at the end fetched from HBM (conv1's 180 MB of
    // features are long evicted by then).  Same partial sums, same final reduction: bit-identical scores.
    static const int score_split = getenv("EVR_LPIPS_SCORE_SPLIT") ? atoi(getenv("EVR_LPIPS_SCORE_SPLIT")) : 1;
    const int C[5] = {64, 192, 384, 256, 256};
    ScoreArgs sa;
    const int sblocks = score_blocks(n);
    for (int l = 0; l < 5; ++l) { sa.feat[l] = m->feat[l]; sa.lin[l] = m->d_lin[l]; sa.hw[l] = m->h[l] * m->w[l]; sa.C[l] = C[l]; sa.packed[l] = l > 0 ? pk : 0; }
    auto score = [&](int l0, int nl) -> int {
        hipLaunchKernelGGL(lpips_score_kernel, dim3(sblocks, n, nl), dim3(256), 0, stream, sa, n, m->partials, sblocks, l0);
        EVR_LAUNCH_CHECK();
        return EVR_OK;
    };
    if (score_split && (rc = score(0, 1))) return rc;
    if ((rc = pool(m->feat[0], m->pool1, m->h[0], m->w[0], 64, m->h[1], m->w[1], 0))) return rc;
    if ((rc = launch_conv_igemm(P.args[0], d_args + 0, 32, P.wm[0], P.nb[0], stream))) return rc;
    if (score_split && (rc = score(1, 1))) return rc;
    if ((rc = pool(m->feat[1], m->pool2, m->h[1], m->w[1], 192, m->h[2], m->w[2], pk))) return rc;
    // (relu3 .. relu5 are 69 MB together at 64 frames -- they outlive their convolutions in the Infinity Cache -- so ONE launch scores
    // the three of them behind conv5 (grid.z = layer): two launches fewer on the evaluation stream; EVR_LPIPS_SCORE_SPLIT=2: one each)
    for (int i = 1; i < 4; ++i) {
        if ((rc = launch_conv_igemm(P.args[i], d_args + i, 32, P.wm[i], P.nb[i], stream))) return rc;
        if (score_split == 2 && (rc = score(i + 1, 1))) return rc;
    }
    if (score_split == 1 && (rc = score(2, 3))) return rc;
    if (!score_split && (rc = score(0, 5))) return rc;
    hipLaunchKernelGGL(lpips_final_kernel, dim3(n), dim3(64), 0, stream, m->partials, out, sblocks, 5, n, m->d_hw);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

extern "C" double evr_lpips_flops(const evr_lpips* m) {
    if (!m || m->n == 0) return 0.0;
    const int cin[5] = {3, 64, 192, 384, 256}, cout[5] = {64, 192, 384, 256, 256}, ks[5] = {11, 5, 3, 3, 3};
    double f = 0.0;
    for (int l = 0; l < 5; ++l) f += 2.0 * (2.0 * m->n) * m->h[l] * m->w[l] * cin[l] * cout[l] * ks[l] * ks[l];
    return f;
}

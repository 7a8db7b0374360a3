// Convolution building blocks of the recurrent networks (NHWC activations on device).
//
// Activation formats.  PLAIN: fp32, pixel-major, channels contiguous.  PACKED (split mode, tensors that feed the
// matrix cores): the same 4 bytes per channel, laid out per group of 16 channels (64 B) as
//     16 x f16  hi  = RNE_f16(x)                                  (32 B)
//     16 x fp8  lo8 = RNE_e4m3((x - hi) * 2^12)                   (16 B)   value ~ hi + lo8 * 2^-12
//     16 x fp8  x8  = RNE_e4m3(x)                                 (16 B)   (only ever multiplied by a weight's low part)
// (fp8 = OCP e4m3fn, saturating at +-448; hi saturates at +-65504).  A 16-B slot of a pixel row is then directly an
// MFMA operand piece, so the consumer's main loop has no conversion work; the PRODUCER's epilogue does the split
// once per element instead of once per (tap, N tile).  Weight rows use the same 64-B groups per 16 k-values with
//     hi = RNE_f16(w),  w8 = RNE_e4m3(w * 2^e),  wlo8 = RNE_e4m3((w - hi) * 2^(e + 12)),  e chosen per tensor.
// Arithmetic per 32 k (conv.hip): acc += hi_x . hi_w  (two v_mfma_f32_32x32x16_f16) + 2^-(12+e) * (lo8 . w8 + x8 . wlo8)
// (one MX-scaled v_mfma_scale_f32_32x32x64_f8f6f4 whose K = 64 is [lo8 | x8] against [w8 | wlo8]; the block
// scales 2^-12 and 2^-e are the instruction's E8M0 operands) -- 2/3 of the matrix cycles of three bf16 products.
#pragma once
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <vector>

namespace evr {

constexpr int MAX_TAPS = 25;
constexpr int MAX_PHASES = 4;

enum InMode { IN_SINGLE = 0, IN_CAT = 1 };
enum Epilogue {
    EPI_BIAS = 0,           // out = acc + bias
    EPI_BIAS_RELU = 1,      // out = relu(acc + bias)
    EPI_RESIDUAL_RELU = 2,  // out = relu(acc + bias + residual)       (ResidualBlock, submodules.py:169-184)
    EPI_LSTM = 3,           // ConvLSTM gates (submodules.py:227-245); needs NB == 4 and permuted rows
    EPI_GRU_ZR = 4,         // ConvGRU update/reset (submodules.py:281-282): z -> aux0, h*r -> out
    EPI_GRU_OUT = 5,        // ConvGRU candidate + blend (submodules.py:283-285): h' in place in `state`
    EPI_BIAS_TANH = 6       // out = tanh(acc + bias)   (HyperE2VID bases_net, hyper_dynamic.py:41-48)
};

// Tap list of one convolution.  For ConvTranspose2d(k, stride 2) the four sub-pixel phases are column GROUPS of
// one GEMM (N = 4*Cout, phase-major): they share the input taps, and tap_groups[t] says which groups use tap t
// (the kernel skips the MFMA blocks of the others and whole K steps no group of its N tile needs).
struct ConvTaps {
    int ntaps;
    int tap[MAX_TAPS];          // (dy & 0xffff) | (dx << 16): dwords so the kernel fetches them with s_load
    int tap_groups[MAX_TAPS];   // bitmask over column groups
    int ngroups, grp_cols;      // columns per group (multiple of 32); ngroups*grp_cols == cout
    int grp_ofy[MAX_PHASES], grp_ofx[MAX_PHASES];   // output pixel = (my*os + ofy[g], mx*os + ofx[g])
    int inter;                  // 1: the four phases are interleaved per 128-column tile (round 5; grp_cols == 32, ngroups == 4): column
                                //   col = phase (col / 32) % 4, channel (col / 128) * 32 + col % 32 -- every tile then uses all nine taps
    void set_tap(int i, int dy, int dx, int groups) { tap[i] = (dy & 0xffff) | (dx * 65536); tap_groups[i] = groups; }
};

// One step of the programmed band kernel = one (tap, 32-channel chunk) whose weights are not structurally zero,
// packed into 32 bits (the kernel keeps the whole program in two VGPRs and picks entries with v_readlane: a scalar
// load per step would share lgkmcnt with the fragment ds_reads and force full waits):
//   bits 0-3   tap t = (dy+1)*3 + (dx+1)
//   bit  4     first step of its band (chunk, dy)
//   bit  5     a further band follows -> request it in this step (bits 16-27)
//   bit  6     that request may stay in flight past this step (the band has more steps)
//   bits 8-15  K chunk cc of the step: weight tile at float offset (t*nch2 + cc)*32 of a row
//   bits 16-25 next band's source chunk: py | px << 1 | channel chunk << 2
//   bits 26-27 next band's dy + 1
// Entry 0 describes the first band (bits 16-27 only); entries 1..prog_steps are the steps.
constexpr int BAND_PROG_MAX = 128;

struct ConvArgs {
    const float* in0; const float* in1;
    int c0, c1;               // channels of in0/in1 (IN_SINGLE: c1 == 0)
    int in_mode;
    int n, hin, win;          // input tensor [n, hin, win, c]
    int hm, wm;               // M-grid per image; GEMM M = n*hm*wm
    int stride;               // input pixel = m*stride + tap offset
    ConvTaps tp;
    const float* wgt;         // [cout][ntaps*(cin_total)], K contiguous (zero blocks for unused (tap, group) pairs)
    const float* bias;        // [cout]
    int cout;                 // GEMM N (multiple of 32*NB; rows >= n_valid are zero padding)
    int n_valid;              // real output channels
    float* out; int hout, wout, cout_total, os;   // NHWC output [n, hout, wout, cout_total]
    int epi;
    const float* residual;    // EPI_RESIDUAL_RELU: same shape as out
    const float* post_add;    // optional, plain epilogues: out = f(acc) + post_add (fused skip_sum)
    float* state;             // EPI_LSTM: cell state (in place); EPI_GRU_*: hidden state h
    float* aux0;              // EPI_GRU_ZR: z out; EPI_GRU_OUT: z in
    int hidden;               // EPI_LSTM / GRU: number of hidden channels C
    // fused prediction layer (model/unet.py:136-138): when pred_w != null and the GEMM has ONE N tile, the
    // epilogue reduces (value [+ post_add]) . pred_w over the channels, adds pred_b, applies the final
    // activation and writes the centre-cropped pixel to the image passed at launch; `out` may then be null.
    const float* pred_w; float pred_b; int pred_sigmoid;
    const float* pred_skip_dot;   // optional [n, hout, wout]: sum_c pred_w[c] * skip[c] computed by the skip's producer (then post_add is null)
    int crop_h, crop_w, crop_y0, crop_x0;
    int x3;                   // arithmetic mode (arith_mode()): 2 / 3 = weights in that mode's split layout and in0/in1 PACKED / H2
    int in_packed;            //   tensors: the main loop feeds LDS slots straight to the MFMAs; 0 = fp32 MFMA on PLAIN tensors
    float acc_scale;          // mode 3: accumulators (which start at bias / acc_scale) are multiplied by this 2^-(e_w + H2_ACT_EXP)
    unsigned div_hw_mul, div_hw_sh, div_w_mul, div_w_sh;   // m / (hm*wm) and r / wm by multiply-high (set_fastdiv)
    int group_store;          // PACKED outputs as whole 64-B groups after a lane exchange (packed.h xchg16); 0: 4-channel pieces
    int mx_sa, mx_sb;         // E8M0 block scales of the fp8 correction MFMA: 127 - 12 (activations), 127 - e (weights)
    int out_packed;           // write `out` PACKED (n_valid and cout_total multiples of 8)
    int res_packed, padd_packed, state_packed;   // format of residual / post_add / the ConvGRU hidden state
    // space-to-depth form of a k5 stride-2 convolution for the programmed band kernel (conv.hip): the input seen as
    // [n, hin/2, win/2, 4*c0] (2x2 pixel blocks -> channels, phase-major) makes it a 3x3 stride-1 convolution whose
    // unused (tap, phase) chunks are simply absent from the step program
    const float* wgt2;        // [cout][9][4*c0] weights of that form (same split packing), or null
    const unsigned* prog;     // device: BAND_PROG_MAX packed entries (see above)
    int prog_steps;
    int debug_ablate;         // timing ablation only (EVR_ABLATE env): bit0 skip barriers, bit1 skip DMA, bit2 skip epilogue math
    float* prev_rec;          // optional [n,1,hout,wout]: the un-cropped prediction (E2VIDRecurrent.prev_recs, model.py:143)
    unsigned* sat;            // optional device counter: output runs beyond the packed format's exact range (packed.h sat_note)
    int band_lds_pad;         // conv_bandk_kernel: extra dynamic LDS bytes per block (caps the blocks per CU; host-side only)
    int no_band5;             // keep a 5x5 stride-1 convolution on the implicit GEMM (LPIPS conv2: on the evaluation stream the band form's
                              //   three 51-KB blocks per CU crowd the reconstruction stream's work-groups out: 6.0k vs 6.8k frames/s)
    float* ksplit_ws;         // split-K partial sums (conv.hip launch_band / launch_band_prog): KSPLIT_WS_BYTES of device memory owned by the
                              //   handle whose launches use it (evr_model per shape, evr_lpips per plan) -- launches of ONE handle are ordered
                              //   on one stream, so they can share it; null: never split.  Host-side only.
    // Winograd F(2x2, 3x3) form of a 3x3 stride-1 convolution in the exact-fp32 mode (wino.hip): transform-domain weights in the
    // kernel's streaming order, or null; the grid of 2x2 output tiles per image and its multiply-high divisors
    const float* wgt_wino;
    int wino_th, wino_tw;
    unsigned wdiv_t_mul, wdiv_t_sh, wdiv_tw_mul, wdiv_tw_sh;
    int wino_s2d;             // 0, or 1 + log2(cin / 8): a k5 stride-2 convolution on the Winograd kernel in space-to-depth form (wgt_wino from its s2d weights)
    int wino_order;           // 1: items of a launch ordered 8 tile blocks x 4 column blocks per XCD round (wino.hip); 0: column block fastest
};
// A split launch has at most 512 blocks (one per resident slot) of 64 KB of partial accumulators each
constexpr size_t KSPLIT_WS_BYTES = (size_t)512 * 4 * 16 * 64 * 16;

// Division of n < 2^31 by an invariant d >= 1 as (umulhi(n, mul) + n) >> sh (Granlund-Montgomery, round-up form):
// sh = ceil(log2 d), mul = floor(2^32 (2^sh - d) / d) + 1 (0 for powers of two).  A run-time v_udiv costs ~35 VALU
// instructions; the short-K kernels decode a pixel index per block with two of them.
inline void fastdiv_magic(unsigned d, unsigned* mul, unsigned* sh) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *sh = s;
    *mul = (unsigned)((((1ull << s) - d) << 32) / d + 1ull);
    if (((1ull << s) - d) == 0) *mul = 0;
}
inline void set_fastdiv(ConvArgs& a) {
    fastdiv_magic((unsigned)(a.hm * a.wm), &a.div_hw_mul, &a.div_hw_sh);
    fastdiv_magic((unsigned)a.wm, &a.div_w_mul, &a.div_w_sh);
}

// fp32 -> bf16 bits, round to nearest even (head_mfma_kernel's weight fragments)
inline unsigned short bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
inline float bf16_to_f32(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

// fp32 -> IEEE half bits, round to nearest even, saturating at +-65504 (what v_cvt_f16_f32 gives after a clamp)
inline unsigned short f16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    const unsigned short sign = (unsigned short)((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return (unsigned short)(sign | 0x7e00u);                       // NaN
    float af; memcpy(&af, &u, 4);
    if (af >= 65504.0f) return (unsigned short)(sign | 0x7bffu);
    if (u < 0x38800000u) {                                                             // below 2^-14: half subnormals, step 2^-24
        const float scaled = af * 16777216.0f;
        const float r = __builtin_nearbyintf(scaled);                                  // RNE in the default rounding mode
        return (unsigned short)(sign | (unsigned)r);                                   // 1024 = smallest normal: still right
    }
    unsigned hbits = ((((u >> 23) - 112u) << 10) | ((u & 0x7fffffu) >> 13));
    const unsigned rem = u & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (hbits & 1u))) ++hbits;
    return (unsigned short)(sign | hbits);
}
inline float f16_to_f32(unsigned short hb) {
    const unsigned sign = ((unsigned)hb & 0x8000u) << 16, e = (hb >> 10) & 31u, m = hb & 0x3ffu;
    float f;
    if (e == 0) { f = (float)m * (1.0f / 16777216.0f); unsigned u; memcpy(&u, &f, 4); u |= sign; memcpy(&f, &u, 4); return f; }
    const unsigned u = sign | (e == 31 ? 0x7f800000u : ((e + 112u) << 23)) | (m << 13);
    memcpy(&f, &u, 4);
    return f;
}
// fp32 -> OCP e4m3fn bits (bias 7, 3 mantissa bits, no inf, max 448), round to nearest even, saturating
inline unsigned char e4m3_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    const unsigned char sign = (unsigned char)((u >> 24) & 0x80u);
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return (unsigned char)(sign | 0x7fu);                          // NaN
    float af; memcpy(&af, &u, 4);
    if (af >= 448.0f) return (unsigned char)(sign | 0x7eu);
    if (af < 0.015625f) return (unsigned char)(sign | (unsigned)__builtin_nearbyintf(af * 512.0f));   // subnormals, step 2^-9
    unsigned b = ((((u >> 23) - 120u) << 3) | ((u & 0x7fffffu) >> 20));
    const unsigned rem = u & 0xfffffu;
    if (rem > 0x80000u || (rem == 0x80000u && (b & 1u))) ++b;
    if (b > 0x7eu) b = 0x7eu;
    return (unsigned char)(sign | b);
}
inline float e4m3_to_f32(unsigned char b) {
    const unsigned e = (b >> 3) & 15u, m = b & 7u;
    float f = (e == 0) ? (float)m * (1.0f / 512.0f) : __builtin_ldexpf(1.0f + (float)m * 0.125f, (int)e - 7);
    return (b & 0x80u) ? -f : f;
}
inline float clampf(float v, float lim) { return v > lim ? lim : (v < -lim ? -lim : v); }

constexpr int MX_LO_EXP = 12;     // lo8 = (x - hi) * 2^12

// One 16-value group -> the 64-B PACKED group.  `e8` scales the plain fp8 copy (activations: 0), `elo` the residual.
inline void pack_group16(const float* v, int e8, int elo, unsigned char* dst) {
    unsigned short hi[16]; unsigned char lo8[16], x8[16];
    for (int k = 0; k < 16; ++k) {
        const float c = clampf(v[k], 65504.0f);
        hi[k] = f16_rne(c);
        lo8[k] = e4m3_rne(clampf(__builtin_ldexpf(c - f16_to_f32(hi[k]), elo), 448.0f));
        x8[k] = e4m3_rne(clampf(__builtin_ldexpf(v[k], e8), 448.0f));
    }
    memcpy(dst, hi, 32);
    memcpy(dst + 32, lo8, 16);
    memcpy(dst + 48, x8, 16);
}
// PACKED activation codec on the host (tests; the device twin is packed.h)
inline void pack_split_act(std::vector<float>& x) {
    for (size_t base = 0; base + 16 <= x.size(); base += 16) {
        unsigned char g[64];
        // activations: [hi | lo8 = (x - hi) 2^12 | x8 = x]
        pack_group16(&x[base], 0, MX_LO_EXP, g);
        memcpy(&x[base], g, 64);
    }
}
inline void unpack_split_act(const float* src, float* dst, size_t n) {
    for (size_t base = 0; base + 16 <= n; base += 16) {
        unsigned char g[64]; memcpy(g, src + base, 64);
        unsigned short hi[16]; memcpy(hi, g, 32);
        for (int k = 0; k < 16; ++k) dst[base + k] = f16_to_f32(hi[k]) + __builtin_ldexpf(e4m3_to_f32(g[32 + k]), -MX_LO_EXP);
    }
}
// Split weight packing (host side, at model creation): every aligned group of 16 k-values of a row becomes
// [16 f16 hi | 16 fp8 w8 = w 2^e | 16 fp8 wlo8 = (w - hi) 2^(e+12)] -- note the fp8 pieces are in the order that pairs
// them with the activation group's [lo8 | x8].  Returns e: the largest exponent that keeps both fp8 pieces in range
// (|w| 2^e <= 224; |w - hi| <= |w| 2^-11 keeps the residual below 448 too).
inline int pack_split_weights(std::vector<float>& w) {
    float mx = 0.f;
    for (float v : w) { const float a = v < 0 ? -v : v; if (a == a && a > mx) mx = a; }
    int e = 0;
    if (mx > 0.f) { int ex; (void)__builtin_frexpf(224.0f / mx, &ex); e = ex - 1; }    // 2^e <= 224/mx
    if (e > 24) e = 24;
    if (e < -24) e = -24;
    for (size_t base = 0; base + 16 <= w.size(); base += 16) {
        unsigned char g[64];
        pack_group16(&w[base], e, e + MX_LO_EXP, g);
        // pack_group16 wrote [hi | residual | plain]; weights want [hi | plain (w8) | residual (wlo8)]
        unsigned char t[16]; memcpy(t, g + 32, 16); memcpy(g + 32, g + 48, 16); memcpy(g + 48, t, 16);
        memcpy(&w[base], g, 64);
    }
    return e;
}
// ---- H2: the second PACKED format (fp32-grade arithmetic mode).  Per 16 values 16 f16 hi | 16 f16 lo of v * 2^e:
// hi = RNE_f16(v 2^e), lo = RNE_f16(v 2^e - hi) -> 22 significant bits while lo stays a normal half (|v 2^e| >= 2^-3).
// Activations use the fixed exponent H2_ACT_EXP (range +-4094; full precision from 2^-7 up, 2^-29 absolute below);
// weights a per-tensor exponent that brings max|w| to [2^13, 2^14).  conv.hip multiplies hi_w hi_x + hi_w lo_x + lo_w hi_x
// on three v_mfma_f32_32x32x16_f16 per 16 k (the dropped lo lo term is 2^-22 of the product), fp32 accumulation of
// products scaled by 2^(e_w + H2_ACT_EXP); the epilogue multiplies by ConvArgs::acc_scale = 2^-(e_w + H2_ACT_EXP).
constexpr int H2_ACT_EXP = 4;
inline void pack_group16_h2(const float* v, int e, unsigned char* dst) {
    unsigned short hi[16], lo[16];
    for (int k = 0; k < 16; ++k) {
        const float c = clampf(__builtin_ldexpf(v[k], e), 65504.0f);
        hi[k] = f16_rne(c);
        lo[k] = f16_rne(c - f16_to_f32(hi[k]));
    }
    memcpy(dst, hi, 32);
    memcpy(dst + 32, lo, 32);
}
inline void pack_h2_act(std::vector<float>& x) {
    for (size_t base = 0; base + 16 <= x.size(); base += 16) {
        unsigned char g[64];
        pack_group16_h2(&x[base], H2_ACT_EXP, g);
        memcpy(&x[base], g, 64);
    }
}
inline void unpack_h2(const float* src, float* dst, size_t n, int e) {
    for (size_t base = 0; base + 16 <= n; base += 16) {
        unsigned short g[32]; memcpy(g, src + base, 64);
        for (int k = 0; k < 16; ++k) dst[base + k] = __builtin_ldexpf(f16_to_f32(g[k]) + f16_to_f32(g[16 + k]), -e);
    }
}
// weights -> H2 groups; returns the exponent e (max|w| 2^e in [2^13, 2^14))
inline int pack_h2_weights(std::vector<float>& w) {
    float mx = 0.f;
    for (float v : w) { const float a = v < 0 ? -v : v; if (a == a && a > mx) mx = a; }
    int e = 0;
    if (mx > 0.f) { int ex; (void)__builtin_frexpf(16384.0f / mx, &ex); e = ex - 1; if (__builtin_ldexpf(mx, e) >= 16384.0f) --e; }
    if (e > 40) e = 40;
    if (e < -40) e = -40;
    for (size_t base = 0; base + 16 <= w.size(); base += 16) {
        unsigned char g[64];
        pack_group16_h2(&w[base], e, g);
        memcpy(&w[base], g, 64);
    }
    return e;
}
// ---- P6: the third PACKED format (f16 + MX-fp6 arithmetic, EVR_ARITH=mx6).  Per 16 values (64 B):
//   bytes  0-31  16 f16 hi = RNE_f16(v) (saturating)
//   bytes 32-55  32 e2m3 codes (6 bits each, element j at bits 6j): element 2i = v_i / S, element 2i + 1 = (v_i - hi_i) 2^11 / S
//                with the GROUP's own scale S = 2^(E - 2), E = floor(log2 max|v|): the largest value lands in [4, 8) (e2m3
//                saturates at 7.5), a residual -- at most 2^-11 of its value -- below that
//   byte  56     the E8M0 scale byte (biased exponent of S, at least 1; weights: of S 2^-11), bytes 57-63 zero
// conv.hip multiplies hi hi on v_mfma_f32_32x32x16_f16 and the two cross terms of 32 channels on ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 operands -- 8 passes instead of the 16 of the fp8 form -- whose per-lane scale
// registers are these bytes: a lane half holds one group, i.e. its 32 k-values [v_0/S, lo_0, v_1/S, lo_1, ...] against the
// weight group's [wlo_0, w_0/S, ...] (note the swapped order).  Weights are pre-multiplied by 2^e (max|w| 2^e in [2^13, 2^14),
// as for H2) so that small weights keep a normal f16 half; the epilogue multiplies by ConvArgs::acc_scale = 2^-e.
constexpr int P6_LO_EXP = 11;
// fp32 -> e2m3 code (bias 1, 3 mantissa bits, no inf/NaN, max 7.5): round to nearest even, saturating -- what
// v_cvt_scalef32_2xpk16_fp6_f32 does (tools/fp6_cvt_probe.hip)
inline unsigned char e2m3_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    const unsigned char sign = (unsigned char)((u >> 26) & 0x20u);
    u &= 0x7fffffffu;
    float af; memcpy(&af, &u, 4);
    if (!(af == af)) return (unsigned char)(sign | 0x1fu);
    if (af >= 7.5f) return (unsigned char)(sign | 0x1fu);
    float q;            // the value in units of its binade's step
    unsigned base;
    if (af < 1.0f) { q = af * 8.0f; base = 0; }                               // subnormals and the first binade share step 1/8
    else if (af < 2.0f) { q = (af - 1.0f) * 8.0f; base = 8; }
    else if (af < 4.0f) { q = (af - 2.0f) * 4.0f; base = 16; }
    else { q = (af - 4.0f) * 2.0f; base = 24; }
    unsigned code = base + (unsigned)__builtin_nearbyintf(q);                  // RNE; a carry walks into the next binade's code
    if (code > 0x1fu) code = 0x1fu;
    return (unsigned char)(sign | code);
}
inline float e2m3_to_f32(unsigned char c) {
    const unsigned m = c & 31u;
    const float f = m < 8u ? (float)m * 0.125f : __builtin_ldexpf((float)(8u | (m & 7u)), (int)(m >> 3) - 4);
    return (c & 32u) ? -f : f;
}
// (v scaled by 2^e first; `weights`: elements swapped and the scale byte lowered by P6_LO_EXP)
inline void pack_group16_p6(const float* v, int e, bool weights, unsigned char* dst) {
    unsigned short hi[16]; float c[16], lo[16]; float mx = 0.f;
    for (int k = 0; k < 16; ++k) {
        c[k] = clampf(__builtin_ldexpf(v[k], e), 65504.0f);
        if (!(c[k] == c[k])) c[k] = 0.f;
        hi[k] = f16_rne(c[k]);
        lo[k] = __builtin_ldexpf(c[k] - f16_to_f32(hi[k]), P6_LO_EXP);
        const float a = c[k] < 0 ? -c[k] : c[k];
        if (a > mx) mx = a;
    }
    unsigned mu; memcpy(&mu, &mx, 4);
    int eb = (int)(mu >> 23);
    eb = eb > 3 ? eb - 2 : 1;
    unsigned char codes[32];
    for (int k = 0; k < 16; ++k) {
        const unsigned char cv = e2m3_rne(__builtin_ldexpf(c[k], 127 - eb)), cl = e2m3_rne(__builtin_ldexpf(lo[k], 127 - eb));
        codes[2 * k] = weights ? cl : cv; codes[2 * k + 1] = weights ? cv : cl;
    }
    memset(dst + 32, 0, 32);
    for (int j = 0; j < 32; ++j) {
        const int bit = 6 * j;
        const unsigned v16 = (unsigned)codes[j] << (bit & 7);
        dst[32 + (bit >> 3)] |= (unsigned char)v16;
        if ((bit & 7) > 2) dst[32 + (bit >> 3) + 1] |= (unsigned char)(v16 >> 8);
    }
    int sb = weights ? eb - P6_LO_EXP : eb;
    if (sb < 1) sb = 1;
    dst[56] = (unsigned char)sb;
    memcpy(dst, hi, 32);
}
inline void pack_p6_act(std::vector<float>& x) {
    for (size_t base = 0; base + 16 <= x.size(); base += 16) {
        unsigned char g[64];
        pack_group16_p6(&x[base], 0, false, g);
        memcpy(&x[base], g, 64);
    }
}
inline void unpack_p6(const float* src, float* dst, size_t n) {      // activations (hi + residual)
    for (size_t base = 0; base + 16 <= n; base += 16) {
        unsigned char g[64]; memcpy(g, src + base, 64);
        unsigned short hi[16]; memcpy(hi, g, 32);
        for (int k = 0; k < 16; ++k) {
            const int bit = 6 * (2 * k + 1);
            const unsigned w = (unsigned)g[32 + (bit >> 3)] | ((unsigned)g[32 + (bit >> 3) + 1] << 8);
            dst[base + k] = f16_to_f32(hi[k]) + __builtin_ldexpf(e2m3_to_f32((unsigned char)((w >> (bit & 7)) & 63u)), (int)g[56] - 127 - P6_LO_EXP);
        }
    }
}
// weights -> P6 groups; returns the exponent e (max|w| 2^e in [2^13, 2^14))
inline int pack_p6_weights(std::vector<float>& w) {
    float mx = 0.f;
    for (float v : w) { const float a = v < 0 ? -v : v; if (a == a && a > mx) mx = a; }
    int e = 0;
    if (mx > 0.f) { int ex; (void)__builtin_frexpf(16384.0f / mx, &ex); e = ex - 1; if (__builtin_ldexpf(mx, e) >= 16384.0f) --e; }
    if (e > 40) e = 40;
    if (e < -40) e = -40;
    for (size_t base = 0; base + 16 <= w.size(); base += 16) {
        unsigned char g[64];
        pack_group16_p6(&w[base], e, true, g);
        memcpy(&w[base], g, 64);
    }
    return e;
}
// arithmetic mode of the 32-channel-chunk convolutions (ConvArgs::x3):
//   3  three f16 products on H2 tensors, fp32-grade ("h3"; the DEFAULT since round 4: the reference computes in fp32, and this is the
//      fastest arithmetic here whose image error stays at the level of an fp32 summation-order change, <= 1e-6);
//   4  split f16 + MX-fp6 corrections on P6 tensors (EVR_ARITH=mx6, the opt-in fast mode: 1.4e-5 on the image) -- for layouts whose packed
//      tensors are all written as whole groups by matrix-core epilogues; evr_model_create narrows it to mode 2 for the others, LPIPS then runs mode 2;
//   2  split f16 + MX-fp8 corrections on PACKED tensors (EVR_ARITH=mx: everywhere);
//   0  exact fp32 MFMA on PLAIN tensors (EVR_FP32=1 or EVR_ARITH=fp32)
inline int arith_mode() {
    if (getenv("EVR_FP32")) return 0;
    const char* e = getenv("EVR_ARITH");
    if (!e || !*e || !strcmp(e, "h3")) return 3;
    if (!strcmp(e, "mx6")) return 4;
    if (!strcmp(e, "mx")) return 2;
    if (!strcmp(e, "fp32")) return 0;
    return 3;
}
inline bool use_split_mode() { return arith_mode() != 0; }
// value of the `packed` flags for tensors of a mode: 0 PLAIN, 1 PACKED (f16 | fp8 | fp8), 2 H2, 3 P6
inline int packed_fmt(int mode) { return mode == 3 ? 2 : (mode == 2 ? 1 : (mode == 4 ? 3 : 0)); }
// weights (K contiguous, multiples of 16) -> the mode's split layout in place; returns the tensor exponent
inline int pack_weights_for(int mode, std::vector<float>& w) { return mode == 3 ? pack_h2_weights(w) : (mode == 4 ? pack_p6_weights(w) : pack_split_weights(w)); }
// (A/B switch for the whole-group PACKED stores)
inline int use_group_store() { const char* e = getenv("EVR_GROUP_STORE"); return e ? atoi(e) : 1; }

// kc: K chunk (16 or 32 channels); wm: waves per block along M (1,2,4); nb: 32-column blocks per wave (1,2,4).
// `a` is the host copy (grid sizing, validation); `d_args` the same plan resident in device memory (the
// kernel reads it with scalar loads; it is uploaded once per shape, not per launch).
int launch_conv_igemm(const ConvArgs& a, const ConvArgs* d_args, int kc, int wm, int nb, hipStream_t stream, float* img = nullptr);
// (conv.hip is compiled once per split arithmetic: mode 2 + the fp32 kernels, mode 3, mode 4; launch_conv_igemm dispatches on a.x3)
int launch_conv_igemm_mx(const ConvArgs& a, const ConvArgs* d_args, int kc, int wm, int nb, hipStream_t stream, float* img);
int launch_conv_igemm_h3(const ConvArgs& a, const ConvArgs* d_args, int kc, int wm, int nb, hipStream_t stream, float* img);
int launch_conv_igemm_m6(const ConvArgs& a, const ConvArgs* d_args, int kc, int wm, int nb, hipStream_t stream, float* img);
// Winograd F(2x2, 3x3) path of the exact-fp32 mode (wino.hip; EVR_WINO=0: never): host-side weight transform (w = [n_gemm][9][cin]
// in prep_conv2d's order; lstm_hidden > 0: ConvLSTM row permutation), the eligibility test of a launch plan, the launch
void wino_pack_weights(const std::vector<float>& w, int n_gemm, int cin, int lstm_hidden, std::vector<float>& out);
bool wino_enabled();
bool wino_eligible(const ConvArgs& a);
int launch_conv_wino(const ConvArgs& a, const ConvArgs* d_args, hipStream_t stream, float* img = nullptr);
inline void set_wino_grid(ConvArgs& a) {
    // (wino_s2d: a k5 stride-2 convolution read in space-to-depth form -- a 3x3 stride-1 convolution over 2x2 pixel blocks with 4 cin
    // channels: the tiles live on the block grid hm x wm)
    const int gh = a.wino_s2d ? a.hm : a.hin, gw = a.wino_s2d ? a.wm : a.win;
    a.wino_th = (gh + 1) / 2; a.wino_tw = (gw + 1) / 2;
    fastdiv_magic((unsigned)(a.wino_th * a.wino_tw), &a.wdiv_t_mul, &a.wdiv_t_sh);
    fastdiv_magic((unsigned)a.wino_tw, &a.wdiv_tw_mul, &a.wdiv_tw_sh);
    a.wino_order = getenv("EVR_WINO_ORDER") ? atoi(getenv("EVR_WINO_ORDER")) : 1;
}
// picks (wm, nb) for the shape: fills the 256 CUs when M is small
void pick_conv_tile(const ConvArgs& a, int kc, int* wm, int* nb);

// Direct small-Cin head convolution: planar vox [n,B,H,W] (unpadded; optional on-the-fly event-tensor
// normalization, eval.py:398-410) -> zero pad to (hp,wp) -> conv kxk s1 p k/2 -> bias -> relu -> NHWC [n,hp,wp,cout].
struct HeadArgs {
    const float* vox; const double* stats;   // stats: [n,3] {sum,sumsq,nnz} or null
    int n, B, H, W, hp, wp, pad_top, pad_left;
    int k, cout;
    const float* wgt;    // [B*k*k][cout]
    const float* bias;   // [cout]
    float* out;
    int relu;
    int out_packed;      // write `out` in the PACKED activation format
    int group_store;     //   as whole 64-B groups (see ConvArgs)
    const void* wfrag;   // k5/32-channel matrix-core form: weights in MFMA-fragment order (head_mfma_kernel), or null
    float wfrag_scale, wfrag_inv_scale;   //   2^-e_w and its inverse (head_pack_wfrag returns e_w; the input is used unscaled)
    // head_mfma_kernel only: the prediction layer's skip term of every pixel, sum_c pred_w[c] * out[c] (fp32, before the
    // PACKED rounding), so the last decoder reads one float per pixel instead of the 32 channels (ConvArgs::pred_skip_dot)
    const float* pred_w; float* pred_dot;   // [32] / [n, hp, wp], or null
    unsigned* sat;       // optional device counter of output runs beyond the packed format's exact range (packed.h sat_note)
};
// weights [B*k*k][32] (k = 5, B bins) -> the fragment-order table head_mfma_kernel reads (10 slabs x {hi, lo} x 64 lanes x 16 B);
// returns the weight exponent e_w
int head_pack_wfrag(const float* w, int B, std::vector<unsigned>& out);
int launch_head_conv(const HeadArgs& a, hipStream_t stream);

// Prediction layer: 1x1 conv C->1 on (x [+ skip]) + bias [+ sigmoid], centre crop -> planar img [n,1,H,W].
struct PredArgs {
    const float* x; const float* skip;   // NHWC [n,hp,wp,c]; skip may be null
    int n, hp, wp, c;
    const float* wgt; float bias;        // BN folded
    int sigmoid;
    int H, W, iy0, ix0;                  // crop window
    float* img;
    float* prev_rec;                     // optional un-cropped copy [n,1,hp,wp]
    int x_packed, skip_packed;           // activation formats
};
int launch_pred(const PredArgs& a, hipStream_t stream);

// HyperE2VID context (hyper_dynamic.py:19-23): cat(padded [normalized] event tensor, prev_rec) -> bilinear x1/4
// (align_corners=False: the mean of the 2x2 block at rows/cols 4o+1..4o+2) -> planar [n, B+1, hp/4, wp/4].
struct CtxArgs {
    const float* vox; const double* stats; const float* prev_rec;
    int n, B, H, W, hp, wp, pad_top, pad_left;
    float* out;
};
int launch_ctx_down(const CtxArgs& a, hipStream_t stream);
// HyperE2VID per-pixel dynamic filtering (hyper_dynamic.py:50-57,83-88): atoms = coeff[6,12] x bases[12,25];
// out[pix][c*6+m] = sum_l atoms[m][l] * x[pix + offset(l)][c] over the 5x5 neighbourhood (zero padded).
// (out_fmt: 0 PLAIN, 1 PACKED, 2 H2 -- the format of `out`, written directly; P6 has no 4-channel writer)
int launch_dynamic_filter(const float* x, const float* coeff, const float* bases, float* out, int n, int h, int w,
                          int c, hipStream_t stream, int out_fmt = 0);

// Bilinear x2 (align_corners=False) of (x + skip): NHWC [n,h,w,c] -> [n,2h,2w,c]  (submodules.py:88)
// (x_packed / skip_packed / out_packed: tensor formats)
int launch_upsample2x_sum(const float* x, const float* skip, float* out, int n, int h, int w, int c, int x_packed, int skip_packed, int out_packed, hipStream_t stream);
// out = x + y (skip_sum, model_util.py:4-5) when it cannot be fused into a producer epilogue
// (packed: all three tensors are PACKED)
int launch_add(const float* x, const float* y, float* out, int64_t n, int packed, hipStream_t stream);
// PLAIN -> PACKED (in place allowed), n a multiple of 16: the output of a VALU kernel that feeds a matrix-core convolution
int launch_to_packed(const float* src, float* dst, int64_t n, hipStream_t stream, int fmt = 1);
// SPADE-E2VID helpers (spade.hip; model/spade_e2v.py of the reference)
struct SpadePredArgs {
    const float* x; const float* head;   // NHWC [n,hp,wp,32]
    int x_packed, head_packed;
    int n, hp, wp;
    const float* wgt;                    // device [3][32], bn_img folded
    float bias[3];
    float* prev;                         // [n,3,hp,wp] planar: prev_recs / the next frame's segmentation map
    float* img; int H, W, iy0, ix0;      // cropped mean image [n,1,H,W]
};
int launch_spade_pad(const float* vox, float* xpad, int n, int B, int H, int W, int hp, int wp, int pad_top, int pad_left, hipStream_t stream);
int launch_spade_first(float* xpad, float* xorg, int n, int B, int hp, int wp, hipStream_t stream);
int launch_nearest_half(const float* in, float* out, int planes, int h, int w, hipStream_t stream);
int launch_spade_apply(const float* xn, const float* gb, const float* skip, float* out, int64_t pix, int C, int skip_packed, int out_packed, hipStream_t stream);
int launch_spade_pred(const SpadePredArgs& a, hipStream_t stream);
// ET-Net token kernels (etnet.hip; model/eitr of the reference)
struct AttnArgs {
    const float* q; const float* k; const float* v;   // PLAIN fp32 rows; head h = columns [off + 32h, off + 32h + 32)
    int ldq, ldk, ldv, qo, ko, vo;
    int n, Lq, Lk, heads;
    float* out; int out_packed;                       // [n, Lq, 256]
};
int launch_layernorm256(const float* x, const float* w, const float* b, float* out, int64_t rows, int out_packed, hipStream_t stream);
int launch_attention(const AttnArgs& a, hipStream_t stream);
int launch_add_pos(const float* x, const float* pos, float* out, int n, int L, int x_packed, hipStream_t stream);
int launch_mean6(const float* const* in, float* out, int64_t rows, int out_packed, hipStream_t stream);
// InstanceNorm2d of the norm='IN' ResidualBlocks (spade.hip): out = relu(IN(x) [+ res]) [+ skip]
int launch_instnorm(const float* x, const float* res, const float* skip, float* out, int n, int hw, int c, int res_packed,
                    int skip_packed, int out_packed, hipStream_t stream);
// NHWC -> NCHW copy (debug/parity reads)
// (c_stride: channels per pixel row of `src` when only its first c are wanted; 0 = c)
int launch_nhwc_to_nchw(const float* src, float* dst, int n, int h, int w, int c, int packed, hipStream_t stream, int c_stride = 0);

}  // namespace evr

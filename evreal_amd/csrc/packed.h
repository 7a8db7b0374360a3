// Device helpers for the packed activation formats (conv.h).  PACKED: per 16 channels 16 f16 'hi' | 16 fp8 'lo8' | 16 fp8 'x8';
// H2 and P6 further down.
#pragma once
#include <hip/hip_runtime.h>

namespace evr {

#if defined(__HIP_DEVICE_COMPILE__)
// two fp32 -> packed bf16 (RNE): `lo` lands in bits 15:0, `hi` in bits 31:16 (no builtin on gfx950); head_mfma_kernel only
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
constexpr float PK_LO_SCALE = 4096.0f, PK_LO_INV = 1.0f / 4096.0f;      // 2^12 (conv.h MX_LO_EXP)

// 4 consecutive channels c4..c4+3 (c4 % 4 == 0) of a pixel row: an 8-B hi piece, a 4-B lo8 piece and a 4-B x8 piece
// inside the row's 64-B group c4 / 16.  Offsets in float elements from the tensor base.
__device__ __forceinline__ unsigned pk_off(unsigned row_off, int c4) { return row_off + (unsigned)((c4 & ~15) + ((c4 & 15) >> 1)); }
__device__ __forceinline__ unsigned pk_lo_off(unsigned row_off, int c4) { return row_off + (unsigned)((c4 & ~15) + 8 + ((c4 & 15) >> 2)); }
__device__ __forceinline__ f4 unpack4(uint2 hi, unsigned lo8) {
    const h2_t h0 = __builtin_bit_cast(h2_t, hi.x), h1 = __builtin_bit_cast(h2_t, hi.y);
    f4 v;
    v[0] = fmaf(__builtin_amdgcn_cvt_f32_fp8((int)lo8, 0), PK_LO_INV, (float)h0[0]);
    v[1] = fmaf(__builtin_amdgcn_cvt_f32_fp8((int)lo8, 1), PK_LO_INV, (float)h0[1]);
    v[2] = fmaf(__builtin_amdgcn_cvt_f32_fp8((int)lo8, 2), PK_LO_INV, (float)h1[0]);
    v[3] = fmaf(__builtin_amdgcn_cvt_f32_fp8((int)lo8, 3), PK_LO_INV, (float)h1[1]);
    return v;
}
__device__ __forceinline__ void pack4(f4 v, uint2& hi, unsigned& lo8, unsigned& x8) {
    float c[4], r[4], s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_fmed3f(v[j], -65504.0f, 65504.0f);
    const h2_t h0 = {(_Float16)c[0], (_Float16)c[1]}, h1 = {(_Float16)c[2], (_Float16)c[3]};
    hi.x = __builtin_bit_cast(unsigned, h0); hi.y = __builtin_bit_cast(unsigned, h1);
    const float hf[4] = {(float)h0[0], (float)h0[1], (float)h1[0], (float)h1[1]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = __builtin_amdgcn_fmed3f((c[j] - hf[j]) * PK_LO_SCALE, -448.0f, 448.0f);
        s[j] = __builtin_amdgcn_fmed3f(v[j], -448.0f, 448.0f);
    }
    int t = __builtin_amdgcn_cvt_pk_fp8_f32(r[0], r[1], 0, false);
    lo8 = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(r[2], r[3], t, true);
    t = __builtin_amdgcn_cvt_pk_fp8_f32(s[0], s[1], 0, false);
    x8 = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(s[2], s[3], t, true);
}
__device__ __forceinline__ f4 load4_packed(const float* p, unsigned row_off, int c4) {
    return unpack4(*(const uint2*)(p + pk_off(row_off, c4)), *(const unsigned*)(p + pk_lo_off(row_off, c4)));
}
__device__ __forceinline__ void store4_packed(float* p, unsigned row_off, int c4, f4 v) {
    uint2 hi; unsigned lo8, x8;
    pack4(v, hi, lo8, x8);
    *(uint2*)(p + pk_off(row_off, c4)) = hi;
    unsigned* q = (unsigned*)(p + pk_lo_off(row_off, c4));
    q[0] = lo8; q[4] = x8;
}
// ---- the second PACKED format, "H2" (conv.h): per 16 channels 16 f16 'hi' | 16 f16 'lo' of x * 2^H2_ACT_EXP -------------
// hi = RNE_f16(x 2^a) (saturating), lo = RNE_f16(x 2^a - hi): 22 significant bits -- the fp32-grade arithmetic mode (three
// f16 products per term, conv.hip).  A 4-channel run is an 8-B hi piece and an 8-B lo piece 32 B further on.
constexpr float H2_SCALE = 16.0f, H2_INV = 1.0f / 16.0f;               // 2^H2_ACT_EXP (conv.h)
__device__ __forceinline__ f4 unpack4_h2(uint2 hi, uint2 lo) {
    const h2_t a0 = __builtin_bit_cast(h2_t, hi.x), a1 = __builtin_bit_cast(h2_t, hi.y);
    const h2_t b0 = __builtin_bit_cast(h2_t, lo.x), b1 = __builtin_bit_cast(h2_t, lo.y);
    f4 v;
    v[0] = ((float)a0[0] + (float)b0[0]) * H2_INV; v[1] = ((float)a0[1] + (float)b0[1]) * H2_INV;
    v[2] = ((float)a1[0] + (float)b1[0]) * H2_INV; v[3] = ((float)a1[1] + (float)b1[1]) * H2_INV;
    return v;
}
__device__ __forceinline__ void pack4_h2(f4 v, uint2& hi, uint2& lo) {
    float c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_fmed3f(v[j] * H2_SCALE, -65504.0f, 65504.0f);
    const h2_t h0 = {(_Float16)c[0], (_Float16)c[1]}, h1 = {(_Float16)c[2], (_Float16)c[3]};
    const h2_t l0 = {(_Float16)(c[0] - (float)h0[0]), (_Float16)(c[1] - (float)h0[1])};
    const h2_t l1 = {(_Float16)(c[2] - (float)h1[0]), (_Float16)(c[3] - (float)h1[1])};
    hi.x = __builtin_bit_cast(unsigned, h0); hi.y = __builtin_bit_cast(unsigned, h1);
    lo.x = __builtin_bit_cast(unsigned, l0); lo.y = __builtin_bit_cast(unsigned, l1);
}
__device__ __forceinline__ f4 load4_h2(const float* p, unsigned row_off, int c4) {
    const float* q = p + pk_off(row_off, c4);
    return unpack4_h2(*(const uint2*)q, *(const uint2*)(q + 8));
}
__device__ __forceinline__ void store4_h2(float* p, unsigned row_off, int c4, f4 v) {
    uint2 hi, lo;
    pack4_h2(v, hi, lo);
    float* q = p + pk_off(row_off, c4);
    *(uint2*)q = hi; *(uint2*)(q + 8) = lo;
}
// ---- the third PACKED format, "P6" (conv.h): per 16 channels 16 f16 hi | 32 e2m3 codes [x_0/S, lo_0, x_1/S, lo_1, ...] with the
// group's own scale S = 2^(E - 2) | its E8M0 byte.  Written as WHOLE groups only (the scale needs the group's maximum):
// store16_p6 from a matrix-core epilogue; 4-channel readers decode by hand (debug copies, VALU consumers).
typedef unsigned v6u_t __attribute__((ext_vector_type(6)));
typedef float f32x16v_t __attribute__((ext_vector_type(16)));
typedef float f32x32v_t __attribute__((ext_vector_type(32)));
constexpr float P6_LO_SCALE = 2048.0f, P6_LO_INV = 1.0f / 2048.0f;      // 2^P6_LO_EXP (conv.h)
__device__ __forceinline__ float e2m3_dec(unsigned v) {
    const unsigned m = v & 31u;
    const unsigned mag = m < 8u ? m : ((8u | (m & 7u)) << ((m >> 3) - 1u));
    const float f = (float)mag * 0.125f;
    return (v & 32u) ? -f : f;
}
__device__ __forceinline__ f4 load4_p6(const float* p, unsigned row_off, int c4) {
    const float* g = p + row_off + (unsigned)(c4 & ~15);
    const int q = (c4 & 15) >> 2;
    const uint2 hi = *(const uint2*)(g + 2 * q);
    const unsigned short* s = (const unsigned short*)(g + 8) + 3 * q;      // 4 channels = 8 codes = 48 bits
    const unsigned long long bits = (unsigned long long)s[0] | ((unsigned long long)s[1] << 16) | ((unsigned long long)s[2] << 32);
    const float sc = __uint_as_float((__float_as_uint(g[14]) & 0xffu) << 23) * P6_LO_INV;
    const h2_t h0 = __builtin_bit_cast(h2_t, hi.x), h1 = __builtin_bit_cast(h2_t, hi.y);
    f4 v;
    v[0] = fmaf(e2m3_dec((unsigned)(bits >> 6) & 63u), sc, (float)h0[0]);
    v[1] = fmaf(e2m3_dec((unsigned)(bits >> 18) & 63u), sc, (float)h0[1]);
    v[2] = fmaf(e2m3_dec((unsigned)(bits >> 30) & 63u), sc, (float)h1[0]);
    v[3] = fmaf(e2m3_dec((unsigned)(bits >> 42) & 63u), sc, (float)h1[1]);
    return v;
}
// one whole P6 group (g = its 64 B) -> 16 values: a single v_cvt_scalef32_pk32_f32_fp6 decodes all the codes
__device__ __forceinline__ void load16_p6(const float* g, float (&v)[16]) {
    // (the 16 dwords are read as float4s and taken apart by name: indexing a 4-wide integer vector with the unrolled loop variable
    // made hipcc use element 0 for every k here -- tools/up_diag.py found channels 2-7 / 10-15 carrying the halves of channels 0-1 / 8-9)
    const float4* q = (const float4*)g;
    const float4 qa = q[0], qb = q[1], qc = q[2], qd = q[3];
    const unsigned a[4] = {__float_as_uint(qa.x), __float_as_uint(qa.y), __float_as_uint(qa.z), __float_as_uint(qa.w)};
    const unsigned b[4] = {__float_as_uint(qb.x), __float_as_uint(qb.y), __float_as_uint(qb.z), __float_as_uint(qb.w)};
    const v6u_t pk = {__float_as_uint(qc.x), __float_as_uint(qc.y), __float_as_uint(qc.z), __float_as_uint(qc.w), __float_as_uint(qd.x), __float_as_uint(qd.y)};
    const f32x32v_t un = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(pk, __uint_as_float((__float_as_uint(qd.z) & 0xffu) << 23));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const h2_t h0 = __builtin_bit_cast(h2_t, a[k]), h1 = __builtin_bit_cast(h2_t, b[k]);
        v[2 * k] = fmaf(un[4 * k + 1], P6_LO_INV, (float)h0[0]); v[2 * k + 1] = fmaf(un[4 * k + 3], P6_LO_INV, (float)h0[1]);
        v[8 + 2 * k] = fmaf(un[16 + 4 * k + 1], P6_LO_INV, (float)h1[0]); v[8 + 2 * k + 1] = fmaf(un[16 + 4 * k + 3], P6_LO_INV, (float)h1[1]);
    }
}
// 16 consecutive channels -> one P6 group
__device__ __forceinline__ void store16_p6(float* p, unsigned row_off, int cg, const float (&w)[16]) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    f32x16v_t c, lo;
    u4 hi0, hi1;
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { c[k] = __builtin_amdgcn_fmed3f(w[k], -65504.0f, 65504.0f); mx = fmaxf(mx, fabsf(c[k])); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const h2_t a = {(_Float16)c[2 * k], (_Float16)c[2 * k + 1]}, b = {(_Float16)c[8 + 2 * k], (_Float16)c[8 + 2 * k + 1]};
        hi0[k] = __builtin_bit_cast(unsigned, a); hi1[k] = __builtin_bit_cast(unsigned, b);
        lo[2 * k] = (c[2 * k] - (float)a[0]) * P6_LO_SCALE; lo[2 * k + 1] = (c[2 * k + 1] - (float)a[1]) * P6_LO_SCALE;
        lo[8 + 2 * k] = (c[8 + 2 * k] - (float)b[0]) * P6_LO_SCALE; lo[8 + 2 * k + 1] = (c[8 + 2 * k + 1] - (float)b[1]) * P6_LO_SCALE;
    }
    unsigned eb = __float_as_uint(mx) >> 23;
    eb = eb > 3u ? eb - 2u : 1u;
    // (v_cvt_scalef32_2xpk16_fp6_f32: element 2i = src0[i] / 2^k, element 2i + 1 = src1[i] / 2^k, RNE, saturating at 7.5)
    const v6u_t pk = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(c, lo, __uint_as_float(eb << 23));
    u4* q = (u4*)(p + row_off + (unsigned)cg);
    q[0] = hi0; q[1] = hi1; q[2] = u4{pk[0], pk[1], pk[2], pk[3]}; q[3] = u4{pk[4], pk[5], eb, 0u};
}
// Range guard of the packed formats.  PACKED: the fp8 residual lo8 = (x - hi) 2^12 saturates from |x| >= 256 on (and the fp8
// copy x8 at 448): beyond that a value keeps only its f16 half (2^-12 relative).  H2: x 2^4 saturates at +-65504, i.e.
// |x| > 4094 is CLAMPED.  Producers count the 4- / 16-channel runs that cross the limit in a per-layer counter
// (evr_model_saturation); the test is a handful of v_max per run and the atomic fires only when it trips.
template <int FMT> __device__ __forceinline__ void sat_note(unsigned* sat, float mx) {
    constexpr float LIM = (FMT == 2) ? 65504.0f / H2_SCALE : (FMT == 3 ? 65504.0f : 256.0f);      // (P6 scales per group: only the f16 half can clamp)
    if (sat && mx > LIM) atomicAdd(sat, 1u);
}
template <int FMT> __device__ __forceinline__ void sat_check4(unsigned* sat, f4 v) {
    sat_note<FMT>(sat, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
}
template <int FMT> __device__ __forceinline__ void sat_check16(unsigned* sat, const float (&w)[16]) {
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) mx = fmaxf(mx, fabsf(w[k]));
    sat_note<FMT>(sat, mx);
}
// compile-time format selection for the matrix-core kernels (FMT: 1 = the f16 | fp8 | fp8 format above, 2 = H2)
template <int FMT> __device__ __forceinline__ f4 load4_fmt(const float* p, unsigned row_off, int c4) {
    if constexpr (FMT == 3) return load4_p6(p, row_off, c4);
    else if constexpr (FMT == 2) return load4_h2(p, row_off, c4);
    else return load4_packed(p, row_off, c4);
}
template <int FMT> __device__ __forceinline__ void store4_fmt(float* p, unsigned row_off, int c4, f4 v) {
    if constexpr (FMT == 3) __builtin_trap();      // P6 has no 4-channel writer (launch_conv_igemm_m6 refuses such plans)
    else if constexpr (FMT == 2) store4_h2(p, row_off, c4, v);
    else store4_packed(p, row_off, c4, v);
}
// ---- whole-group stores from the MFMA accumulator layout ----------------------------------------------------------
// A lane of the 32x32 MFMA result holds, of a 32-channel block, v[4q + j] = channel 8q + 4h + j (h = lane >> 5, lanes l
// and l ^ 32 = the two halves of one pixel).  xchg16 trades two 4-runs with the partner (v_permlane32_swap) so that the
// lane owns the 16 CONSECUTIVE channels 16h .. 16h + 15 -- one whole PACKED group, stored as four contiguous 16-B
// pieces instead of twelve scattered 8- and 4-B ones.  Partners must be both active or both inactive.
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void xchg16(const f32x16_t& v, float (&w)[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // swap(X, Y): X's upper-half lanes <-> Y's lower-half lanes.  (q0, q2): lower lanes end with own q0 | partner's q0,
        // upper lanes with partner's q2 | own q2; (q1, q3) likewise
        const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[8 + j]), false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[4 + j]), __float_as_uint(v[12 + j]), false, false);
        w[j] = __uint_as_float(a[0]); w[4 + j] = __uint_as_float(a[1]);
        w[8 + j] = __uint_as_float(b[0]); w[12 + j] = __uint_as_float(b[1]);
    }
}
// 16 consecutive channels (cg % 16 == 0) of the pixel row at float offset row_off -> one PACKED group
__device__ __forceinline__ void store16_packed(float* p, unsigned row_off, int cg, const float (&w)[16]) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    float c[16];
    u4 hi0, hi1, lo, x8;
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = __builtin_amdgcn_fmed3f(w[k], -65504.0f, 65504.0f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const h2_t a = {(_Float16)c[2 * k], (_Float16)c[2 * k + 1]}, b = {(_Float16)c[8 + 2 * k], (_Float16)c[8 + 2 * k + 1]};
        hi0[k] = __builtin_bit_cast(unsigned, a); hi1[k] = __builtin_bit_cast(unsigned, b);
        c[2 * k] = (c[2 * k] - (float)a[0]) * PK_LO_SCALE; c[2 * k + 1] = (c[2 * k + 1] - (float)a[1]) * PK_LO_SCALE;
        c[8 + 2 * k] = (c[8 + 2 * k] - (float)b[0]) * PK_LO_SCALE; c[8 + 2 * k + 1] = (c[8 + 2 * k + 1] - (float)b[1]) * PK_LO_SCALE;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int t = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c[4 * k], -448.0f, 448.0f), __builtin_amdgcn_fmed3f(c[4 * k + 1], -448.0f, 448.0f), 0, false);
        lo[k] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c[4 * k + 2], -448.0f, 448.0f), __builtin_amdgcn_fmed3f(c[4 * k + 3], -448.0f, 448.0f), t, true);
        t = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(w[4 * k], -448.0f, 448.0f), __builtin_amdgcn_fmed3f(w[4 * k + 1], -448.0f, 448.0f), 0, false);
        x8[k] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(w[4 * k + 2], -448.0f, 448.0f), __builtin_amdgcn_fmed3f(w[4 * k + 3], -448.0f, 448.0f), t, true);
    }
    u4* q = (u4*)(p + row_off + (unsigned)cg);
    q[0] = hi0; q[1] = hi1; q[2] = lo; q[3] = x8;
}
// H2 twins of store16_packed / unpack16_xchg: the group is hi 8 dwords | lo 8 dwords
__device__ __forceinline__ void store16_h2(float* p, unsigned row_off, int cg, const float (&w)[16]) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 hi0, hi1, lo0, lo1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float c0 = __builtin_amdgcn_fmed3f(w[2 * k] * H2_SCALE, -65504.0f, 65504.0f), c1 = __builtin_amdgcn_fmed3f(w[2 * k + 1] * H2_SCALE, -65504.0f, 65504.0f);
        const float d0 = __builtin_amdgcn_fmed3f(w[8 + 2 * k] * H2_SCALE, -65504.0f, 65504.0f), d1 = __builtin_amdgcn_fmed3f(w[8 + 2 * k + 1] * H2_SCALE, -65504.0f, 65504.0f);
        const h2_t a = {(_Float16)c0, (_Float16)c1}, b = {(_Float16)d0, (_Float16)d1};
        const h2_t al = {(_Float16)(c0 - (float)a[0]), (_Float16)(c1 - (float)a[1])}, bl = {(_Float16)(d0 - (float)b[0]), (_Float16)(d1 - (float)b[1])};
        hi0[k] = __builtin_bit_cast(unsigned, a); hi1[k] = __builtin_bit_cast(unsigned, b);
        lo0[k] = __builtin_bit_cast(unsigned, al); lo1[k] = __builtin_bit_cast(unsigned, bl);
    }
    u4* q = (u4*)(p + row_off + (unsigned)cg);
    q[0] = hi0; q[1] = hi1; q[2] = lo0; q[3] = lo1;
}
__device__ __forceinline__ void unpack16_xchg_h2(const unsigned (&g)[16], f32x16_t& v) {
    float w[16];
#pragma unroll
    for (int d = 0; d < 8; ++d) {      // dword d = channels 2d, 2d + 1 (hi in g[d], lo in g[8 + d])
        const h2_t hh = __builtin_bit_cast(h2_t, g[d]), ll = __builtin_bit_cast(h2_t, g[8 + d]);
        w[2 * d] = ((float)hh[0] + (float)ll[0]) * H2_INV;
        w[2 * d + 1] = ((float)hh[1] + (float)ll[1]) * H2_INV;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[j]), __float_as_uint(w[4 + j]), false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[8 + j]), __float_as_uint(w[12 + j]), false, false);
        v[j] = __uint_as_float(a[0]); v[8 + j] = __uint_as_float(a[1]);
        v[4 + j] = __uint_as_float(b[0]); v[12 + j] = __uint_as_float(b[1]);
    }
}
// (P6: dwords 0-7 hi | 8-13 codes | 14 scale byte)
__device__ __forceinline__ void unpack16_xchg_p6(const unsigned (&g)[16], f32x16_t& v) {
    float w[16];
    const v6u_t pk = {g[8], g[9], g[10], g[11], g[12], g[13]};
    const f32x32v_t un = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(pk, __uint_as_float((g[14] & 0xffu) << 23));
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const h2_t hh = __builtin_bit_cast(h2_t, g[d]);
        w[2 * d] = fmaf(un[4 * d + 1], P6_LO_INV, (float)hh[0]);
        w[2 * d + 1] = fmaf(un[4 * d + 3], P6_LO_INV, (float)hh[1]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[j]), __float_as_uint(w[4 + j]), false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[8 + j]), __float_as_uint(w[12 + j]), false, false);
        v[j] = __uint_as_float(a[0]); v[8 + j] = __uint_as_float(a[1]);
        v[4 + j] = __uint_as_float(b[0]); v[12 + j] = __uint_as_float(b[1]);
    }
}
template <int FMT> __device__ __forceinline__ void store16_fmt(float* p, unsigned row_off, int cg, const float (&w)[16]) {
    if constexpr (FMT == 3) store16_p6(p, row_off, cg, w);
    else if constexpr (FMT == 2) store16_h2(p, row_off, cg, w);
    else store16_packed(p, row_off, cg, w);
}
// the reverse: one PACKED group as loaded (hi 8 dwords | lo8 4 dwords) -> its 16 values -> the accumulator order of the
// lane pair (the same swaps: v_permlane32_swap is its own inverse on a register pair)
__device__ __forceinline__ void unpack16_xchg(const unsigned (&g)[12], f32x16_t& v) {
    float w[16];
#pragma unroll
    for (int d = 0; d < 4; ++d) {      // lo8 dword d = channels 4d .. 4d + 3 = hi dwords 2d, 2d + 1
        const h2_t h0 = __builtin_bit_cast(h2_t, g[2 * d]), h1 = __builtin_bit_cast(h2_t, g[2 * d + 1]);
        const int lo = (int)g[8 + d];
        w[4 * d] = fmaf(__builtin_amdgcn_cvt_f32_fp8(lo, 0), PK_LO_INV, (float)h0[0]);
        w[4 * d + 1] = fmaf(__builtin_amdgcn_cvt_f32_fp8(lo, 1), PK_LO_INV, (float)h0[1]);
        w[4 * d + 2] = fmaf(__builtin_amdgcn_cvt_f32_fp8(lo, 2), PK_LO_INV, (float)h1[0]);
        w[4 * d + 3] = fmaf(__builtin_amdgcn_cvt_f32_fp8(lo, 3), PK_LO_INV, (float)h1[1]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[j]), __float_as_uint(w[4 + j]), false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[8 + j]), __float_as_uint(w[12 + j]), false, false);
        v[j] = __uint_as_float(a[0]); v[8 + j] = __uint_as_float(a[1]);
        v[4 + j] = __uint_as_float(b[0]); v[12 + j] = __uint_as_float(b[1]);
    }
}
// one channel of a PACKED pixel row (fmt: 1 = f16 | fp8 | fp8, 2 = H2)
__device__ __forceinline__ float load1_packed(const float* row, int ch, int fmt = 1) {
    const unsigned char* g = (const unsigned char*)(row + (ch & ~15));
    const int k = ch & 15;
    if (fmt == 2) return ((float)((const _Float16*)g)[k] + (float)((const _Float16*)g)[16 + k]) * H2_INV;
    if (fmt == 3) {
        const int bit = 12 * k + 6;
        const unsigned short* s = (const unsigned short*)(g + 32) + (bit >> 4);
        const unsigned code = (((unsigned)s[0] | ((unsigned)s[1] << 16)) >> (bit & 15)) & 63u;
        return fmaf(e2m3_dec(code), __uint_as_float((unsigned)g[56] << 23) * P6_LO_INV, (float)((const _Float16*)g)[k]);
    }
    return fmaf(__builtin_amdgcn_cvt_f32_fp8((int)g[32 + k], 0), PK_LO_INV, (float)((const _Float16*)g)[k]);
}
#endif

// 4 consecutive channels (c4 % 4 == 0) of the pixel row at `row`: packed = 0 PLAIN, 1 PACKED (f16 | fp8 | fp8), 2 H2
__device__ __forceinline__ float4 ld4_any(const float* row, int c4, int packed) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (packed == 3) { const f4 t = load4_p6(row, 0u, c4); return make_float4(t[0], t[1], t[2], t[3]); }
    if (packed == 2) { const f4 t = load4_h2(row, 0u, c4); return make_float4(t[0], t[1], t[2], t[3]); }
    if (packed) { const f4 t = load4_packed(row, 0u, c4); return make_float4(t[0], t[1], t[2], t[3]); }
#endif
    return *(const float4*)(row + c4);
}
__device__ __forceinline__ void st4_any(float* row, int c4, float4 v, int packed) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (packed == 3) __builtin_trap();      // P6 tensors are written as whole groups only (packed.h store16_p6)
    if (packed == 2) { const f4 t = {v.x, v.y, v.z, v.w}; store4_h2(row, 0u, c4, t); return; }
    if (packed) { const f4 t = {v.x, v.y, v.z, v.w}; store4_packed(row, 0u, c4, t); return; }
#endif
    *(float4*)(row + c4) = v;
}


}  // namespace evr

// Device helpers for the PACKED activation format (conv.h): per 16 channels 16 f16 'hi' | 16 fp8 'lo8' | 16 fp8 'x8'.
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
// one channel of a PACKED pixel row
__device__ __forceinline__ float load1_packed(const float* row, int ch) {
    const unsigned char* g = (const unsigned char*)(row + (ch & ~15));
    const int k = ch & 15;
    return fmaf(__builtin_amdgcn_cvt_f32_fp8((int)g[32 + k], 0), PK_LO_INV, (float)((const _Float16*)g)[k]);
}
#endif

// 4 consecutive channels (c4 % 4 == 0) of the pixel row at `row`, PLAIN or PACKED
__device__ __forceinline__ float4 ld4_any(const float* row, int c4, int packed) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (packed) { const f4 t = load4_packed(row, 0u, c4); return make_float4(t[0], t[1], t[2], t[3]); }
#endif
    return *(const float4*)(row + c4);
}
__device__ __forceinline__ void st4_any(float* row, int c4, float4 v, int packed) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (packed) { const f4 t = {v.x, v.y, v.z, v.w}; store4_packed(row, 0u, c4, t); return; }
#endif
    *(float4*)(row + c4) = v;
}


}  // namespace evr

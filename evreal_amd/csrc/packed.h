// Device helpers for the PACKED activation format (conv.h): bf16 hi|lo halves per 8 channels.
#pragma once
#include <hip/hip_runtime.h>

namespace evr {

#if defined(__HIP_DEVICE_COMPILE__)
// two fp32 -> packed bf16 (RNE): `lo` lands in bits 15:0, `hi` in bits 31:16 (no builtin on gfx950)
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
typedef float f4 __attribute__((ext_vector_type(4)));
// PACKED activation format (conv.h): 4 consecutive channels c4..c4+3 (c4 % 4 == 0) of a pixel row are an 8-B 'hi'
// piece and, 16 B further, an 8-B 'lo' piece.  pk_off = float-element offset of the hi piece; lo = +4 floats.
__device__ __forceinline__ unsigned pk_off(unsigned row_off, int c4) { return row_off + (unsigned)((c4 & ~7) + ((c4 & 4) >> 1)); }
__device__ __forceinline__ f4 unpack4(uint2 hi, uint2 lo) {
    f4 v;
    v[0] = __uint_as_float(hi.x << 16) + __uint_as_float(lo.x << 16);
    v[1] = __uint_as_float(hi.x & 0xffff0000u) + __uint_as_float(lo.x & 0xffff0000u);
    v[2] = __uint_as_float(hi.y << 16) + __uint_as_float(lo.y << 16);
    v[3] = __uint_as_float(hi.y & 0xffff0000u) + __uint_as_float(lo.y & 0xffff0000u);
    return v;
}
__device__ __forceinline__ void pack4(f4 v, uint2& hi, uint2& lo) {
    hi.x = cvt_pk_bf16(v[0], v[1]); hi.y = cvt_pk_bf16(v[2], v[3]);
    lo.x = cvt_pk_bf16(v[0] - __uint_as_float(hi.x << 16), v[1] - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = cvt_pk_bf16(v[2] - __uint_as_float(hi.y << 16), v[3] - __uint_as_float(hi.y & 0xffff0000u));
}
__device__ __forceinline__ f4 load4_packed(const float* p, unsigned row_off, int c4) {
    const float* q = p + pk_off(row_off, c4);
    return unpack4(*(const uint2*)q, *(const uint2*)(q + 4));
}
__device__ __forceinline__ void store4_packed(float* p, unsigned row_off, int c4, f4 v) {
    uint2 hi, lo;
    pack4(v, hi, lo);
    float* q = p + pk_off(row_off, c4);
    *(uint2*)q = hi; *(uint2*)(q + 4) = lo;
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

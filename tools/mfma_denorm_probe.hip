// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal INPUTS on gfx950?  (Decides H2_ACT_EXP, csrc/conv.h: the lo half of an
// H2 value is a subnormal half for small |x|.)  Prints the accumulated product of A = 2^-20 (subnormal) with B = 2^10.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float* out, unsigned short abits, unsigned short bbits) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = __builtin_bit_cast(_Float16, abits); b[i] = __builtin_bit_cast(_Float16, bbits); }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
    // the conversion path the packers use: fp32 -> f16 of a value in the subnormal range
    if (threadIdx.x == 0) { _Float16 h = (_Float16)3.0e-6f; out[1] = (float)h; }
}
int main() {
    float* d; hipMalloc(&d, 8);
    // 2^-20 as a half: subnormal, mantissa = 2^-20 / 2^-24 = 16 -> bits 0x0010; 2^10 = 0x6400
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, (unsigned short)0x0010, (unsigned short)0x6400);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("mfma(2^-20 subnormal x 2^10) over K=16: got %.9g, exact %.9g -> %s\n", h[0], 16.0 * 0.0009765625, h[0] > 0.015 ? "subnormals HONOURED" : "subnormals FLUSHED");
    printf("cvt_f16_f32(3.0e-6) = %.9g (exact grid value 2.98e-6 if subnormals are kept, 0 if flushed)\n", h[1]);
    return 0;
}

// Probe: semantics of v_cvt_scalef32_2xpk16_fp6_f32 / v_cvt_scalef32_pk32_f32_fp6 (scale direction, rounding, saturation) and of
// the E8M0 scale operands of v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));
__global__ void k(const float* in, float scale_pack, float scale_unpack, float* out, unsigned* raw, float* mm, int sa, int sb) {
    v16f s0, s1;
    for (int i = 0; i < 16; ++i) { s0[i] = in[i]; s1[i] = in[16 + i]; }
    const v6u pk = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(s0, s1, scale_pack);
    const v32f un = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(pk, scale_unpack);
    if (threadIdx.x == 0) { for (int i = 0; i < 32; ++i) out[i] = un[i]; for (int i = 0; i < 6; ++i) raw[i] = pk[i]; }
    // all-ones (1.0 = 0b001000) A and B with scale bytes sa / sb: C = 64 * 2^(sa - 127) * 2^(sb - 127)
    v8i a = {}; unsigned long long b[3] = {0, 0, 0};
    for (int j = 0; j < 32; ++j) { const int bit = 6 * j; b[bit >> 6] |= 8ull << (bit & 63); if ((bit & 63) > 58) b[(bit >> 6) + 1] |= 8ull >> (64 - (bit & 63)); }
    a[0] = (int)b[0]; a[1] = (int)(b[0] >> 32); a[2] = (int)b[1]; a[3] = (int)(b[1] >> 32); a[4] = (int)b[2]; a[5] = (int)(b[2] >> 32);
    v8i a2 = a; a2[6] = sa; v8i b2 = a; b2[6] = sb;
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a2, b2, acc, 2, 2, 0, a2[6], 0, b2[6]);
    if (threadIdx.x == 0) mm[0] = acc[0];
}
int main() {
    float h[32] = {0.06f, 0.0625f, 0.1f, 0.19f, 0.3f, 0.9f, 0.94f, 1.06f, 1.0625f, 1.1875f, 2.1f, 2.125f, 2.375f, 3.9f, 4.2f, 4.25f,
                   4.75f, 7.2f, 7.3f, 7.6f, 7.8f, 8.5f, 100.f, -0.06f, -0.07f, -1.06f, -7.9f, -1e9f, 1e-9f, 5.0f, 6.0f, 0.f};
    float *in, *out, *mm; unsigned* raw;
    hipMalloc(&in, 128); hipMalloc(&out, 128); hipMalloc(&raw, 24); hipMalloc(&mm, 4);
    hipMemcpy(in, h, 128, hipMemcpyHostToDevice);
    const float sp[4] = {1.f, 4.f, 0.25f, 6.f}, su[4] = {1.f, 1.f, 1.f, 1.f};
    for (int t = 0; t < 5; ++t) {
        const float a = t < 4 ? sp[t] : 1.f, b = t < 4 ? su[t] : 4.f;
        k<<<1, 64>>>(in, a, b, out, raw, mm, 127 + t, 127 - 2 * t);
        float o[32]; float m; hipMemcpy(o, out, 128, hipMemcpyDeviceToHost); hipMemcpy(&m, mm, 4, hipMemcpyDeviceToHost);
        printf("pack scale %g, unpack scale %g (mfma scales %d, %d -> C = %g):\n ", a, b, 127 + t, 127 - 2 * t, m);
        // un-interleave: element 2i = s0[i], 2i + 1 = s1[i]
        for (int i = 0; i < 32; ++i) printf(" %g->%g", h[i], o[i < 16 ? 2 * i : 2 * (i - 16) + 1]);
        printf("\n");
    }
    return 0;
}

// Probe (not part of the product): (1) element order of v_cvt_scalef32_2xpk16_fp6_f32 / v_cvt_scalef32_pk32_f32_fp6 and of the
// fp6 operands of v_mfma_scale_f32_32x32x64_f8f6f4; (2) chip-wide matrix-core rates of the instruction mixes of the split
// arithmetic under the power limit: f16 only, 2 f16 + 1 fp8 (the 'mx' mix), 2 f16 + 1 fp6 (e2m3).
//   hipcc --offload-arch=gfx950 -O3 tools/fp6_probe.hip -o tools/_bin/fp6_probe && tools/_bin/fp6_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

__global__ void order_kernel(const float* tab, float* unpacked, float* cmat) {
    const int l = threadIdx.x;
    v16f s0, s1;
    for (int i = 0; i < 16; ++i) { s0[i] = tab[i]; s1[i] = tab[16 + i]; }
    const v6u pk = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(s0, s1, 1.0f);
    const v32f un = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(pk, 1.0f);
    if (l == 0) for (int i = 0; i < 32; ++i) unpacked[i] = un[i];
    // MFMA: A row r = lane & 31 holds for k-half h = lane >> 5 the packed values above (same for every row); B = manual one-hot
    // assuming element j of a lane sits at bits [6j, 6j + 5]: column c selects k = c (pass 0) or k = 32 + c (pass 1)
    v8i a = {}; for (int i = 0; i < 6; ++i) a[i] = (int)pk[i];
    for (int pass = 0; pass < 2; ++pass) {
        const int c = l & 31, h = l >> 5;
        unsigned long long bits[3] = {0, 0, 0};   // 192 bits
        if (h == pass) { const int j = c, bit = 6 * j; bits[bit >> 6] |= (unsigned long long)8 << (bit & 63);
                         if ((bit & 63) > 58) bits[(bit >> 6) + 1] |= (unsigned long long)8 >> (64 - (bit & 63)); }
        v8i b = {};
        b[0] = (int)bits[0]; b[1] = (int)(bits[0] >> 32); b[2] = (int)bits[1]; b[3] = (int)(bits[1] >> 32); b[4] = (int)bits[2]; b[5] = (int)(bits[2] >> 32);
        v16f acc = {};
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        // acc[4q + j] of lane l = C[row = 8q + 4(l >> 5) + j][col = l & 31]
        for (int q = 0; q < 4; ++q) for (int j = 0; j < 4; ++j) cmat[pass * 1024 + (8 * q + 4 * (l >> 5) + j) * 32 + (l & 31)] = acc[4 * q + j];
    }
}

template <int MIX> __global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned seed) {
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    v8h ah[2], bh[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) { ah[i][j] = (_Float16)((int)(rnd() >> 20) * (1.0f / 2048.0f) - 1.0f); bh[i][j] = (_Float16)((int)(rnd() >> 20) * (1.0f / 2048.0f) - 1.0f); }
    v8i a8, b8;
    for (int j = 0; j < 8; ++j) { a8[j] = (int)(rnd() & 0x3f3f3f3fu); b8[j] = (int)(rnd() & 0x3f3f3f3fu); }   // finite fp8 / arbitrary fp6 bits
    v16f acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MIX == 0 || MIX == 1 || MIX == 2) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[0], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[1], acc[u], 0, 0, 0);
            }
            if (MIX == 1 || MIX == 3) acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[u], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            if (MIX == 2 || MIX == 4) acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[u], 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            if (MIX == 5) {      // three f16 products (the 'h3' mix)
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[0], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[1], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[1], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[0], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[1], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[0], acc[u], 0, 0, 0);
            }
        }
    }
    float r = 0.f;
    for (int u = 0; u < 4; ++u) for (int j = 0; j < 16; ++j) r += acc[u][j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MIX> void rate(const char* name, float* d, int waves_per_simd, int iters, double units_per_iter_block) {
    const int blocks = 256 * waves_per_simd;        // 4 waves per block = one per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MIX><<<blocks, 256>>>(d, iters / 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<MIX><<<blocks, 256>>>(d, iters, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // one "unit" = a 32 x 32 output block x 32 channels of K: 2 * 32 * 32 * 32 algorithmic flops
    const double units = (double)blocks * 4 * iters * 4 * units_per_iter_block;
    printf("{\"mix\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"alg_TFLOPs\": %.1f}\n", name, waves_per_simd, ms, units * 65536.0 / (ms * 1e-3) / 1e12);
}

int main() {
    const float tabh[32] = {0, .125f, .25f, .375f, .5f, .625f, .75f, .875f, 1, 1.125f, 1.25f, 1.375f, 1.5f, 1.625f, 1.75f, 1.875f,
                            2, 2.25f, 2.5f, 2.75f, 3, 3.25f, 3.5f, 3.75f, 4, 4.5f, 5, 5.5f, 6, 6.5f, 7, 7.5f};
    float *tab, *un, *cm, *out;
    hipMalloc(&tab, 128); hipMalloc(&un, 128); hipMalloc(&cm, 8192); hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMemcpy(tab, tabh, 128, hipMemcpyHostToDevice);
    order_kernel<<<1, 64>>>(tab, un, cm);
    float unh[32]; std::vector<float> cmh(2048);
    hipMemcpy(unh, un, 128, hipMemcpyDeviceToHost); hipMemcpy(cmh.data(), cm, 8192, hipMemcpyDeviceToHost);
    auto idx = [&](float v) { for (int i = 0; i < 32; ++i) if (tabh[i] == v) return i; return -1; };
    printf("unpack(2xpk16(s0, s1)) order (table index; s0 = 0..15, s1 = 16..31):");
    for (int i = 0; i < 32; ++i) printf(" %d", idx(unh[i]));
    printf("\nMFMA k -> table index (row 0; one-hot B assumes element j at bits 6j):");
    for (int k = 0; k < 64; ++k) printf(" %d", idx(cmh[(k >> 5) * 1024 + 0 * 32 + (k & 31)]));
    printf("\nrow 5 check:");
    for (int k = 0; k < 64; k += 7) printf(" %d", idx(cmh[(k >> 5) * 1024 + 5 * 32 + (k & 31)]));
    printf("\n");
    for (int w = 1; w <= 2; ++w) {
        rate<0>("f16_only", out, w, 4000, 1.0);
        rate<1>("2f16+fp8 (mx)", out, w, 4000, 1.0);
        rate<2>("2f16+fp6", out, w, 4000, 1.0);
        rate<3>("fp8_only", out, w, 4000, 1.0);
        rate<4>("fp6_only", out, w, 4000, 1.0);
        rate<5>("3 f16 products (h3)", out, w, 2000, 1.0);
    }
    return 0;
}

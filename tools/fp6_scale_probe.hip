// Probe: E8M0 scale operands of v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 (e2m3) operands.  One MFMA per launch, every operand
// (A, B, C, per-lane scale registers) comes from memory, so the host decides what each lane holds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int F> __global__ void k(const v8i* a, const v8i* b, const int* sa, const int* sb, v16f* c) {
    const int l = threadIdx.x;
    v16f acc = c[l];
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, F, F, 0, sa[l], 0, sb[l]);
    c[l] = acc;
}
static void ones6(int* r) {     // 32 e2m3 elements = 1.0 (0b001000) in 6 registers
    unsigned long long b[3] = {0, 0, 0};
    for (int j = 0; j < 32; ++j) { const int bit = 6 * j; b[bit >> 6] |= 8ull << (bit & 63); if ((bit & 63) > 58) b[(bit >> 6) + 1] |= 8ull >> (64 - (bit & 63)); }
    memcpy(r, b, 24); r[6] = r[7] = 0;
}
int main() {
    int ha[64][8], hb[64][8], hsa[64], hsb[64]; float hc[64][16];
    v8i *a, *b; int *sa, *sb; v16f* c;
    hipMalloc(&a, sizeof ha); hipMalloc(&b, sizeof hb); hipMalloc(&sa, 256); hipMalloc(&sb, 256); hipMalloc(&c, sizeof hc);
    auto run = [&](const char* what) {
        memset(hc, 0, sizeof hc);
        hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
        hipMemcpy(sa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(sb, hsb, 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, sizeof hc, hipMemcpyHostToDevice);
        k<2><<<1, 64>>>(a, b, sa, sb, c);
        hipMemcpy(hc, c, sizeof hc, hipMemcpyDeviceToHost);
        // acc[4q + j] of lane l = C[i = 8q + 4(l >> 5) + j][j' = l & 31], i indexes the FIRST operand's rows
        printf("%-58s C[0][0] %g  C[1][0] %g  C[0][1] %g  C[4][0] %g  C[5][3] %g\n", what, hc[0][0], hc[0][1], hc[1][0], hc[32][0], hc[32 + 3][1]);
    };
    for (int l = 0; l < 64; ++l) { ones6(ha[l]); ones6(hb[l]); hsa[l] = 127; hsb[l] = 127; }
    run("all ones, scales 127/127 (expect 64):");
    for (int l = 0; l < 64; ++l) hsa[l] = 128;
    run("A scale 128 in every lane (expect 128):");
    for (int l = 0; l < 64; ++l) hsa[l] = 127 + 3 * (l >> 5);
    run("A scale 127 lanes 0-31, 130 lanes 32-63 (32 + 256 = 288?):");
    for (int l = 0; l < 64; ++l) hsa[l] = 127 + (l & 1);
    run("A scale 127 + (lane & 1): rows alternate 64 / 128?:");
    for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127 + (l & 1); }
    run("B scale 127 + (lane & 1): columns alternate?:");
    for (int l = 0; l < 64; ++l) { hsa[l] = 127 | (130 << 8); hsb[l] = 127; }
    run("A scale byte1 = 130, byte0 = 127, opsel 0 (expect 64):");
    for (int l = 0; l < 64; ++l) { hsa[l] = 127; if (l >= 32) memset(ha[l], 0, 32); }
    run("A zero in lanes 32-63 (k block 1 empty: expect 32):");
    return 0;
}

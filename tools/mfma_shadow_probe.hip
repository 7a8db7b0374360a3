// How much independent work does ONE wave hide in the shadow of its own MFMAs on gfx950?  (Round 6: wino.hip's side work -- 32 packed
// adds, 8 LDS stores, 24 memory requests per 64 fp32 MFMAs -- cost 12 % although it is spread over the gaps between the MFMAs.)
// One wave per SIMD (256-thread blocks, 1 per CU, 512 registers), a loop of dependent accumulate chains like the kernel's:
//   per iteration: 4 x v_mfma_f32_32x32x2_f32 on one accumulator (256 matrix cycles), with N independent side instructions after
//   the first MFMA -- v_pk_add_f32 (VALU), ds_write_b64 (LDS store), ds_read_b128 (LDS load), or the same MFMAs with f16 32x32x16.
// Prints cycles per iteration (s_memtime) for N = 0 .. 32: flat = hidden, slope = exposed issue cost.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shadow_probe.hip -o /tmp/shadow && /tmp/shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int N, int F16>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters, float seed) {
    __shared__ float4 lds[4096];
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int p = 0; p < 4; ++p) for (int j = 0; j < 16; ++j) acc[p][j] = 0.f;
    f2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f2{seed + i, seed - i};
    float4 r[4];
    for (int i = 0; i < 4; ++i) r[i] = make_float4(seed, seed, seed, seed);
    float a = seed, b = seed * 0.5f;
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(seed + i); hb[i] = (_Float16)(seed - i); }
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_float4(seed, 0, 0, 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (F16) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[p], 0, 0, 0);
            else acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
                if (KIND == 1) asm volatile("ds_write_b64 %0, %1" :: "v"((lane * 8 + (i & 7) * 512) & 0x7fff), "v"(v[i & 7]) : "memory");
                if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(r[i & 3]) : "v"((lane * 16 + (i & 7) * 1024) & 0xffff) : "memory");
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (F16) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[p], 0, 0, 0);
                else acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int p = 0; p < 4; ++p) for (int j = 0; j < 16; ++j) s += acc[p][j];
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    for (int i = 0; i < 4; ++i) s += r[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int N, int F16>
double run(float* d, unsigned long long* c) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<KIND, N, F16>), dim3(256), dim3(256), 0, 0, d, c, iters, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, N, F16>), dim3(256), dim3(256), 0, 0, d, c, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / iters / 16.0;      // seconds per MFMA
}

int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 256 * 256 * 4); hipMalloc(&c, 8);
    const char* kinds[3] = {"v_pk_add_f32", "ds_write_b64", "ds_read_b128"};
#define ROW(K, F) printf("%-13s %s: ns per MFMA with N side instructions per 4-MFMA chain, N = 0 4 8 16 32:  %.1f %.1f %.1f %.1f %.1f\n", kinds[K], F ? "f16 32x32x16" : "f32 32x32x2 ", \
    1e9 * run<K, 0, F>(d, c), 1e9 * run<K, 4, F>(d, c), 1e9 * run<K, 8, F>(d, c), 1e9 * run<K, 16, F>(d, c), 1e9 * run<K, 32, F>(d, c));
    ROW(0, 0) ROW(1, 0) ROW(2, 0) ROW(0, 1) ROW(1, 1) ROW(2, 1)
    printf("(one wave per SIMD, all 256 CUs; 64 / 32 matrix cycles per MFMA: at 2.4 GHz 26.7 / 13.3 ns)\n");
    return 0;
}

// How much does the ORDER of 16-B stores inside a wave matter on gfx950?  The matrix-core epilogues write PACKED rows as
// "lane = (pixel, half): four 16-B pieces of its own 64-B group" (pattern A: every store instruction touches 32 lines with
// 32 B each); pattern B writes the same bytes with lane-contiguous 1-KB instructions (8 full lines per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void pat(float* out, long tiles, int mode, int row_bytes) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    const f4 v = {1.f + lane, 2.f, 3.f, 4.f};
    for (long t = wave; t < tiles; t += nw) {
        char* base = (char*)out + t * 32L * row_bytes;          // a wave tile = 32 pixels x row_bytes (one 32-channel block = 128 B of it)
        if (mode == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f4*)(base + (long)(lane & 31) * row_bytes + (lane >> 5) * 64 + q * 16) = v;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f4*)(base + ((q * 64 + lane) >> 3) * (long)row_bytes + ((q * 64 + lane) & 7) * 16) = v;   // lane-linear inside the 128-B pieces
        }
    }
}
int main(int argc, char** argv) {
    const long tiles = 64L * 264 * 352 / 32;                  // the head tensor of 64 sequences: 761 MB at 128 B per pixel
    for (int row_bytes : {128, 256, 512}) {
        float* d; if (hipMalloc(&d, tiles * 32L * row_bytes) != hipSuccess) return 1;
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(pat, dim3(256 * 8), dim3(256), 0, 0, d, tiles, mode, row_bytes);
            hipEventRecord(a);
            for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(pat, dim3(256 * 8), dim3(256), 0, 0, d, tiles, mode, row_bytes);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
            printf("row %d B, pattern %s: %.1f us, %.2f TB/s (of the %ld MB touched)\n", row_bytes, mode ? "B lane-linear" : "A (pixel, half) groups",
                   ms * 1e3, tiles * 32.0 * 128 / (ms * 1e-3) / 1e12, tiles * 32L * 128 >> 20);
        }
        hipFree(d);
    }
    return 0;
}

// Shared helpers for libevreal_hip.so (gfx950 only; no CUDA/portable paths by design).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "evreal_hip.h"

namespace evr {
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int hip_fail(hipError_t e, const char* what, const char* file, int line);
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}  // namespace evr

#define EVR_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) return evr::hip_fail(e_, #call, __FILE__, __LINE__);     \
    } while (0)
#define EVR_REQUIRE(cond, ...)                  \
    do {                                        \
        if (!(cond)) {                          \
            evr::set_error(__VA_ARGS__);        \
            return EVR_ERR_INVALID;             \
        }                                       \
    } while (0)
#define EVR_LAUNCH_CHECK() EVR_HIP(hipGetLastError())

// wave64 helpers
__device__ __forceinline__ int evr_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ double evr_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;   // valid in lane 0
}
__device__ __forceinline__ float evr_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

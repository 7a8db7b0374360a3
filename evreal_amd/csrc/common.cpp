#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace evr {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return EVR_ERR_HIP;
}
const char* last_error() { return g_err; }
}  // namespace evr

extern "C" const char* evr_last_error(void) { return evr::last_error(); }
// 1001 (round 4): evr_percentile_normalize requires its workspace (NULL is rejected); evr_model_arith reports the effective mode
// 1002 (round 5): evr_model_desc::reserved[2] = arithmetic mode + 1 of THIS model (0: EVR_ARITH); evr_model_saturation_async
// 1003 (round 6): evr_model_release_shape; evr_png_pool_* (native PNG writers, hostcodec.cpp)
extern "C" int evr_version(void) { return 1003; }
extern "C" int evr_device_info(int device, int* n_cu, int* clock_mhz, char* name_out, size_t name_len) {
    hipDeviceProp_t p;
    EVR_HIP(hipGetDeviceProperties(&p, device));
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (clock_mhz) *clock_mhz = p.clockRate / 1000;
    if (name_out && name_len) {
        snprintf(name_out, name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    return EVR_OK;
}

// A HIP stream whose kernels may only be dispatched to `n_cus` compute units, taken evenly from every XCD (the chip
// enumerates CUs XCD-major: CU i of XCD x is bit x * cus_per_xcd + i).  evreal_amd.pipeline runs the evaluation half of a
// frame (robust normalisation, MSE/SSIM, LPIPS) on such a stream so that its small kernels stop displacing the ConvLSTM
// work-groups of the reconstruction stream on every CU.
extern "C" int evr_stream_create_cu_masked(int device, int n_cus, int from_top, evr_stream_t* out) {
    EVR_REQUIRE(out != nullptr && n_cus >= 1, "evr_stream_create_cu_masked: bad arguments");
    hipDeviceProp_t p;
    EVR_HIP(hipGetDeviceProperties(&p, device));
    const int total = p.multiProcessorCount;
    EVR_REQUIRE(n_cus <= total, "evr_stream_create_cu_masked: %d CUs requested, the device has %d", n_cus, total);
    const int n_xcd = (total % 8 == 0) ? 8 : 1, per = total / n_xcd;
    uint32_t mask[32];
    memset(mask, 0, sizeof(mask));
    EVR_REQUIRE(total <= 32 * 32, "evr_stream_create_cu_masked: %d CUs exceed the mask", total);
    for (int k = 0; k < n_cus; ++k) {
        const int x = k % n_xcd, i = k / n_xcd;
        const int cu = x * per + (from_top ? per - 1 - i : i);
        mask[cu >> 5] |= 1u << (cu & 31);
    }
    int prev = 0;
    EVR_HIP(hipGetDevice(&prev));
    EVR_HIP(hipSetDevice(device));
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return evr::hip_fail(e, "hipExtStreamCreateWithCUMask", __FILE__, __LINE__);
    *out = (evr_stream_t)s;
    return EVR_OK;
}
extern "C" int evr_stream_destroy(evr_stream_t stream) {
    if (stream) EVR_HIP(hipStreamDestroy((hipStream_t)stream));
    return EVR_OK;
}

// HIP events through the ABI: evr_model_set_gate records one inside a model step (after a named layer), another stream waits for
// it -- how evreal_amd.pipeline starts the evaluation half of frame t only once frame t+1 has passed its first ConvLSTM layer.
extern "C" int evr_event_create(evr_event_t* out) {
    EVR_REQUIRE(out != nullptr, "evr_event_create: null argument");
    hipEvent_t e = nullptr;
    EVR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *out = (evr_event_t)e;
    return EVR_OK;
}
extern "C" int evr_event_destroy(evr_event_t ev) {
    if (ev) EVR_HIP(hipEventDestroy((hipEvent_t)ev));
    return EVR_OK;
}
extern "C" int evr_stream_wait_event(evr_stream_t stream, evr_event_t ev) {
    EVR_REQUIRE(ev != nullptr, "evr_stream_wait_event: null event");
    EVR_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    return EVR_OK;
}

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
extern "C" int evr_version(void) { return 1000; }
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

// Host-only entry points that expose pieces of the split arithmetic's host side to the CPU test-suite (include/evreal_hip.h):
// the weight packer the model and LPIPS builders run (conv.h pack_split_weights) and the multiply-high division constants of
// the launch plans (conv.h fastdiv_magic).  No GPU code.
#include "conv.h"

using namespace evr;

extern "C" int evr_split_pack_weights(const float* src, float* dst, int64_t n, int* exponent) {
    EVR_REQUIRE(src && dst && exponent && n >= 0 && n % 16 == 0, "evr_split_pack_weights: n = %lld must be a multiple of 16", (long long)n);
    std::vector<float> w(src, src + n);
    *exponent = pack_split_weights(w);
    memcpy(dst, w.data(), (size_t)n * sizeof(float));
    return EVR_OK;
}

extern "C" int evr_fastdiv_magic(unsigned d, unsigned* mul, unsigned* shift) {
    EVR_REQUIRE(d >= 1 && mul && shift, "evr_fastdiv_magic: divisor must be >= 1");
    fastdiv_magic(d, mul, shift);
    return EVR_OK;
}

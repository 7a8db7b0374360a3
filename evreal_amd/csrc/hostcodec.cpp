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

// The H2 format of the fp32-grade mode (conv.h): activations (fixed exponent H2_ACT_EXP), weights (per-tensor exponent), decode.
extern "C" int evr_h2_pack(const float* src, float* dst, int64_t n) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_h2_pack: n = %lld must be a multiple of 16", (long long)n);
    std::vector<float> w(src, src + n);
    pack_h2_act(w);
    memcpy(dst, w.data(), (size_t)n * sizeof(float));
    return EVR_OK;
}
extern "C" int evr_h2_pack_weights(const float* src, float* dst, int64_t n, int* exponent) {
    EVR_REQUIRE(src && dst && exponent && n >= 0 && n % 16 == 0, "evr_h2_pack_weights: n = %lld must be a multiple of 16", (long long)n);
    std::vector<float> w(src, src + n);
    *exponent = pack_h2_weights(w);
    memcpy(dst, w.data(), (size_t)n * sizeof(float));
    return EVR_OK;
}
extern "C" int evr_h2_unpack(const float* src, float* dst, int64_t n, int exponent) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_h2_unpack: n = %lld must be a multiple of 16", (long long)n);
    unpack_h2(src, dst, (size_t)n, exponent);
    return EVR_OK;
}
extern "C" int evr_h2_act_exponent(void) { return H2_ACT_EXP; }

// The P6 format of the f16 + MX-fp6 mode (conv.h): activations, weights (swapped code order, per-tensor exponent), decode.
extern "C" int evr_p6_pack(const float* src, float* dst, int64_t n) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_p6_pack: n = %lld must be a multiple of 16", (long long)n);
    std::vector<float> w(src, src + n);
    pack_p6_act(w);
    memcpy(dst, w.data(), (size_t)n * sizeof(float));
    return EVR_OK;
}
extern "C" int evr_p6_pack_weights(const float* src, float* dst, int64_t n, int* exponent) {
    EVR_REQUIRE(src && dst && exponent && n >= 0 && n % 16 == 0, "evr_p6_pack_weights: n = %lld must be a multiple of 16", (long long)n);
    std::vector<float> w(src, src + n);
    *exponent = pack_p6_weights(w);
    memcpy(dst, w.data(), (size_t)n * sizeof(float));
    return EVR_OK;
}
extern "C" int evr_p6_unpack(const float* src, float* dst, int64_t n) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_p6_unpack: n = %lld must be a multiple of 16", (long long)n);
    unpack_p6(src, dst, (size_t)n);
    return EVR_OK;
}

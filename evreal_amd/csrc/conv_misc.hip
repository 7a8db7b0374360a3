// The convolution family's kernels that do not depend on the split arithmetic: the head convolutions (direct VALU and
// matrix-core forms), the standalone prediction layer, HyperE2VID's context / dynamic-filter kernels, bilinear upsample,
// skip-sum, format conversion -- and the dispatcher over conv.hip's three compilations (ConvArgs::x3).
#include "conv.h"
#include "packed.h"
#include <cstdlib>

namespace evr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// conv.hip is compiled three times (build.py): -DEVR_ARITH=2 carries the f16 + MX-fp8 split kernels and the exact-fp32 ones,
// -DEVR_ARITH=3 the three-f16-product kernels on H2 tensors, -DEVR_ARITH=4 the f16 + MX-fp6 kernels on P6 tensors.
int launch_conv_igemm(const ConvArgs& a, const ConvArgs* d_args, int kc, int wm, int nb, hipStream_t stream, float* img) {
    if (a.x3 == 0 && a.wgt_wino && wino_eligible(a)) return launch_conv_wino(a, d_args, stream, img);      // exact-fp32 3x3 stride-1 layers: Winograd F(2x2, 3x3)
    if (a.x3 == 3) return launch_conv_igemm_h3(a, d_args, kc, wm, nb, stream, img);
    if (a.x3 == 4) return launch_conv_igemm_m6(a, d_args, kc, wm, nb, stream, img);
    return launch_conv_igemm_mx(a, d_args, kc, wm, nb, stream, img);
}

void pick_conv_tile(const ConvArgs& a, int kc, int* wm, int* nb) {
    int n_b = (a.cout % 128 == 0) ? 4 : (a.cout % 64 == 0) ? 2 : 1;
    if (kc == 16) n_b = 1;
    if (a.epi == EPI_LSTM) n_b = 4;
    const int64_t M = (int64_t)a.n * a.hm * a.wm;
    const int ntiles = a.cout / (32 * n_b);
    // Wave-quantisation model: a CU holds bpc blocks (LDS- and register-limited); every wave does the same work
    // whatever WM is, and the waves resident on a SIMD share its matrix pipe, so
    //   time ~ rounds(WM) * waves_per_SIMD(WM),  rounds = ceil(blocks / (256 CUs * bpc)).
    // Pick the WM that minimises it (ties -> the larger tile: fewer weight re-loads).
    const int occ = (n_b == 4) ? 2 : (n_b == 2) ? 3 : 4;          // waves/SIMD the kernels' VGPR budgets allow
    int best = 4; double best_cost = 1e30;
    for (int w = 4; w >= 1; w >>= 1) {
        const int lds = 2 * (32 * w + 32 * n_b) * kc * 4;
        int bpc = (160 * 1024) / lds;
        if (bpc > (4 * occ) / w) bpc = (4 * occ) / w;
        if (bpc < 1) bpc = 1;
        const int64_t blocks = ((M + 32 * w - 1) / (32 * w)) * ntiles;
        const int64_t rounds = (blocks + 256LL * bpc - 1) / (256LL * bpc);
        const double eff = (w == 4) ? 1.0 : (w == 2) ? 0.93 : 0.85;   // smaller tiles re-load the weight tile more often
        const double cost = (double)rounds * (bpc * w / 4.0) / eff;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = w; }
    }
    if (a.epi == EPI_LSTM) {   // the ConvLSTM kernel is tuned (register staging) at WM = 4; shrink only to fill the chip
        best = 4;
        while (best > 1 && ((M + 32 * best - 1) / (32 * best)) * ntiles < 512) best >>= 1;
    }
    if (a.epi == EPI_LSTM && getenv("EVR_LSTM_WM8")) best = 8;   // experiment: 256x128 block tile
    *wm = best; *nb = n_b;
}

// ---------------------------------------------------------------------------------------------------
// Head convolution: tiny Cin (num_bins), so K = B*k*k is too ragged for the MFMA tiles; a direct
// VALU kernel with the input tile in LDS and wave-uniform weights is HBM-bound on its NHWC output.
// eval.py:398-410 from the tensorizer's {sum, sumsq, nnz}: returns false when the tensor stays as is
__device__ __forceinline__ bool norm_params(const double* stats, int n, float& mean, float& sd) {
    mean = 0.f; sd = 1.f;
    if (!stats) return false;
    const double s1 = stats[n * 3], s2 = stats[n * 3 + 1], nz = stats[n * 3 + 2];
    if (!(nz > 0.0)) return false;
    const float nf = (float)nz;
    mean = (float)s1 / nf;
    const float ex2 = (float)s2 / nf;
    sd = sqrtf(__fsub_rn(ex2, __fmul_rn(mean, mean)));
    if (sd == sd) sd = fmaxf(sd, 1e-6f);
    return true;
}
__device__ __forceinline__ float norm_apply(float v, float mean, float sd) {
    const float mask = (v != 0.f) ? 1.f : 0.f;
    return __fmul_rn(mask, __fsub_rn(v, mean)) / sd;
}

template <int K, int COUT>
__global__ __launch_bounds__(256) void head_conv_kernel(const HeadArgs a) {
    constexpr int TS = 16, IS = TS + K - 1;
    extern __shared__ float tile[];   // [B][IS][IS]
    const int n = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS, tid = threadIdx.x;

    float mean, sd;
    const bool norm = norm_params(a.stats, n, mean, sd);   // eval.py:398-410 fused into the load
    const float* vin = a.vox + (int64_t)n * a.B * a.H * a.W;
    for (int i = tid; i < a.B * IS * IS; i += 256) {
        const int b = i / (IS * IS), rr = (i / IS) % IS, cc = i % IS;
        const int y = ty0 + rr - K / 2 - a.pad_top, x = tx0 + cc - K / 2 - a.pad_left;
        float v = 0.f;
        if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) {
            v = vin[((int64_t)b * a.H + y) * a.W + x];
            if (norm) v = norm_apply(v, mean, sd);
        }
        tile[i] = v;
    }
    __syncthreads();
    const int ly = tid / TS, lx = tid % TS;
    const int oy = ty0 + ly, ox = tx0 + lx;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = a.bias[co];
    for (int b = 0; b < a.B; ++b) {
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float v = tile[(b * IS + ly + ky) * IS + lx + kx];
                const float* w = a.wgt + ((b * K + ky) * K + kx) * COUT;   // wave-uniform -> scalar loads
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(v, w[co], acc[co]);
            }
        }
    }
    if (oy < a.hp && ox < a.wp) {
        float* o = a.out + (((int64_t)n * a.hp + oy) * a.wp + ox) * COUT;
#pragma unroll
        for (int co = 0; co < COUT; co += 4) {
            float4 v = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
#if defined(__HIP_DEVICE_COMPILE__)
            if (a.out_packed) {
                const f4 t = {v.x, v.y, v.z, v.w};
                if (a.out_packed == 2) { sat_check4<2>(a.sat, t); store4_h2(o, 0u, co, t); } else { sat_check4<1>(a.sat, t); store4_packed(o, 0u, co, t); }
                continue;
            }
#endif
            *(float4*)(o + co) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Head convolution on the matrix cores (split modes, k = 5, 32 output channels; its K = 125 is too ragged for the 32-channel
// chunks of the other layers, so this kernel has its own three-f16-product loop in BOTH split modes: x = hi + lo,
// w 2^e = hi + lo in IEEE halves, acc += hi*lo + lo*hi + hi*hi -- 22 significant bits per factor): GEMM M = pixels, N = 32,
// K = (bin, ky, kx).  The direct VALU kernel above spends 4000 FMAs per pixel (0.65 ms per 64 frames, the HBM
// floor of its 761-MB output is 0.15 ms); here a pixel costs ~5 instructions per lane.
//   K order  chosen so the two lane halves of an MFMA operand differ by ONE LDS row: the 5 kernel rows are padded
//            to 6 (the 6th has zero weights) and half h takes ky = 2*kyp + h.  Element e < 75 = (b, kyp, kx),
//            8 elements per 16-k slab, 10 slabs (K = 160 incl. padding): every gather address is the lane's base
//            plus a compile-time offset -- no address arithmetic.
//   A (weights)  live in 80 VGPRs for the whole kernel, loaded in fragment order (head_pack_wfrag).
//   B (pixels)   the normalised, zero-padded input tile [B][8+5][32+4] sits in LDS as u32 = hi | lo << 16 (split once
//            per element); a lane gathers 8 elements per slab and two v_perm build the hi and lo operands.
//   C^T      lane = pixel, registers = channels -> bias, ReLU, PACKED 8-B stores.
// WLDS (round 5): the weight fragments live in LDS (20 KB, staged once per persistent block) and are read per slab instead of sitting in
// 80 registers: 236 -> ~160 VGPRs, THREE work-groups per CU instead of two.  The kernel is the one convolution of the step that is not
// power-limited (2.47 GHz, 0.20 matrix-busy, 45 % of the wave cycles parked: profiles/r05_pmc_sq_single_stream.md) -- it lacks waves to
// hide its per-tile latencies behind, not matrix cycles.
template <bool WLDS>
__global__ __launch_bounds__(256, WLDS ? 3 : 2) void head_mfma_kernel(const HeadArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = 8, TW = 32, LR = TH + 5, LC = TW + 4, PLANE = LR * LC;
    extern __shared__ __attribute__((aligned(16))) unsigned htile[];   // [B][LR][LC] (5 bins: 9360 B) | WLDS: 20 slabs x 64 lanes x 16 B of weight fragments
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;

    // weight fragments (slab s, {hi, lo}: 16 B per lane) stay in registers across ALL tiles of this persistent block;
    // the empty asm makes them opaque so hipcc neither sinks the loads into the tile loop nor re-loads them there
    const u32x4_t* wf = (const u32x4_t*)a.wfrag;
    u32x4_t w_hi[WLDS ? 1 : 10], w_lo[WLDS ? 1 : 10];
    const u32x4_t* wl = (const u32x4_t*)(htile + ((a.B * PLANE + 3) & ~3));
    if constexpr (WLDS) {
        u32x4_t* wls = (u32x4_t*)(htile + ((a.B * PLANE + 3) & ~3));
        for (int i = tid; i < 20 * 64; i += 256) wls[i] = wf[i];
    } else {
#pragma unroll
        for (int s = 0; s < 10; ++s) { w_hi[s] = wf[(2 * s) * 64 + lane]; w_lo[s] = wf[(2 * s + 1) * 64 + lane]; }
#pragma unroll
        for (int s = 0; s < 10; ++s) { asm volatile("" : "+v"(w_hi[s])); asm volatile("" : "+v"(w_lo[s])); }
    }
    f4 bias4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bias4[q] = *(const f4*)(a.bias + 8 * q + 4 * h);

    const int tiles_x = (a.wp + TW - 1) / TW, tiles_y = (a.hp + TH - 1) / TH;
    const int ntiles = a.n * tiles_y * tiles_x;
    // input tile fetch, software-pipelined: the raw values of tile t+1 are requested (all loads back to back, no
    // consumer in between) before tile t is computed and land in LDS after it
    constexpr int NL = (5 * PLANE + 255) / 256;
    float raw[NL];
    auto fetch = [&](int tile) {
        const int n = tile / (tiles_y * tiles_x), trem = tile - n * (tiles_y * tiles_x);
        const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
        const float* vin = a.vox + (int64_t)n * a.B * a.H * a.W;
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            const int i = tid + it * 256;
            const int b = i / PLANE, rem = i - b * PLANE, rr = rem / LC, cc = rem - rr * LC;
            const int y = ty0 + rr - 2 - a.pad_top, x = tx0 + cc - 2 - a.pad_left;
            const bool ok = i < a.B * PLANE && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const float v = vin[ok ? ((int64_t)b * a.H + y) * a.W + x : 0];      // (no branch around the load)
            raw[it] = ok ? v : __uint_as_float(0x7fc00001u);                  // NaN payload marks "outside"
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x), trem = tile - n * (tiles_y * tiles_x);
        const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
        float mean, sd;
        const bool norm = norm_params(a.stats, n, mean, sd);   // eval.py:398-410 fused into the load
        __syncthreads();                                       // everyone is done with the previous tile
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            const int i = tid + it * 256;
            float v = raw[it];
            const bool outside = __float_as_uint(v) == 0x7fc00001u;
            if (outside) v = 0.f; else if (norm) v = norm_apply(v, mean, sd);
            // v = hi + lo in two IEEE halves, UNSCALED: the f16 MFMA honours subnormal operands (tools/mfma_denorm_probe.hip),
            // so small inputs keep an absolute 2^-25 and the range is the half's own +-65504 (beyond: clamped and counted)
            const float c = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
            if (fabsf(v) > 65504.0f && a.sat) atomicAdd(a.sat, 1u);
            const _Float16 hh = (_Float16)c;
            const _Float16 ll = (_Float16)(c - (float)hh);
            const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, hh), lo = (unsigned)__builtin_bit_cast(unsigned short, ll);
            if (i < a.B * PLANE) htile[i] = hi | (lo << 16);
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
#pragma unroll 1
        for (int rp = 0; rp < 2; ++rp) {
            const int ty = wv * 2 + rp;
            const unsigned* base = htile + (ty + h) * LC + r;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * q + j] = bias4[q][j] * a.wfrag_inv_scale;      // products accumulate at 2^e_w
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                unsigned e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int el = 8 * s + j; if (el > 74) el = 74;                // padding slots: zero weights, any address
                    const int b = el / 15, rem = el % 15, kyp = rem / 5, kx = rem % 5;
                    e[j] = base[b * PLANE + (2 * kyp) * LC + kx];
                }
                u32x4_t ah, al;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ah[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x05040100u);
                    al[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x07060302u);
                }
                const f16x8 a_hi = __builtin_bit_cast(f16x8, ah), a_lo = __builtin_bit_cast(f16x8, al);
                const f16x8 b_hi = __builtin_bit_cast(f16x8, WLDS ? wl[(2 * s) * 64 + lane] : w_hi[WLDS ? 0 : s]);
                const f16x8 b_lo = __builtin_bit_cast(f16x8, WLDS ? wl[(2 * s + 1) * 64 + lane] : w_lo[WLDS ? 0 : s]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_hi, a_lo, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_lo, a_hi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_hi, a_hi, acc, 0, 0, 0);
                if constexpr (WLDS) { if (s & 1) __builtin_amdgcn_sched_barrier(0); }      // (two slabs' gathers in flight at most: the register budget of three work-groups per CU)
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= a.wfrag_scale;
            const int oy = ty0 + ty, ox = tx0 + r;
            if (a.relu) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = fmaxf(acc[i], 0.f);
            }
            float* o = a.out + (((int64_t)n * a.hp + oy) * a.wp + ox) * 32;
            if (a.pred_dot) {       // skip term of the fused prediction layer (model/unet.py:136-138): sum_c w[c] * head[c]
                float dot = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4 w4 = *(const f4*)(a.pred_w + 8 * q + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dot = fmaf(acc[4 * q + j], w4[j], dot);
                }
                dot += __shfl_xor(dot, 32, 64);
                if (h == 0 && oy < a.hp && ox < a.wp) a.pred_dot[((int64_t)n * a.hp + oy) * a.wp + ox] = dot;
            }
            if (a.out_packed && a.group_store) {     // the lane pair of a pixel trades runs: each stores one whole 64-B PACKED group
                float w16[16];
                xchg16(acc, w16);
                if (oy < a.hp && ox < a.wp) {
                    if (a.out_packed == 3) { sat_check16<3>(a.sat, w16); store16_p6(o, 0u, 16 * h, w16); }
                    else if (a.out_packed == 2) { sat_check16<2>(a.sat, w16); store16_h2(o, 0u, 16 * h, w16); }
                    else { sat_check16<1>(a.sat, w16); store16_packed(o, 0u, 16 * h, w16); }
                }
            } else if (oy < a.hp && ox < a.wp) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    if (a.out_packed == 2) store4_h2(o, 0u, 8 * q + 4 * h, v);
                    else if (a.out_packed) store4_packed(o, 0u, 8 * q + 4 * h, v);
                    else *(f4*)(o + 8 * q + 4 * h) = v;
                }
            }
        }
    }
#endif
}

int head_pack_wfrag(const float* w, int B, std::vector<unsigned>& out) {
    // out[((2*s + part) * 64 + lane) * 4 + d]: lane = channel c + 32*h; dword d holds k-positions 2d, 2d+1 of the lane's 8.
    // Values: w 2^e = hi + lo in two IEEE halves, e bringing max|w| to [2^13, 2^14) (conv.h pack_h2_weights' rule); returns e.
    out.assign((size_t)20 * 64 * 4, 0u);
    float mx = 0.f;
    for (int i = 0; i < B * 25 * 32; ++i) { const float a = w[i] < 0 ? -w[i] : w[i]; if (a == a && a > mx) mx = a; }
    int e = 0;
    if (mx > 0.f) { int ex; (void)__builtin_frexpf(16384.0f / mx, &ex); e = ex - 1; if (__builtin_ldexpf(mx, e) >= 16384.0f) --e; }
    if (e > 40) e = 40;
    if (e < -40) e = -40;
    for (int s = 0; s < 10; ++s)
        for (int lane = 0; lane < 64; ++lane) {
            const int c = lane & 31, h = lane >> 5;
            for (int j = 0; j < 8; ++j) {
                const int el = 8 * s + j;
                float v = 0.f;
                if (el < 15 * B && el <= 74) {
                    const int b = el / 15, rem = el % 15, ky = 2 * (rem / 5) + h, kx = rem % 5;
                    if (ky <= 4) v = w[((size_t)(b * 5 + ky) * 5 + kx) * 32 + c];
                }
                const float vs = __builtin_ldexpf(v, e);
                const unsigned short hi = f16_rne(vs), lo = f16_rne(vs - f16_to_f32(hi));
                out[((size_t)(2 * s) * 64 + lane) * 4 + j / 2] |= (unsigned)hi << (16 * (j & 1));
                out[((size_t)(2 * s + 1) * 64 + lane) * 4 + j / 2] |= (unsigned)lo << (16 * (j & 1));
            }
        }
    return e;
}

int launch_head_conv(const HeadArgs& a, hipStream_t stream) {
    EVR_REQUIRE(a.out_packed != 3 || (a.wfrag && a.k == 5 && a.cout == 32 && a.B == 5 && a.group_store),
                "head_conv: a P6 output is written by the matrix-core head kernel only (5 bins, k5, 32 channels, whole-group stores)");
    if (a.wfrag && a.k == 5 && a.cout == 32 && a.B == 5) {
        const HeadArgs& a2 = a;
        const int ntiles = a.n * ((a.hp + 7) / 8) * ((a.wp + 31) / 32);
        // persistent: the 80 weight registers (236 VGPRs in all) leave room for TWO work-groups per CU, so the grid is 512.  Measured on
        // one box (tools/head_blocks_ab.sh, two rounds): 768 work-groups (the third per CU starts when a first one ends) 421 us, 512 or
        // 1024: 382-384 us; a 168-register build that does fit three spills the weight fragments (730 us); both row passes of a wave
        // unrolled together 390 us.  EVR_HEAD_BLOCKS overrides the grid.
        static const int wlds = [] { const char* e = getenv("EVR_HEAD_WLDS"); return e ? atoi(e) : 0; }();      // (A/B: weights in LDS, three work-groups per CU)
        static const int head_blocks = [] { const char* e = getenv("EVR_HEAD_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : (wlds ? 768 : 512); }();
        const dim3 g((unsigned)(ntiles < head_blocks ? ntiles : head_blocks));
        const size_t tile_b = (((size_t)a.B * 13 * 36 + 3) & ~(size_t)3) * sizeof(unsigned);
        if (wlds) hipLaunchKernelGGL(head_mfma_kernel<true>, g, dim3(256), tile_b + (size_t)20 * 64 * 16, stream, a2);
        else hipLaunchKernelGGL(head_mfma_kernel<false>, g, dim3(256), tile_b, stream, a2);
        EVR_LAUNCH_CHECK();
        return EVR_OK;
    }
    const dim3 grid((a.wp + 15) / 16, (a.hp + 15) / 16, a.n);
    const int IS = 16 + a.k - 1;
    const size_t lds = (size_t)a.B * IS * IS * sizeof(float);
    EVR_REQUIRE(lds <= 64 * 1024, "head_conv: num_bins %d too large", a.B);
#define EVR_HEAD(K_, C_) if (a.k == K_ && a.cout == C_) { hipLaunchKernelGGL((head_conv_kernel<K_, C_>), grid, dim3(256), lds, stream, a); EVR_LAUNCH_CHECK(); return EVR_OK; }
    EVR_HEAD(5, 32) EVR_HEAD(3, 16) EVR_HEAD(3, 32) EVR_HEAD(5, 16) EVR_HEAD(3, 64)
#undef EVR_HEAD
    set_error("head_conv: unsupported kernel_size %d / base channels %d", a.k, a.cout);
    return EVR_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pred_kernel(const PredArgs a) {
    // 8 lanes per pixel, one float4 (4 channels) each per 32-channel group: a wave reads 8 pixels x 128 B
    // contiguous lines (the thread-per-pixel form over-fetched 3.7x, profiles/r01_pmc_fetch_size_nseq16.md)
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = gid >> 3;
    const int sub = (int)(gid & 7);
    const int64_t total = (int64_t)a.n * a.H * a.W;
    const bool live = i < total;
    float acc = 0.f;
    if (live) {
        const int n = (int)(i / ((int64_t)a.H * a.W));
        const int rem = (int)(i - (int64_t)n * a.H * a.W);
        const int y = rem / a.W, x = rem - y * a.W;
        const int64_t pix = ((int64_t)n * a.hp + (y + a.iy0)) * a.wp + (x + a.ix0);
        for (int c4 = sub; c4 < a.c / 4; c4 += 8) {
            float4 v = ld4_any(a.x + pix * a.c, c4 * 4, a.x_packed);
            if (a.skip) { const float4 u = ld4_any(a.skip + pix * a.c, c4 * 4, a.skip_packed); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
            const float* w = a.wgt + c4 * 4;
            acc = fmaf(v.x, w[0], acc); acc = fmaf(v.y, w[1], acc); acc = fmaf(v.z, w[2], acc); acc = fmaf(v.w, w[3], acc);
        }
    }
    acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 1, 64);
    if (live && sub == 0) {
        acc += a.bias;
        if (a.sigmoid) acc = sigmoidf_(acc);
        a.img[i] = acc;   // (prev_rec needs the un-cropped frame: models that use it keep crop == full, see model.cpp)
    }
}

int launch_pred(const PredArgs& a, hipStream_t stream) {
    EVR_REQUIRE(a.c % 4 == 0, "pred: channels %d not a multiple of 4", a.c);
    const int64_t total = (int64_t)a.n * a.H * a.W * 8;
    hipLaunchKernelGGL(pred_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

// ---------------------------------------------------------------------------------------------------
// HyperE2VID context downsample (ConvolutionalContextFusion, hyper_dynamic.py:19-23, before its conv)
__global__ __launch_bounds__(256) void ctx_down_kernel(const CtxArgs a) {
    const int ho = a.hp / 4, wo = a.wp / 4, C = a.B + 1;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)a.n * C * ho * wo;
    if (i >= total) return;
    const int ox = (int)(i % wo); int64_t p = i / wo;
    const int oy = (int)(p % ho); p /= ho;
    const int ch = (int)(p % C);
    const int n = (int)(p / C);
    float mean, sd;
    const bool norm = (ch < a.B) && norm_params(a.stats, n, mean, sd);
    auto ld = [&](int py, int px) -> float {      // padded coordinates
        if (ch == a.B) return a.prev_rec[((int64_t)n * a.hp + py) * a.wp + px];
        const int y = py - a.pad_top, x = px - a.pad_left;
        if ((unsigned)y >= (unsigned)a.H || (unsigned)x >= (unsigned)a.W) return 0.f;
        float v = a.vox[(((int64_t)n * a.B + ch) * a.H + y) * a.W + x];
        return norm ? norm_apply(v, mean, sd) : v;
    };
    // aten upsample_bilinear2d with scale 4: src = 4*o + 1.5 -> rows 4o+1, 4o+2 with lambdas 0.5/0.5
    const int y0 = 4 * oy + 1, x0 = 4 * ox + 1;
    const float v00 = ld(y0, x0), v01 = ld(y0, x0 + 1), v10 = ld(y0 + 1, x0), v11 = ld(y0 + 1, x0 + 1);
    a.out[i] = 0.5f * (0.5f * v00 + 0.5f * v01) + 0.5f * (0.5f * v10 + 0.5f * v11);
}

int launch_ctx_down(const CtxArgs& a, hipStream_t stream) {
    EVR_REQUIRE(a.hp % 4 == 0 && a.wp % 4 == 0, "ctx_down: padded size %dx%d not a multiple of 4", a.wp, a.hp);
    const int64_t total = (int64_t)a.n * (a.B + 1) * (a.hp / 4) * (a.wp / 4);
    hipLaunchKernelGGL(ctx_down_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

// Per-pixel dynamic 5x5 filtering.  One block = 8 consecutive pixels of a row; thread = channel (c <= 256,
// coalesced NHWC reads).  The pixel's 6x25 atoms are built in LDS from its 72 coefficients and the 12x25 bases.
// out_fmt != 0: the c*6 outputs of a pixel are staged in LDS and written in the packed format of the 1x1 convolution that
// consumes them (4-channel runs, st4_any) -- the separate PLAIN -> PACKED pass over the 1536-channel tensor is gone.
__global__ __launch_bounds__(256) void dynamic_filter_kernel(const float* __restrict__ x, const float* __restrict__ coeff,
                                                              const float* __restrict__ bases, float* __restrict__ out,
                                                              int n, int h, int w, int c, int out_fmt) {
    constexpr int PX = 8, NA = 6, NBAS = 12, KK = 25;
    __shared__ float sb[NBAS * KK];
    __shared__ __attribute__((aligned(16))) float atoms[PX][KK * 8];      // [tap][atom, padded 6 -> 8]: a tap's six atoms are two 16-B LDS reads
    __shared__ __attribute__((aligned(16))) float stage[2][256 * NA];
    const int tid = threadIdx.x;
    const int wblk = (w + PX - 1) / PX;
    const int bx = blockIdx.x % wblk, row = blockIdx.x / wblk;   // row over n*h
    const int img = row / h, py = row % h, px0 = bx * PX;
    for (int i = tid; i < NBAS * KK; i += 256) sb[i] = bases[i];
    __syncthreads();
    for (int i = tid; i < PX * NA * KK; i += 256) {
        const int p = i / (NA * KK), r = i % (NA * KK), m = r / KK, l = r % KK;
        float s = 0.f;
        if (px0 + p < w) {
            const float* cf = coeff + (((int64_t)img * h + py) * w + px0 + p) * (NA * NBAS) + m * NBAS;
            for (int k = 0; k < NBAS; ++k) s = fmaf(cf[k], sb[k * KK + l], s);      // einsum 'bmkhw,kl->bmlhw'
        }
        atoms[p][l * 8 + m] = s;
    }
    __syncthreads();
    for (int p = 0; p < PX && px0 + p < w; ++p) {          // (block-uniform trip count)
        const int px = px0 + p;
        float* orow = out + (((int64_t)img * h + py) * w + px) * (int64_t)(c * NA);
        for (int ch = tid; ch < c; ch += 256) {
            float acc[NA] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const int yy = py + ky - 2;
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    const int xx = px + kx - 2;
                    float v = 0.f;
                    if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) v = x[(((int64_t)img * h + yy) * w + xx) * c + ch];
                    const int l = ky * 5 + kx;
                    const float4 a0 = *(const float4*)&atoms[p][l * 8];
                    const float2 a1 = *(const float2*)&atoms[p][l * 8 + 4];
                    acc[0] = fmaf(a0.x, v, acc[0]); acc[1] = fmaf(a0.y, v, acc[1]); acc[2] = fmaf(a0.z, v, acc[2]);   // 'bmlhw,bclhw->bcmhw'
                    acc[3] = fmaf(a0.w, v, acc[3]); acc[4] = fmaf(a1.x, v, acc[4]); acc[5] = fmaf(a1.y, v, acc[5]);
                }
            }
            if (out_fmt == 0) {
#pragma unroll
                for (int m = 0; m < NA; ++m) orow[ch * NA + m] = acc[m];
            } else {
#pragma unroll
                for (int m = 0; m < NA; ++m) stage[p & 1][ch * NA + m] = acc[m];        // (c <= 256 here: one channel per thread)
            }
        }
        if (out_fmt != 0) {
            __syncthreads();      // (the other stage buffer is rewritten only after the NEXT iteration's barrier)
            for (int run = tid; run < c * NA / 4; run += 256)
                st4_any(orow, run * 4, *(const float4*)&stage[p & 1][run * 4], out_fmt);
        }
    }
}

int launch_dynamic_filter(const float* x, const float* coeff, const float* bases, float* out, int n, int h, int w, int c,
                          hipStream_t stream, int out_fmt) {
    EVR_REQUIRE(out_fmt == 0 || (c <= 256 && (c * 6) % 16 == 0), "dynamic_filter: packed output needs c <= 256 and c*6 %% 16 == 0 (c = %d)", c);
    const int wblk = (w + 7) / 8;
    hipLaunchKernelGGL(dynamic_filter_kernel, dim3((unsigned)(wblk * n * h)), dim3(256), 0, stream, x, coeff, bases, out, n, h, w, c, out_fmt);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

// ---------------------------------------------------------------------------------------------------
// F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) of (x + skip), NHWC.
__global__ __launch_bounds__(256) void upsample2x_sum_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                              float* __restrict__ out, int n, int h, int w, int c,
                                                              int x_packed, int skip_packed, int out_packed) {
    const int c4n = c / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n * 2 * h * 2 * w * c4n;
    if (i >= total) return;
    const int c4 = (int)(i % c4n);
    int64_t p = i / c4n;
    const int ox = (int)(p % (2 * w)); p /= 2 * w;
    const int oy = (int)(p % (2 * h));
    const int img = (int)(p / (2 * h));
    // aten area_pixel_compute_source_index: src = 0.5*(dst+0.5)-0.5, clamped at 0
    float sy = 0.5f * (oy + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = 0.5f * (ox + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + ((y0 < h - 1) ? 1 : 0), x1 = x0 + ((x0 < w - 1) ? 1 : 0);
    const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    auto ld = [&](int yy, int xx) {
        const int64_t o = (((int64_t)img * h + yy) * w + xx) * c;
        float4 v = ld4_any(x + o, c4 * 4, x_packed);
        if (skip) { const float4 u = ld4_any(skip + o, c4 * 4, skip_packed); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        return v;
    };
    const float4 v00 = ld(y0, x0), v01 = ld(y0, x1), v10 = ld(y1, x0), v11 = ld(y1, x1);
    float4 o4;
    o4.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o4.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o4.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o4.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    st4_any(out + (((int64_t)img * 2 * h + oy) * 2 * w + ox) * c, c4 * 4, o4, out_packed);
}

// The same on P6 tensors: one thread per (output pixel, 16-channel group) -- a P6 group is written whole (its scale is the
// group's maximum) and decodes whole with one conversion instruction (packed.h load16_p6 / store16_p6).
__global__ __launch_bounds__(256) void upsample2x_sum_p6_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                                 float* __restrict__ out, int n, int h, int w, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int gn = c / 16;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n * 2 * h * 2 * w * gn;
    if (i >= total) return;
    const int g = (int)(i % gn);
    int64_t p = i / gn;
    const int ox = (int)(p % (2 * w)); p /= 2 * w;
    const int oy = (int)(p % (2 * h));
    const int img = (int)(p / (2 * h));
    float sy = 0.5f * (oy + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = 0.5f * (ox + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + ((y0 < h - 1) ? 1 : 0), x1 = x0 + ((x0 < w - 1) ? 1 : 0);
    const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    float v[4][16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yy = (k & 2) ? y1 : y0, xx = (k & 1) ? x1 : x0;
        const int64_t o = (((int64_t)img * h + yy) * w + xx) * c + g * 16;
        load16_p6(x + o, v[k]);
        if (skip) {
            float u[16];
            load16_p6(skip + o, u);
#pragma unroll
            for (int j = 0; j < 16; ++j) v[k][j] += u[j];
        }
    }
    float o16[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) o16[j] = hy * (hx * v[0][j] + lx * v[1][j]) + ly * (hx * v[2][j] + lx * v[3][j]);
    store16_p6(out + (((int64_t)img * 2 * h + oy) * 2 * w + ox) * c, 0u, g * 16, o16);
#endif
}

int launch_upsample2x_sum(const float* x, const float* skip, float* out, int n, int h, int w, int c, int x_packed, int skip_packed, int out_packed, hipStream_t stream) {
    if (out_packed == 3) {
        EVR_REQUIRE(c % 16 == 0 && x_packed == 3 && (!skip || skip_packed == 3), "upsample: a P6 output takes P6 inputs (%d / %d), channels %d a multiple of 16", x_packed, skip_packed, c);
        const int64_t total = (int64_t)n * 4 * h * w * (c / 16);
        hipLaunchKernelGGL(upsample2x_sum_p6_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, skip, out, n, h, w, c);
        EVR_LAUNCH_CHECK();
        return EVR_OK;
    }
    EVR_REQUIRE(c % 4 == 0, "upsample: channels %d not a multiple of 4", c);
    EVR_REQUIRE(!(x_packed || skip_packed || out_packed) || c % 16 == 0, "upsample: PACKED tensors need channels %d to be a multiple of 16", c);
    const int64_t total = (int64_t)n * 4 * h * w * (c / 4);
    hipLaunchKernelGGL(upsample2x_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, skip, out, n, h, w, c, x_packed, skip_packed, out_packed);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ o, int64_t n4, int packed) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        // 4-channel run i: PACKED tensors are addressed per 16-channel group (the row base is the group, c4 = 0..12)
        const int64_t base = packed ? (i >> 2) * 16 : i * 4;
        const int c4 = packed ? (int)(i & 3) * 4 : 0;
        const float4 a = ld4_any(x + base, c4, packed), b = ld4_any(y + base, c4, packed);
        st4_any(o + base, c4, make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w), packed);
    }
}

int launch_add(const float* x, const float* y, float* out, int64_t n, int packed, hipStream_t stream) {
    EVR_REQUIRE(n % 16 == 0, "add: element count not a multiple of 16");
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y, out, n / 4, packed);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}
// PLAIN -> PACKED, one 16-channel group per thread (may run in place: a thread reads its 64 B before it writes them)
__global__ __launch_bounds__(256) void to_packed_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t groups, int fmt) {
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
        const float4* sp = (const float4*)(src + g * 16);
        const float4 v0 = sp[0], v1 = sp[1], v2 = sp[2], v3 = sp[3];
        float* d = dst + g * 16;
#if defined(__HIP_DEVICE_COMPILE__)
        if (fmt == 3) {      // P6: the group's scale needs all 16 values
            const float w[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
            store16_p6(d, 0u, 0, w);
            continue;
        }
#endif
        st4_any(d, 0, v0, fmt); st4_any(d, 4, v1, fmt); st4_any(d, 8, v2, fmt); st4_any(d, 12, v3, fmt);
    }
}
int launch_to_packed(const float* src, float* dst, int64_t n, hipStream_t stream, int fmt) {
    EVR_REQUIRE(n % 16 == 0, "to_packed: element count not a multiple of 16");
    int64_t blocks = (n / 16 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(to_packed_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, n / 16, fmt);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}


__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int h, int w, int c, int packed, int cs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n * h * w * c;
    if (i >= total) return;
    const int x = (int)(i % w); int64_t p = i / w;
    const int y = (int)(p % h); p /= h;
    const int ch = (int)(p % c);
    const int img = (int)(p / c);
    const float* row = src + (((int64_t)img * h + y) * w + x) * cs;
    float v = row[ch];
#if defined(__HIP_DEVICE_COMPILE__)
    if (packed) v = load1_packed(row, ch, packed);
#endif
    dst[i] = v;
}

int launch_nhwc_to_nchw(const float* src, float* dst, int n, int h, int w, int c, int packed, hipStream_t stream, int c_stride) {
    if (c_stride <= 0) c_stride = c;
    EVR_REQUIRE(!packed || c_stride % 16 == 0, "nhwc_to_nchw: PACKED tensor with %d channels", c_stride);
    const int64_t total = (int64_t)n * h * w * c;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, dst, n, h, w, c, packed, c_stride);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

}  // namespace evr

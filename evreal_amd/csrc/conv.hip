// Convolutions of the recurrent networks on the matrix cores of gfx950: an implicit GEMM (this header) and, for the
// 3x3 stride-1 layers in split mode, the band kernels further down (conv3x3_band_kernel, conv3x3_wide_kernel).
//
// Reference ops (model/submodules.py): ConvLayer :8-35, TransposedConvLayer :38-66,
// UpsampleConvLayer :69-97, ResidualBlock :152-184, ConvLSTM :187-245, ConvGRU :248-287; wired by
// model/unet.py:107-143 and model/legacy.py:79-111.  The reference dispatches them to aten conv2d /
// conv_transpose2d on NCHW tensors; here every contraction is ONE kernel family:
//
//   GEMM view   M = n*hm*wm output pixels, N = output channels, K = taps x input channels
//   data        activations NHWC fp32 (a K chunk of 16/32 channels of one pixel = one 64/128-B line);
//               weights pre-laid as [N][K] (K contiguous) at model creation, BatchNorm folded
//   MFMA        the arithmetic modes on the same tiles (template X3 = fp32 or split; EVR_ARITH picks the split):
//               fp32   v_mfma_f32_32x32x2_f32, an exact fp32 fma chain (157 TF peak) -- FireNet (16-channel chunks)
//                      and the reference mode (EVR_FP32=1);
//               split  (conv.h) x = hi + lo8 2^-12, w = hi + wlo8 2^-(e+12) with f16 'hi' halves (RNE) and fp8 e4m3
//                      residuals: acc += hi_w*hi_x on v_mfma_f32_32x32x16_f16, and the two cross terms w8*lo8 + wlo8*x8
//                      (w8, x8 = fp8 copies of the values) on ONE MX-scaled v_mfma_scale_f32_32x32x64_f8f6f4 per 32 k,
//                      fp32 accumulation: 2 x 32 + 64 matrix cycles per 32 k instead of 16 x 64 (8x fewer), and 2/3 of
//                      the three-bf16-product split this mode replaced.  Every product term carries a relative error
//                      of ~2^-16; measured through a 60-frame recurrence the image error is 6e-5 of a range-3 image
//                      (tools/split_scheme_sim.py; gate 1e-4; three bf16 products 1.7e-5, plain bf16 9e-3).
//                      Weights are pre-split on the host, activations by the PRODUCING kernel's epilogue (PACKED
//                      format): the main loop feeds 16-B LDS slots straight to the MFMAs
//               split6 (EVR_ARITH = 4, conv.h P6) the same with the cross terms in e2m3 (fp6) and one E8M0 scale per 16-channel
//                      group: both operands fp6 -> the MX MFMA takes 8 passes instead of 16, 2 x 32 + 32 = 96 matrix cycles
//                      per 32 k; the default for the layouts whose packed tensors are written as whole groups
//               h3     (EVR_ARITH = 3, conv.h H2) three f16 products per term: fp32-grade, 192 matrix cycles per 32 k
//   tile        block = WM waves stacked along M; a wave owns 32 pixels x (NB*32) channels, i.e. NB
//               accumulators of 16 VGPRs; for ConvLSTM NB = 4 and the weight rows are permuted so
//               the four 32-column blocks are the in/remember/out/cell gates of the SAME 32 hidden
//               channels -> the LSTM cell update is a pure register epilogue
//   LDS         per K step: A tile [32*WM][KC] + B tile [32*NB][KC], double buffered, 16-B slots
//               XOR-swizzled by row so both the ds_write_b128 staging and the ds_read_b128 fragment
//               reads are bank-conflict free; a float4 read hands a lane 4 consecutive k, which feed
//               4 MFMAs (the k order inside an MFMA pair is free as long as A and B agree)
//   im2col      on the fly: per K step one tap (dy,dx) and one channel chunk go HBM/L2 -> LDS by
//               `buffer_load_dwordx4 ... lds` (LDS-DMA, no VGPR staging); zero padding and ragged M
//               come from the buffer descriptor's range check (out-of-range offset reads 0); channel
//               concat (x|h) switches descriptors per chunk; ConvTranspose2d(k,s=2) is ONE GEMM whose N
//               is phase-major (4 sub-pixel phases x Cout): the phases share the 3x3 input taps (A tile
//               loaded once), and (tap, phase) pairs the transposed kernel does not connect are skipped;
//               skip-sum is fused into the producer
//   epilogues   bias/ReLU, residual+ReLU, ConvLSTM cell, ConvGRU (update/reset, candidate+blend)
//   grid        1-D, remapped so every XCD (private L2) walks a contiguous range of tiles with the N
//               tile fastest: an A tile is reused from L2 across its N tiles and neighbouring M tiles
#include "conv.h"
#include "packed.h"
#include <atomic>
#include <cstdlib>
#include <cstring>

// This file is compiled once per split arithmetic (build.py): EVR_ARITH = 2 -- f16 + MX-fp8 on PACKED tensors, plus the
// exact-fp32 kernels -- EVR_ARITH = 3 -- three f16 products on H2 tensors -- and EVR_ARITH = 4 -- f16 + MX-fp6 on P6 tensors, the
// default where the layout allows (conv.h).  Kernels live in their own inner namespace so the objects do not collide;
// conv_misc.hip dispatches on ConvArgs::x3.
#ifndef EVR_ARITH
#define EVR_ARITH 2
#endif
#if EVR_ARITH == 3
#define EVR_ANS h3
#define EVR_LAUNCH_NAME launch_conv_igemm_h3
#elif EVR_ARITH == 4
#define EVR_ANS m6
#define EVR_LAUNCH_NAME launch_conv_igemm_m6
#else
#define EVR_ANS mx
#define EVR_LAUNCH_NAME launch_conv_igemm_mx
#endif

namespace evr {
namespace EVR_ANS {
[[maybe_unused]] constexpr int ARITH = EVR_ARITH;
[[maybe_unused]] constexpr int FMT = (EVR_ARITH == 3) ? 2 : (EVR_ARITH == 4 ? 3 : 1);      // format of this build's packed tensors (packed.h)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// Split arithmetic (conv.h), one 32 x 32 block x 32 k.  A PACKED row chunk of 32 k is 8 slots of 16 B:
//   slots 0,1 = f16 hi of k 0-15, slot 2 = fp8 lo8 (w8) of k 0-15, slot 3 = fp8 x8 (wlo8) of k 0-15, slots 4-7 = k 16-31.
// Lane half h takes slot 4s + h for the f16 MFMA of group s, and slots 2 + h, 6 + h for the fp8 MFMA: its 32 k-values
// there are [lo8 of k 0-31] (h = 0) against [w8] or [x8 of k 0-31] (h = 1) against [wlo8].
struct SplitFrag { u32x4_t h0, h1, f0, f1; unsigned sc; };      // (sc: mode 4 only -- the group's E8M0 scale byte)
__device__ __forceinline__ SplitFrag ld_split(const float4* row, int h, int sw) {
    SplitFrag f;
    f.h0 = __builtin_bit_cast(u32x4_t, row[(h) ^ sw]);
    f.h1 = __builtin_bit_cast(u32x4_t, row[(4 + h) ^ sw]);
    if constexpr (ARITH == 4) {
        // P6 rows: slots 2,3 of a group = its 32 e2m3 codes (24 B) | scale byte | spare: lane half h takes the codes of group h whole
        // (a 64-bit + 32-bit read of slot 3 -- which hipcc fuses into ds_read_b96 -- would spare the two v_mov per fragment that put
        // f1[0..1] next to f0, but measured 3-6 % slower on every layer: the 96-bit LDS read is the slow one)
        f.f0 = __builtin_bit_cast(u32x4_t, row[(4 * h + 2) ^ sw]);
        f.f1 = __builtin_bit_cast(u32x4_t, row[(4 * h + 3) ^ sw]);
        f.sc = f.f1[2];
    } else {
        f.sc = 0u;
        f.f0 = __builtin_bit_cast(u32x4_t, row[(2 + h) ^ sw]);
        f.f1 = __builtin_bit_cast(u32x4_t, row[(6 + h) ^ sw]);
    }
    return f;
}
__device__ __forceinline__ i32x8 cat8(u32x4_t p, u32x4_t q) {
    typedef unsigned u32x8_t __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(i32x8, (u32x8_t)__builtin_shufflevector(p, q, 0, 1, 2, 3, 4, 5, 6, 7));   // a register sequence, no copies
}
// acc += W . X over the chunk; the weights are the instruction's first operand (acc holds C^T, see the epilogue)
__device__ __forceinline__ f32x16 mma_split(f32x16 acc, const SplitFrag& w, const SplitFrag& x, int sc_w, int sc_x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (ARITH == 3) {
        // H2 rows: slots 0,1 = f16 hi of k 0-15, slots 2,3 = f16 lo of k 0-15, slots 4-7 = k 16-31 -- the same four slots per
        // lane half (f0 / f1 now hold the lo halves of the k-values in h0 / h1).  hi hi + hi lo + lo hi; lo lo (2^-22) dropped
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.f0), __builtin_bit_cast(f16x8, x.h0), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.h0), __builtin_bit_cast(f16x8, x.f0), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.h0), __builtin_bit_cast(f16x8, x.h0), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.f1), __builtin_bit_cast(f16x8, x.h1), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.h1), __builtin_bit_cast(f16x8, x.f1), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.h1), __builtin_bit_cast(f16x8, x.h1), acc, 0, 0, 0);
    } else if constexpr (ARITH == 4) {
        // f16 hi hi + ONE fp6 (e2m3) MFMA for both cross terms of the lane half's group: 8 passes instead of fp8's 16; the six code
        // registers are f0 | f1[0..1], the group's E8M0 scale byte sits in f1[2] (byte 0, op_sel 0) -- per lane, i.e. per (row, group)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.h0), __builtin_bit_cast(f16x8, x.h0), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.h1), __builtin_bit_cast(f16x8, x.h1), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(w.f0, w.f1), cat8(x.f0, x.f1), acc, 2, 2, 0, (int)w.sc, 0, (int)x.sc);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.h0), __builtin_bit_cast(f16x8, x.h0), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.h1), __builtin_bit_cast(f16x8, x.h1), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(w.f0, w.f1), cat8(x.f0, x.f1), acc, 0, 0, 0, sc_w, 0, sc_x);
    }
#endif
    return acc;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// n / d for n < 2^31 with the multiplier of conv.h fastdiv_magic
__device__ __forceinline__ int fdiv(int n, unsigned mul, unsigned sh) { return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> sh); }
// Gate activations.  FAST (split mode): v_exp_f32 / v_rcp_f32 forms, absolute error ~2e-7 -- an order below the
// mode's own 2^-16 input rounding; the exact-fp32 mode keeps the libm-grade functions.
template <bool FAST> __device__ __forceinline__ float sigmoid_t(float x) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    else return 1.0f / (1.0f + expf(-x));
}
template <bool FAST> __device__ __forceinline__ float tanh_t(float x) {
    if constexpr (FAST) return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
    else return tanhf(x);
}

template <int KC>
__device__ __forceinline__ int swz(int row) {
    return (KC == 32) ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

// __amdgpu_buffer_rsrc_t and the LDS-DMA builtins exist only in the device pass of hipcc; the host pass
// just needs the kernel's signature to emit the launch stub, so the body is compiled for the device only.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned OOB_OFFSET = 0xFFFFFFF0u;   // >= num_records of every descriptor -> the load returns 0

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}

// ---------------------------------------------------------------------------------------------------
// Epilogue shared by the convolution kernels.
//
// Accumulator layout (the MFMAs run with the WEIGHT fragment as the A operand, so acc holds C^T): lane & 31 = pixel
// (GEMM row m), register `reg` of 32-column block nb = channel n0 + nb*32 + (reg&3) + 8*(reg>>2) + 4*h, h = lane>>5.
// A lane owns ONE pixel and, per block, four runs of 4 consecutive channels: every operand load and result store
// is a 16-B access, the address arithmetic happens once per lane, and the fused prediction layer reduces over
// channels in registers.  epi_setup (before the main loop) starts the accumulators at the bias and requests the
// epilogue's operands (cell state / residual / fused skip) a whole main loop before they are needed.
// timing ablation of the epilogues' memory traffic (tools/ablate_wide.sh; results are garbage): bit 4 (16) no operand loads, bit 6 (64) no stores
#ifdef EVR_WIDE_ABLATE
constexpr int EPI_ABLATE = EVR_WIDE_ABLATE;
#else
constexpr int EPI_ABLATE = 0;
#endif
struct EpiCtx {
    int m; bool mvalid, direct;
    int e_img, e_my, e_mx;      // decoded GEMM row (only when the output pixel is not the row itself)
    unsigned lstm_o;            // ConvLSTM: element offset of the lane's first hidden channel
};
// (row = element offset of the pixel row, c4 = first of the lane's 4 channels; n_valid and cout_total are multiples
// of 4 -- checked at launch -- so a run is never ragged)
__device__ __forceinline__ f4 ld4(const float* p, unsigned row, int c4, int packed) {
    if (packed) return load4_fmt<FMT>(p, row, c4);
    return *(const f4*)(p + row + (unsigned)c4);
}
__device__ __forceinline__ void st4(float* p, unsigned row, int c4, f4 v, int packed) {
    if (packed) store4_fmt<FMT>(p, row, c4, v);
    else *(f4*)(p + row + (unsigned)c4) = v;
}
// column group (sub-pixel phase) of the 32-column block that starts at GEMM column `col`, and the block's first channel inside
// that group.  Phase-major N (tp.inter == 0): group = col / grp_cols.  Tile-interleaved N (tp.inter == 1, round 5: transposed
// decoders with more than 32 output channels): a 128-column tile holds the four phases of 32 channels -- with whole groups per
// tile, three of four tiles of dec0 walked steps whose taps their phase never uses (barrier, weight DMA and all)
__device__ __forceinline__ int col_group(const ConvTaps& tp, int col) { return tp.inter ? ((col >> 5) & 3) : col / tp.grp_cols; }
__device__ __forceinline__ int col_chan(const ConvTaps& tp, int col, int g) { return tp.inter ? (((col >> 7) << 5) + (col & 31)) : col - g * tp.grp_cols; }
// output pixel and first channel (inside its column group) of the lane in 32-column block nb
template <bool GROUPED>
__device__ __forceinline__ void out_addr(const ConvArgs& a, const EpiCtx& ec, int n0, int h, int nb, unsigned& opx, int& cgb, int& oy, int& ox) {
    const int g = GROUPED ? col_group(a.tp, n0 + nb * 32) : 0;
    cgb = (GROUPED ? col_chan(a.tp, n0 + nb * 32, g) : n0 + nb * 32) + 4 * h;
    oy = ec.e_my * a.os + a.tp.grp_ofy[g]; ox = ec.e_mx * a.os + a.tp.grp_ofx[g];
    opx = ec.direct ? (unsigned)ec.m : (unsigned)((ec.e_img * a.hout + oy) * a.wout + ox);
}

template <int NB, bool LSTM, bool GROUPED>
__device__ __forceinline__ void epi_prefetch(const ConvArgs& a, int n0, int h, f32x16 (&pre)[LSTM ? 1 : NB], const EpiCtx& ec);

template <int NB, bool LSTM, bool GROUPED>
__device__ __forceinline__ void epi_setup(const ConvArgs& a, int m, int M, int hw, int n0, int h, f32x16 (&acc)[NB],
                                          f32x16 (&pre)[LSTM ? 1 : NB], EpiCtx& ec, bool lane_ok = true, bool prefetch = true) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4 b4 = *(const f4*)(a.bias + n0 + nb * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[nb][4 * q + j] = b4[j];
        }
    ec.m = m; ec.mvalid = lane_ok && m < M;
    ec.direct = (a.os == 1 && a.hout == a.hm && a.wout == a.wm);
    ec.e_img = 0; ec.e_my = 0; ec.e_mx = 0;
    if (!ec.direct || a.pred_w) {
        const int mm = ec.mvalid ? m : 0;
        ec.e_img = fdiv(mm, a.div_hw_mul, a.div_hw_sh);
        const int rem = mm - ec.e_img * hw;
        ec.e_my = fdiv(rem, a.div_w_mul, a.div_w_sh); ec.e_mx = rem - ec.e_my * a.wm;
    }
    ec.lstm_o = (unsigned)(ec.mvalid ? m : 0) * (unsigned)a.hidden + (unsigned)((n0 >> 2) + 4 * h);
    if (prefetch) epi_prefetch<NB, LSTM, GROUPED>(a, n0, h, pre, ec);
}

// the epilogue's operand loads (cell state / residual / fused skip) into `pre`; see epi_setup
template <int NB, bool LSTM, bool GROUPED>
__device__ __forceinline__ void epi_prefetch(const ConvArgs& a, int n0, int h, f32x16 (&pre)[LSTM ? 1 : NB], const EpiCtx& ec) {
    const int epi = a.epi;
    const unsigned ct = (unsigned)a.cout_total;
    const int nvalid = a.n_valid;
    const bool gru = (epi == EPI_GRU_ZR || epi == EPI_GRU_OUT);
    const bool res = (epi == EPI_RESIDUAL_RELU);
    constexpr int PN = LSTM ? 1 : NB;
    const float* pre_ptr = LSTM ? a.state : (gru ? nullptr : (res ? a.residual : a.post_add));
    if (pre_ptr && !(a.debug_ablate & 16) && !(EPI_ABLATE & 16)) {   // (bit 4 of EVR_ABLATE: timing without the operand loads)
        // The loads are issued back to back and their RAW bits parked in `pre` (PACKED operands are decoded in
        // epi_finish): anything that consumes a value here, or a per-lane branch around a load, makes hipcc wait
        // for each load in turn -- 16+ serialised L2 round trips per lane.  Lanes past the end of M read row 0.
        const int pk = LSTM ? 0 : (res ? a.res_packed : a.padd_packed);
#pragma unroll
        for (int nb = 0; nb < PN; ++nb) {
            unsigned opx; int cgb, oy, ox;
            out_addr<GROUPED>(a, ec, n0, h, nb, opx, cgb, oy, ox);
            const unsigned row = ec.mvalid ? opx * ct : 0u;
            if constexpr (!LSTM) {
                if (pk && a.group_store) {      // a PACKED operand: the lane's whole group (3 x 16 B), exchanged in epi_finish
                    const int cg = cgb - 4 * h + 16 * h;
                    f4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, g2 = g0, g3 = g0;
                    if (cg < nvalid) {               // (a 16-channel tail block: only the lower half of the lane pair loads)
                        const f4* gp = (const f4*)(pre_ptr + row + (unsigned)cg);
                        g0 = gp[0]; g1 = gp[1]; g2 = gp[2];
                        if constexpr (FMT >= 2) g3 = gp[3];      // H2: hi 32 B | lo 32 B; P6: the codes' tail and the scale byte
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { pre[nb][j] = g0[j]; pre[nb][4 + j] = g1[j]; pre[nb][8 + j] = g2[j]; pre[nb][12 + j] = g3[j]; }
                    continue;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4 v = {0.f, 0.f, 0.f, 0.f};
                if constexpr (LSTM) {
                    v = *(const f4*)(pre_ptr + ec.lstm_o + 8 * q);    // c_prev of the lane's 16 hidden channels
                } else {
                    const int c4 = cgb + 8 * q;
                    if (c4 - 4 * h < nvalid) {                        // wave-uniform (n_valid is a multiple of 8 or the run is whole)
                        if (pk && FMT == 3) {
                            v = load4_fmt<FMT>(pre_ptr, row, c4);      // (P6 off the group path: decoded here, not parked raw)
                        } else if (pk) {
                            const float* qp = pre_ptr + pk_off(row, c4);
                            f4 hi_lo = {qp[0], qp[1], 0.f, 0.f};                   // 8-B hi piece + ...
                            if constexpr (FMT == 2) { hi_lo[2] = qp[8]; hi_lo[3] = qp[9]; }      // ... the 8-B lo piece 32 B on (H2)
                            else hi_lo[2] = pre_ptr[pk_lo_off(row, c4)];                        // ... the 4-B lo8 piece
                            v = hi_lo;
                        } else {
                            v = *(const f4*)(pre_ptr + row + (unsigned)c4);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) pre[nb][4 * q + j] = v[j];
            }
        }
    }
}

// ConvGRU operands of conv3x3_c16_rows_kernel (one 32-column block, hidden == 16), requested a step ahead and parked RAW in `pre`
// (epi_finish<..., GRUPRE = true> decodes them): h_prev of the lane's two real runs in slots 2, 3 -- for EPI_GRU_ZR those are the
// reset-gate runs q = 2, 3 themselves, for EPI_GRU_OUT the runs q = 0, 1, whose update gate z sits in slots 0, 1
__device__ __forceinline__ void gru_prefetch16(const ConvArgs& a, int h, f32x16& pre, const EpiCtx& ec) {
    const unsigned row = (unsigned)(ec.mvalid ? ec.m : 0) * 16u;
    const bool zr = a.epi == EPI_GRU_ZR;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ch = 8 * q + 4 * h;
        f4 hp;
        if (a.state_packed) {
            const float* qp = a.state + pk_off(row, ch);
            hp = f4{qp[0], qp[1], 0.f, 0.f};
            if constexpr (FMT == 2) { hp[2] = qp[8]; hp[3] = qp[9]; } else hp[2] = a.state[pk_lo_off(row, ch)];
        } else hp = *(const f4*)(a.state + row + (unsigned)ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) pre[4 * (q + 2) + j] = hp[j];
        if (!zr) {
            const f4 z = *(const f4*)(a.aux0 + row + (unsigned)ch);
#pragma unroll
            for (int j = 0; j < 4; ++j) pre[4 * q + j] = z[j];
        }
    }
}
__device__ __forceinline__ f4 gru_parked_h(const ConvArgs& a, const f32x16& pre, int slot) {
    f4 pv = {pre[4 * slot], pre[4 * slot + 1], pre[4 * slot + 2], pre[4 * slot + 3]};
    if (a.state_packed) {
        const uint2 phi = {__float_as_uint(pv[0]), __float_as_uint(pv[1])};
        if constexpr (FMT == 2) { const uint2 plo = {__float_as_uint(pv[2]), __float_as_uint(pv[3])}; pv = unpack4_h2(phi, plo); }
        else if constexpr (FMT != 3) pv = unpack4(phi, __float_as_uint(pv[2]));
    }
    return pv;
}

template <int NB, bool LSTM, bool GROUPED, bool FAST, bool GRUPRE = false>
__device__ __forceinline__ void epi_finish(const ConvArgs& a, const EpiCtx& ec, int n0, int h, f32x16 (&acc)[NB],
                                           f32x16 (&pre)[LSTM ? 1 : NB], float* __restrict__ img_out) {
    const int epi = a.epi;
    const int m = ec.m;
    const bool mvalid = ec.mvalid;
    if constexpr ((ARITH == 3 || ARITH == 4) && FAST) {      // (FAST = a split kernel) products were accumulated at 2^(e_w + H2_ACT_EXP) / 2^e_w
        const float sc = a.acc_scale;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][i] *= sc;
    }
    if constexpr (LSTM) {
        static_assert(!LSTM || NB == 4, "the ConvLSTM epilogue needs the four gates in one wave tile");
        // N tile of 128 = 4 gates x 32 hidden channels (rows permuted at model creation); the conv is stride 1 on
        // the state's own grid, so the output pixel is the GEMM row.  submodules.py:227-245
        if (!mvalid) return;      // (both halves of a pixel leave together: the lane exchange below stays paired)
        f32x16 hall;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f4 cn, hn;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gi = sigmoid_t<FAST>(acc[0][4 * q + j]);
                const float gf = sigmoid_t<FAST>(acc[1][4 * q + j]);
                const float go = sigmoid_t<FAST>(acc[2][4 * q + j]);
                const float gc = tanh_t<FAST>(acc[3][4 * q + j]);
                cn[j] = __fadd_rn(__fmul_rn(gf, pre[0][4 * q + j]), __fmul_rn(gi, gc));   // submodules.py:242
                hn[j] = go * tanh_t<FAST>(cn[j]);                                          // submodules.py:243
                hall[4 * q + j] = hn[j];
            }
            if (!(EPI_ABLATE & (64 | 256)) || cn[0] == 123.456f) *(f4*)(a.state + ec.lstm_o + 8 * q) = cn;      // (bit 8: the cell-state store alone -- a clean ablation: c feeds no matrix operand)                  // the cell state stays fp32 (never a GEMM operand)
            if (!a.out_packed) *(f4*)(a.out + ec.lstm_o + 8 * q) = hn;
            else if (!a.group_store) store4_fmt<FMT>(a.out, (unsigned)m * (unsigned)a.hidden, (n0 >> 2) + 4 * h + 8 * q, hn);
        }
        if (a.out_packed && a.group_store) {       // the wave's 32 hidden channels of this pixel = two PACKED groups, one per lane of the pair
            float w16[16];
            xchg16(hall, w16);
            if (!(EPI_ABLATE & 64) || w16[0] == 123.456f) store16_fmt<FMT>(a.out, (unsigned)m * (unsigned)a.hidden, (n0 >> 2) + 16 * h, w16);
        }
        return;
    } else {
        const unsigned ct = (unsigned)a.cout_total;
        const int nvalid = a.n_valid;
        const bool gru = (epi == EPI_GRU_ZR || epi == EPI_GRU_OUT);
        const int grp_cols = a.tp.grp_cols;
        // Each epilogue kind is its OWN loop nest under one block-uniform branch: with every kind inside a single
        // (nb, q) body the unrolled epilogue was 11k instructions of which a launch executes a few hundred, spread
        // over the whole range (instruction-cache misses on every iteration).
        if (gru) {
            // ConvGRU (submodules.py:281-285), hidden % 4 == 0 checked at launch
            const int C = a.hidden;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                unsigned opx; int cgb, oy, ox;
                out_addr<GROUPED>(a, ec, n0, h, nb, opx, cgb, oy, ox);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n4 = n0 + nb * 32 + 8 * q + 4 * h;   // GEMM column of the run
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[nb][4 * q + j];
                    if (!mvalid) continue;
                    if (epi == EPI_GRU_ZR) {
                        if (n4 < C) {
                            f4 z;
#pragma unroll
                            for (int j = 0; j < 4; ++j) z[j] = sigmoid_t<FAST>(v[j]);
                            *(f4*)(a.aux0 + opx * (unsigned)C + n4) = z;                      // update gate z
                        } else if (n4 < 2 * C) {
                            f4 hp;
                            if constexpr (GRUPRE) hp = gru_parked_h(a, pre[nb], q); else hp = ld4(a.state, opx * (unsigned)C, n4 - C, a.state_packed);
                            f4 hr;
#pragma unroll
                            for (int j = 0; j < 4; ++j) hr[j] = hp[j] * sigmoid_t<FAST>(v[j]);
                            st4(a.out, opx * (unsigned)C, n4 - C, hr, a.out_packed);      // h * reset
                        }
                    } else if (n4 < C) {
                        const unsigned o = opx * (unsigned)C + (unsigned)n4;
                        f4 z, hp;
                        if constexpr (GRUPRE) { z = f4{pre[nb][4 * q], pre[nb][4 * q + 1], pre[nb][4 * q + 2], pre[nb][4 * q + 3]}; hp = gru_parked_h(a, pre[nb], q + 2); }
                        else { z = *(const f4*)(a.aux0 + o); hp = ld4(a.state, opx * (unsigned)C, n4, a.state_packed); }
                        f4 hn;
#pragma unroll
                        for (int j = 0; j < 4; ++j)   // submodules.py:285: prev*(1-update) + out*update
                            hn[j] = __fadd_rn(__fmul_rn(hp[j], 1.0f - z[j]), __fmul_rn(tanh_t<FAST>(v[j]), z[j]));
                        st4(a.state, opx * (unsigned)C, n4, hn, a.state_packed);
                    }
                }
            }
            return;
        }
        if (epi == EPI_BIAS_TANH) {      // HyperE2VID bases_net (hyper_dynamic.py:41-48): no operands, no fusion
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                unsigned opx; int cgb, oy, ox;
                out_addr<GROUPED>(a, ec, n0, h, nb, opx, cgb, oy, ox);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c4 = cgb + 8 * q;
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = tanh_t<FAST>(acc[nb][4 * q + j]);
                    if (mvalid && c4 < nvalid) st4(a.out, opx * ct, c4, v, a.out_packed);
                }
            }
            return;
        }
        // bias [+ residual] [+ ReLU] [+ fused skip] [+ fused prediction layer]
        const bool res = (epi == EPI_RESIDUAL_RELU);
        const bool relu = (epi != EPI_BIAS);
        const bool has_pre = res || a.post_add != nullptr;
        const int pre_pk = res ? a.res_packed : a.padd_packed;
        const float* pw = a.pred_w;
        float pred_part = 0.f;
        // prediction weights of the lane's channels, fetched ONCE up front when a column group is one 32-column block
        // (E2VID's last decoder): a load inside the (nb, q) body is consumed at once, i.e. 16 serialised L2 round trips
        const bool pw_shared = pw && grp_cols == 32;
        f4 pwq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pwq[q] = pw_shared ? *(const f4*)(pw + 4 * h + 8 * q) : f4{0.f, 0.f, 0.f, 0.f};
        const bool group_store = a.out_packed && !pw && a.group_store;      // PACKED output: whole 64-B groups after a lane exchange
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            unsigned opx; int cgb, oy, ox;
            out_addr<GROUPED>(a, ec, n0, h, nb, opx, cgb, oy, ox);
            f32x16 outv;
#pragma unroll
            for (int i = 0; i < 16; ++i) outv[i] = 0.f;
            const bool pre_group = has_pre && pre_pk && a.group_store;
            f32x16 pvall = pre[nb];
            if (pre_group) {      // whole-group operand (epi_prefetch): decode, back to the accumulator order (all lanes take part)
                if constexpr (FMT >= 2) {
                    unsigned g[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) g[i] = __float_as_uint(pre[nb][i]);
                    if constexpr (FMT == 3) unpack16_xchg_p6(g, pvall); else unpack16_xchg_h2(g, pvall);
                } else {
                    unsigned g[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i) g[i] = __float_as_uint(pre[nb][i]);
                    unpack16_xchg(g, pvall);
                }
            }
            // P6: a fused skip beside a residual (the last residual block) is loaded and decoded as the lane's whole group too -- the
            // 4-channel P6 reader decodes its codes by hand
            [[maybe_unused]] f32x16 padd_all;
            // (NB == 4: the residual blocks' kernels; the narrower tiles of conv_bandk_kernel have no registers to spare for it)
            [[maybe_unused]] const bool padd_group = FMT == 3 && NB == 4 && res && a.post_add && a.padd_packed;
            if constexpr (FMT == 3 && NB == 4) {
                if (padd_group) {      // (block-uniform; every lane takes part in the exchange)
                    unsigned g[16];
                    const int cg = cgb - 4 * h + 16 * h;
                    const bool ld = mvalid && cg < nvalid;
                    const u32x4_t* gp = (const u32x4_t*)(a.post_add + (ld ? opx * ct + (unsigned)cg : 0u));
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const u32x4_t t = gp[i]; g[4 * i] = t[0]; g[4 * i + 1] = t[1]; g[4 * i + 2] = t[2]; g[4 * i + 3] = t[3]; }
                    unpack16_xchg_p6(g, padd_all);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c4 = cgb + 8 * q;                    // channel inside the column group
                f4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[nb][4 * q + j];
                if (mvalid && c4 < nvalid) {
                    const unsigned orow = opx * ct;
                    // the prefetched operand (raw bits from epi_setup), decoded if PACKED
                    f4 pv = {pvall[4 * q], pvall[4 * q + 1], pvall[4 * q + 2], pvall[4 * q + 3]};
                    if (has_pre && pre_pk && !pre_group && FMT != 3) {
                        const uint2 phi = {__float_as_uint(pv[0]), __float_as_uint(pv[1])};
                        if constexpr (FMT == 2) { const uint2 plo = {__float_as_uint(pv[2]), __float_as_uint(pv[3])}; pv = unpack4_h2(phi, plo); }
                        else pv = unpack4(phi, __float_as_uint(pv[2]));
                    }
                    if (res) v += pv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = relu ? fmaxf(v[j], 0.f) : v[j];
                    if (pw && a.out) st4(a.out, orow, c4, v, a.out_packed);        // debug copy of the layer's own output
                    // skip_sum fused into the producer (model_util.py:4-5); prefetched unless the residual took the slot
                    if (a.post_add) {
                        if (res) {
                            if (padd_group) v += f4{padd_all[4 * q], padd_all[4 * q + 1], padd_all[4 * q + 2], padd_all[4 * q + 3]};
                            else { const f4 s4 = ld4(a.post_add, orow, c4, a.padd_packed); v += s4; }
                        }
                        else v += pv;
                    }
                    if (group_store) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) outv[4 * q + j] = v[j];
                    } else if (!pw) { if (a.out_packed) sat_check4<FMT>(a.sat, v); st4(a.out, orow, c4, v, a.out_packed); }
                    else {
                        const f4 w4 = pw_shared ? pwq[q] : *(const f4*)(pw + c4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) pred_part = fmaf(v[j], w4[j], pred_part);
                    }
                }
            }
            if (group_store) {      // (every lane takes part in the exchange; the two halves of a pixel share mvalid)
                float w16[16];
                xchg16(outv, w16);
                const int cg = cgb - 4 * h + 16 * h;           // the lane's group inside the column group
                if (mvalid && cg < nvalid && (!(EPI_ABLATE & 64) || w16[0] == 123.456f)) { sat_check16<FMT>(a.sat, w16); store16_fmt<FMT>(a.out, opx * ct, cg, w16); }
                __builtin_amdgcn_sched_barrier(0);             // one block's conversion temporaries at a time
            }
            // fused 1x1 prediction conv (model/unet.py:136-138): the group's last 32-column block closes one output
            // pixel; the channels of a pixel live in the two lanes r and r+32
            if (pw && ((n0 + nb * 32 + 32) % grp_cols) == 0) {
                pred_part += __shfl_xor(pred_part, 32, 64);
                if (h == 0 && mvalid) {
                    const int y = oy - a.crop_y0, x = ox - a.crop_x0;
                    float sres = pred_part + a.pred_b;
                    if (a.pred_skip_dot) sres += a.pred_skip_dot[opx];
                    if (a.pred_sigmoid) sres = sigmoid_t<false>(sres);
                    if (a.prev_rec) a.prev_rec[opx] = sres;
                    if ((unsigned)y < (unsigned)a.crop_h && (unsigned)x < (unsigned)a.crop_w)
                        img_out[(unsigned)((ec.e_img * a.crop_h + y) * a.crop_w + x)] = sres;
                }
                pred_part = 0.f;
            }
        }
    }
}
#endif

// LSTM = true is the ConvLSTM gate convolution (its own kernel symbol: 65 % of E2VID's FLOPs, the kernel
// bench.py's roofline block and profiles/ quote); LSTM = false carries every other epilogue.
// X3: 0 = fp32 MFMA; 2 = split arithmetic on PACKED activations (1, splitting PLAIN activations in registers, is gone:
// VALU kernels that feed a matrix-core convolution write PACKED themselves)
template <int KC, int WM, int NB, bool LSTM, bool GROUPED, bool REGSTAGE = false, int X3 = 0>
__global__ __launch_bounds__(64 * WM, (X3 == 0) ? ((WM == 4 && !LSTM) ? 2 : 1) : (WM != 4 ? 1 : (NB >= 2 ? 2 : 3))) void conv_igemm_kernel(const ConvArgs* __restrict__ ap, float* __restrict__ img_out) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs& a = *ap;   // plan resident in device memory: wave-uniform -> scalar loads
    constexpr int SP = KC / 4;                 // 16-B slots per row
    constexpr int NT = 64 * WM;
    constexpr int A_F4 = 32 * WM * SP, B_F4 = 32 * NB * SP;
    constexpr int A_PER = A_F4 / NT;           // = SP/2 wave-wide 1-KiB DMA pieces per wave
    constexpr int B_PER = (B_F4 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float4 lds[2][A_F4 + B_F4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wmi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = a.hm * a.wm;
    const int M = a.n * hw;
    const int ntiles = a.cout / (32 * NB);

    // XCD-aware bijective remap of the 1-D grid (block b runs on XCD b % 8)
    int lin;
    {
        const int total = gridDim.x, bid = blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ntile = lin % ntiles;
    const int mtile = lin / ntiles;
    const ConvTaps& tp = a.tp;
    const int m0 = mtile * 32 * WM, n0 = ntile * 32 * NB;

    // column groups (sub-pixel phases of a transposed conv) covered by this N tile -> taps this block needs
    int tile_groups = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) tile_groups |= 1 << col_group(tp, n0 + nb * 32);
    unsigned act_taps = (1u << tp.ntaps) - 1u;     // ntaps <= 25
    if constexpr (GROUPED) {
        act_taps = 0;
        for (int t = 0; t < tp.ntaps; ++t) act_taps |= (tp.tap_groups[t] & tile_groups) ? (1u << t) : 0u;
    }
    const int n_act = __builtin_popcount(act_taps);

    const int c0 = a.c0, c1 = a.c1;
    const int cin_total = c0 + (a.in_mode == IN_CAT ? c1 : 0);
    const int nchunks = cin_total / KC;
    const int nsteps = n_act * nchunks;          // K steps this block runs (active taps only)
    const int ktot = tp.ntaps * nchunks * KC;    // row length of the weight matrix
    const int hin = a.hin, win = a.win;

    // buffer descriptors: out-of-range offsets read as 0.0f -> zero padding and ragged M for free
    const unsigned in_pix = (unsigned)a.n * hin * win;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in0, in_pix * (unsigned)c0 * 4u);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.in1 ? a.in1 : a.in0, in_pix * (unsigned)(a.in1 ? c1 : c0) * 4u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wgt, (unsigned)a.cout * ktot * 4u);

    // LDS image is lane-linear per DMA piece (64 lanes x 16 B): float4 index idx = piece*64 + lane holds
    // row idx/SP, slot idx%SP; the XOR swizzle is applied to the SOURCE quad (and again on the fragment read)
    int r_iy[A_PER], r_ix[A_PER], r_pix[A_PER];
    unsigned a_q4[A_PER];
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int idx = tid + j * NT;
        const int row = idx / SP;
        a_q4[j] = (unsigned)(((idx % SP) ^ swz<KC>(row)) * 4);
        const int m = m0 + row;
        if (m < M) {
            const int img = fdiv(m, a.div_hw_mul, a.div_hw_sh), rem = m - img * hw;
            const int my = fdiv(rem, a.div_w_mul, a.div_w_sh), mx = rem - my * a.wm;
            r_pix[j] = img * hin; r_iy[j] = my * a.stride; r_ix[j] = mx * a.stride;
        } else {
            r_pix[j] = 0; r_iy[j] = -(1 << 28); r_ix[j] = 0;
        }
    }
    unsigned b_off[B_PER];   // float offset of this lane's weight quad at step 0
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
        const int idx = tid + j * NT;
        const int row = idx / SP;
        b_off[j] = (unsigned)((n0 + row) * ktot + ((idx % SP) ^ swz<KC>(row)) * 4);
    }

    // (tap, chunk) of the step being issued / computed: walked incrementally over the set bits of act_taps
    unsigned bits_i = act_taps, bits_c = act_taps;
    int cc_i = 0, cc_c = 0;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[A_PER], rb[B_PER];     // REGSTAGE: tile s+1 waits in registers while tile s is multiplied
    auto issue = [&](int buf) {
        const int t = __builtin_ctz(bits_i), cc = cc_i;
        if (++cc_i == nchunks) { cc_i = 0; bits_i &= bits_i - 1; }
        int coff = cc * KC;
        const bool second = coff >= c0;
        const int csrc = second ? c1 : c0;
        if (second) coff -= c0;
        const int tpv = tp.tap[t];
        const int dy = (int)(short)(tpv & 0xffff), dx = tpv >> 16;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int iy = r_iy[j] + dy, ix = r_ix[j] + dx;
            unsigned voff = OOB_OFFSET;
            if ((unsigned)iy < (unsigned)hin && (unsigned)ix < (unsigned)win)
                voff = ((unsigned)((r_pix[j] + iy) * win + ix) * (unsigned)csrc + (unsigned)coff + a_q4[j]) * 4u;
            if constexpr (REGSTAGE) {
                ra[j] = second ? __builtin_amdgcn_raw_buffer_load_b128(rs1, voff, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, 0, 0);
            } else {
                lds_ptr_t dst = (lds_ptr_t)&lds[buf][(wmi + j * WM) * 64];
                if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, voff, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            if (B_F4 % NT == 0 || (wmi + j * WM) * 64 < B_F4) {   // wave-uniform
                const unsigned woff = (b_off[j] + (unsigned)((t * nchunks + cc) * KC)) * 4u;
                if constexpr (REGSTAGE) {
                    rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, woff, 0, 0);
                } else {
                    lds_ptr_t dst = (lds_ptr_t)&lds[buf][A_F4 + (wmi + j * WM) * 64];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, woff, 0, 0, 0);
                }
            }
        }
    };

    // REGSTAGE: the staged registers land in the same lane-linear LDS image the DMA would have produced
    auto store_staged = [&](int buf) {
        u32x4* l = (u32x4*)&lds[buf][0];
#pragma unroll
        for (int j = 0; j < A_PER; ++j) l[tid + j * NT] = ra[j];
#pragma unroll
        for (int j = 0; j < B_PER; ++j)
            if (B_F4 % NT == 0 || tid + j * NT < B_F4) l[A_F4 + tid + j * NT] = rb[j];
    };

    const int r = lane & 31, h = lane >> 5;
    const int sw = swz<KC>(r);   // rows wmi*32+r and nb*32+r swizzle like r

    // accumulators start at the bias; the epilogue's operands are requested now, a whole main loop early (epi_setup)
    f32x16 acc[NB];
    constexpr int PN = LSTM ? 1 : NB;
    f32x16 pre[PN];   // (ext-vector like acc: a plain 2-D float array was demoted to scratch by hipcc)
    EpiCtx ec;
    // (split mode: only the ConvLSTM cell state is requested a main loop early -- see conv3x3_band_kernel)
    // (fp32 mode, NB = 4: the 64 operand registers parked across the main loop pushed these kernels to 229 + 64 registers = one
    // wave per SIMD; loaded in the epilogue instead they fit two blocks per CU -- the fp32 mode's non-ConvLSTM layers)
    constexpr bool EARLY = LSTM || (X3 == 0 && NB < 4);
    epi_setup<NB, LSTM, GROUPED>(a, m0 + wmi * 32 + r, M, hw, n0, h, acc, pre, ec, true, EARLY);
    const int ablate = a.debug_ablate;   // timing ablation (EVR_ABLATE): results are garbage when non-zero
    const int mx_sa = a.mx_sa, mx_sb = a.mx_sb;
    issue(0);
    if constexpr (REGSTAGE) { store_staged(0); }
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        // LDS-DMA completion is NOT left to hipcc's automatic waitcnt insertion: with every fragment read bit-cast
        // to bf16 (PACKED mode) it no longer sees the reads alias the DMA writes and emits no vmcnt wait at all
        if constexpr (!REGSTAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ablate & 1)) __syncthreads();    // tile s is in LDS for every wave; everyone left tile s-1
        if (s + 1 < nsteps && !(ablate & 2)) issue(buf ^ 1);   // next tile: LDS-DMA, or plain loads into registers
        int groups_now = 0;
        if constexpr (GROUPED) {
            groups_now = tp.tap_groups[__builtin_ctz(bits_c)];
            if (++cc_c == nchunks) { cc_c = 0; bits_c &= bits_c - 1; }
        }
        const float4* la = &lds[buf][(wmi * 32 + r) * SP];
        const float4* lb = &lds[buf][A_F4 + r * SP];
        if constexpr (X3 != 0) {
            static_assert(X3 == 0 || (X3 == 2 && KC == 32), "split tiles are 32 k wide and take PACKED activations");
            // (every fragment is loaded AS float4, the LDS array's own type, and bit-cast: reads through a punned
            // pointer carry no alias with the LDS-DMA writes and hipcc then drops the vmcnt wait in front of them)
            const SplitFrag xa = ld_split(la, h, sw);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if constexpr (GROUPED) {
                    if (!((groups_now >> col_group(tp, n0 + nb * 32)) & 1)) continue;   // wave-uniform: zero weight block
                }
                const SplitFrag wb = ld_split(lb + nb * 32 * SP, h, sw);
                acc[nb] = mma_split(acc[nb], wb, xa, mx_sb, mx_sa);
            }
        } else {
#pragma unroll
        for (int i = 0; i < KC / 8; ++i) {
            const int q = (2 * i + h) ^ sw;
            const float4 av = la[q];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if constexpr (GROUPED) {
                    if (!((groups_now >> col_group(tp, n0 + nb * 32)) & 1)) continue;   // wave-uniform: zero weight block
                }
                const float4 bv = lb[nb * 32 * SP + q];
                // weights are the A operand (rows), activations the B operand (columns): acc = C^T (see epilogue)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.x, av.x, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.y, av.y, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.z, av.z, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.w, av.w, acc[nb], 0, 0, 0);
            }
        }
        }
        if constexpr (REGSTAGE) { if (s + 1 < nsteps) store_staged(buf ^ 1); }   // waits for the loads, ds_write_b128
    }

    if (ablate & 4) return;   // timing ablation: no epilogue at all
    if constexpr (!EARLY) epi_prefetch<NB, LSTM, GROUPED>(a, n0, h, pre, ec);
    epi_finish<NB, LSTM, GROUPED, (X3 != 0)>(a, ec, n0, h, acc, pre, img_out);
#endif   // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------------------
// 3x3 stride-1 'same' convolution (ConvLSTM gates, residual blocks) in split arithmetic on PACKED activations with the
// input rows RESIDENT in LDS -- the "band" kernel.
//
// The implicit-GEMM kernel above re-fetches the A tile for each of the 9 taps (L2 -> LDS traffic 32 flop/B: at the
// split MFMA rate the L2 cannot keep up).  Here a block owns TM = 256 consecutive pixels (flattened n*H*W) and
// 128 output columns, and walks K as  chunk (32 channels) x dy x dx:
//   A   for (chunk, dy) the TM+2 source pixels [m0 + dy*W - 1, m0 + dy*W + TM + 1) are ONE contiguous run of pixel
//       rows -> loaded once (double-buffered "band"), and the three dx taps read it at row offsets 0,1,2.  Pixels
//       that are not real neighbours (image borders, previous/next image of the batch) are zeroed in registers by a
//       per-lane 9-bit validity mask (8 v_cndmask per 16-k slab, hidden under the MFMAs).
//   B   the (tap, chunk) weight tile [128 x 32 k] streams through a 3-deep ring: it is requested two tap-steps ahead
//       and awaited with COUNTED vmcnt (loads complete in order), so a tile has ~2 steps of MFMA time to arrive.
//   L2 -> LDS bytes per (chunk, dy): 33 KB (A) + 48 KB (B) for 576 MFMAs = 2.4x less than the implicit GEMM.
// Everything else (fragment layout, C^T accumulators, epilogues) is shared with the kernel above.
// PHASES (transposed conv k5 s2 p2 only, checked on the host): how the tile's four 32-column blocks map to sub-pixel
// phases, so the (tap, block) pairs the transposed kernel does not connect -- phase py = 1 never takes dy = -1, px = 1
// never dx = -1 -- are dropped at COMPILE time (no branches in the step): 1 = block nb is phase (nb >> 1, nb & 1)
// [32 columns per phase], 2 = blocks {0,1} px = 0, {2,3} px = 1 [64 columns per phase; py is per tile], 0 = unknown.
template <int WM, int RING, bool LSTM, bool GROUPED, bool OVL = false, int PHASES = 0, bool KSPLIT = false>
__global__ __launch_bounds__(64 * WM, 2) void conv3x3_band_kernel(const ConvArgs* __restrict__ ap, float* __restrict__ img_out, int ksplit_arg, float* __restrict__ kws) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs& a = *ap;
    constexpr int NB = 4, SP = 8;
    constexpr int TM = 32 * WM;
    // OVL: tiles overlap by two pixels -- the band is exactly TM rows, the first and last lane of the tile only feed
    // their neighbours' dx taps and store nothing.  126 useful pixels of 128, but the band buffers shrink to 16 KiB
    // and 2 bands + a 3-slot ring are exactly 80 KiB: two blocks per CU WITH two-steps-ahead weight prefetch.
    constexpr int SHIFT = OVL ? 1 : 0;
    constexpr int TMV = OVL ? TM - 2 : TM;          // output pixels per tile
    constexpr int A_ROWS = OVL ? TM : TM + 8;       // TM + 2 source pixels needed; whole 8-row DMA pieces
    constexpr int A_PIECES = A_ROWS / 8;            // 1-KiB pieces per band
    constexpr int A_F4 = A_ROWS * SP, B_F4 = 32 * NB * SP;
    constexpr int NA_MAX = (A_PIECES + WM - 1) / WM, NA_MIN = A_PIECES / WM;   // band pieces per wave
    constexpr int NBW = (B_F4 / 64) / WM;           // weight-tile pieces per wave
    static_assert((B_F4 / 64) % WM == 0, "weight tile pieces must divide over the waves");
    static_assert(RING == 2 || RING == 3, "weight ring: 2 slots (request 1 step ahead) or 3 (2 steps ahead)");
    static_assert(!GROUPED || !LSTM, "a transposed conv has no ConvLSTM epilogue");
    __shared__ __attribute__((aligned(16))) float4 lds[2 * A_F4 + RING * B_F4 + SP];   // [band 0 | band 1 | ring slots | a zero row]
    constexpr int ZOFF = 2 * A_F4 + RING * B_F4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wmi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.win, H = a.hin;
    const int hw = H * W;
    const int M = a.n * hw;
    const int ntiles = a.cout / (32 * NB);
    int lin;
    {   // XCD-aware bijective remap of the 1-D grid (block b runs on XCD b % 8)
        const int total = gridDim.x, bid = blockIdx.x;
        const int q = total >> 3, rr = total & 7, xcd = bid & 7, idx = bid >> 3;
        lin = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    // split K (round 4, launches with fewer tiles than resident slots: small batches, the deep layers): ksplit consecutive blocks share
    // a tile and each takes a run of the input-channel chunks; partial accumulators go to `kws`, conv_ksplit_epilogue_kernel sums them
    // in a fixed order and runs the epilogue
    // (KSPLIT is its own instantiation: the one-tile-per-block form has no register to spare for the run bookkeeping)
    const int ksplit = KSPLIT ? ksplit_arg : 1;
    const int ks_i = KSPLIT ? lin % ksplit : 0;
    if (KSPLIT) lin /= ksplit;
    const int ntile = lin % ntiles, mtile = lin / ntiles;
    const int m0 = mtile * TMV, n0 = ntile * 32 * NB;      // band row 0 = source pixel m0 - 1 (+ dy*W)
    const int c0 = a.c0, c1 = a.c1;
    const int nchunks = (c0 + (a.in_mode == IN_CAT ? c1 : 0)) / 32;
    const int ktot = 9 * nchunks * 32;
    const int cbeg = KSPLIT ? (ks_i * nchunks) / ksplit : 0, cend = KSPLIT ? ((ks_i + 1) * nchunks) / ksplit : nchunks;
    const unsigned in_pix = (unsigned)M;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in0, in_pix * (unsigned)c0 * 4u);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.in1 ? a.in1 : a.in0, in_pix * (unsigned)(a.in1 ? c1 : c0) * 4u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wgt, (unsigned)a.cout * ktot * 4u);

    // per-lane constants of the DMA pieces this wave issues (piece j = wmi + jj*WM; lane -> row 8j + lane/8, slot lane%8)
    // per-lane byte offset of a piece's quad at shift 0 in either source ((pixel * channels + swizzled slot) * 4; may wrap
    // for the pixel before the tensor, which the range test below excludes): the step adds a wave-uniform term only --
    // no per-request multiply (v_mul_lo_u32 is quarter rate) in the loop
    int a_pix[NA_MAX]; unsigned a_v0[NA_MAX], a_v1[NA_MAX];
#pragma unroll
    for (int jj = 0; jj < NA_MAX; ++jj) {
        const int row = 8 * (wmi + jj * WM) + (lane >> 3);
        a_pix[jj] = m0 - 1 + row;
        const unsigned q = (unsigned)((((lane & 7) ^ swz<32>(row)) * 4));
        a_v0[jj] = ((unsigned)(a_pix[jj] * c0) + q) * 4u;
        a_v1[jj] = ((unsigned)(a_pix[jj] * c1) + q) * 4u;
    }
    unsigned b_off[NBW];
#pragma unroll
    for (int jj = 0; jj < NBW; ++jj) {
        const int row = 8 * (wmi + jj * WM) + (lane >> 3);
        b_off[jj] = (unsigned)((n0 + row) * ktot + (((lane & 7) ^ swz<32>(row)) * 4));
    }
    // band (chunk cc, dy = dyi - 1) -> LDS band buffer `buf`
    // (live = false: the same number of requests, all out of range -> no memory traffic, zeros land in the buffer;
    // keeps the counted vmcnt waits valid when a transposed-conv tile skips a band or a tap)
    auto issue_band = [&](int cc, int dyi, int buf, bool live) {
        int coff = cc * 32;
        const bool second = coff >= c0;
        const int csrc = second ? c1 : c0;
        if (second) coff -= c0;
        const int shift = (dyi - 1) * W;
        const unsigned uni = (unsigned)((shift * csrc + coff) * 4);      // wave-uniform part of the byte offset
#pragma unroll
        for (int jj = 0; jj < NA_MAX; ++jj) {
            if (jj < NA_MIN || wmi + jj * WM < A_PIECES) {      // wave-uniform
                const int pix = a_pix[jj] + shift;
                unsigned voff = OOB_OFFSET;
                if (live && (unsigned)pix < in_pix) voff = (second ? a_v1[jj] : a_v0[jj]) + uni;
                lds_ptr_t dst = (lds_ptr_t)&lds[buf * A_F4 + (wmi + jj * WM) * 64];
                if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, voff, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, 0, 0, 0);
            }
        }
    };
    // weight tile (tap t, chunk cc) -> ring slot
    auto issue_w = [&](int t, int cc, int slot, bool live) {
        const unsigned kofs = (unsigned)((t * nchunks + cc) * 32);
#pragma unroll
        for (int jj = 0; jj < NBW; ++jj) {
            // (PHASES = 1, four waves: piece jj of a wave lies in 32-column block jj; a block whose phase does not connect tap t is never
            // multiplied, so its quarter of the weight tile is not fetched -- the counted waits only rely on the band pieces coming last)
            if constexpr (PHASES == 1 && WM == 4 && NBW == 4) { if ((t / 3 == 0 && (jj >> 1) == 1) || (t % 3 == 0 && (jj & 1) == 1)) continue; }
            lds_ptr_t dst = (lds_ptr_t)&lds[2 * A_F4 + slot * B_F4 + (wmi + jj * WM) * 64];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, live ? (b_off[jj] + kofs) * 4u : OOB_OFFSET, 0, 0, 0);
        }
    };

    const int r = lane & 31, h = lane >> 5;
    const int sw = swz<32>(r);
    f32x16 acc[NB];
    constexpr int PN = LSTM ? 1 : NB;
    f32x16 pre[PN];
    EpiCtx ec;
    const int idx = wmi * 32 + r;                          // lane's row of the tile; its output pixel is m0 + idx - SHIFT
    const bool lane_ok = !OVL || (idx >= 1 && idx <= TM - 2);
    // (only the ConvLSTM cell state -- 16 registers -- is requested a main loop early; the 64 registers of a residual /
    // skip operand are loaded in the epilogue: the split fragments of a step already take 80)
    epi_setup<NB, LSTM, GROUPED>(a, m0 + idx - SHIFT, M, hw, n0, h, acc, pre, ec, lane_ok, LSTM);
    if (KSPLIT && ks_i > 0) {      // the bias belongs to the first run only
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    }

    // ConvTranspose2d(k5, s2) as a 3x3 conv whose N is phase-major (model.cpp prep_tconv): a (tap, phase) pair the
    // transposed kernel does not connect has a zero weight block.  tap_use[t] = phases of THIS N tile that use tap t:
    // unused blocks skip their MFMAs, unused taps their weight tile, unused dy their band (the steps still
    // synchronise, so the ring bookkeeping stays that of the dense case)
    int tile_groups = 0;
    if constexpr (GROUPED) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) tile_groups |= 1 << col_group(a.tp, n0 + nb * 32);
    }
    auto tap_use = [&](int t) -> int { return GROUPED ? (a.tp.tap_groups[t] & tile_groups) : 1; };
    auto band_use = [&](int dyi) -> bool { return !GROUPED || (tap_use(3 * dyi) | tap_use(3 * dyi + 1) | tap_use(3 * dyi + 2)) != 0; };

    // validity of the 9 neighbours of this lane's pixel (bit t = tap (t/3 - 1, t%3 - 1))
    unsigned vmask = 0;
    {
        const int m = m0 + idx - SHIFT;
        if (lane_ok && m < M) {
            const int img = fdiv(m, a.div_hw_mul, a.div_hw_sh), rem = m - img * hw;   // (W == a.wm: stride 1 on the input's grid)
            const int py = fdiv(rem, a.div_w_mul, a.div_w_sh), px = rem - py * W;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) vmask |= 1u << t;
            }
        }
    }

    // timing ablation (EVR_ABLATE, results are garbage when non-zero) only in -DEVR_BAND_ABLATE builds: the runtime
    // tests cost this kernel 8 %
#ifdef EVR_BAND_ABLATE
    const int ablate = a.debug_ablate;
#else
    constexpr int ablate = 0;
#endif
    const int mx_sa = a.mx_sa, mx_sb = a.mx_sb;
    // prologue: the zero row, band 0 and the first RING-1 weight tiles
    // (bare s_barrier below: __syncthreads() carries a fence that hipcc lowers to vmcnt(0), which would drain the ring)
    if (tid < SP) lds[ZOFF + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    issue_band(cbeg, 0, 0, band_use(0));
    issue_w(0, cbeg, 0, tap_use(0) != 0);
    if constexpr (RING == 3) {
        issue_w(1, cbeg, 1, tap_use(1) != 0);
        if constexpr (NBW == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    for (int c = cbeg; c < cend; ++c) {
        const int pa = (c - cbeg) & 1;        // parity of band index 3c + t/3 is (c + t/3) & 1 (c counted from the run's first chunk)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // ---- requests: weight tile of step s+2, and at the first tap of a band the NEXT band
            {
                constexpr int D = RING - 1;
                const int t2 = (t + D) % 9;
                int c2 = c + (t + D) / 9;
                if (c2 >= cend) c2 = cend - 1;                       // tail: harmless re-load into a free slot
                // ring slot of step s = 9c + t: s % 3, or s & 1 = (c + t) & 1
                const int slot2 = (RING == 3) ? (t + D) % 3 : (pa ^ ((t + D) & 1));
                if (!(ablate & 2)) issue_w(t2, c2, slot2, tap_use(t2) != 0);
            }
            if (t % 3 == 0 && !(ablate & 2)) {
                const int d2 = (t / 3 + 1) % 3;
                int c2 = c + (t / 3 + 1) / 3;
                if (c2 >= cend) c2 = cend - 1;
                issue_band(c2, d2, (pa ^ ((t / 3 + 1) & 1)), band_use(d2));
            }
            // ---- 12 MFMAs (8 f16 + 4 fp8) on band (c, t/3) rows r + t%3 and weight tile t%3 of the ring
            const int ab = pa ^ ((t / 3) & 1);
            int i = idx + (t % 3) - SHIFT;                  // band row of the lane's (dx) neighbour
            if constexpr (OVL) i = i < 0 ? 0 : (i > TM - 1 ? TM - 1 : i);   // (only the two non-storing edge lanes clamp)
            const int swi = swz<32>(i);
            // a pixel that is not a real neighbour (image border, neighbouring image of the batch) reads the zero row: one
            // select on the row offset instead of 16 on the fragment (these short-K layers are VALU-bound)
            const bool keep = (vmask >> t) & 1u;
            const float4* la = &lds[(t == 4 || keep || (ablate & 8)) ? ab * A_F4 + i * SP : ZOFF];
            const float4* lb = &lds[2 * A_F4 + ((RING == 3) ? (t % 3) : (pa ^ (t & 1))) * B_F4 + r * SP];
            const int use_t = tap_use(t);
            if (!GROUPED || use_t) {       // block-uniform: a tap no phase of this tile uses is skipped whole
                const SplitFrag xa = ld_split(la, h, swi);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    // (without PHASES a tap used by only SOME phases of the tile still runs all four blocks: the unused
                    // ones multiply zero weights; RUNTIME per-block branches cut the step into 3-MFMA fragments)
                    if constexpr (PHASES == 1) { if ((t / 3 == 0 && (nb >> 1) == 1) || (t % 3 == 0 && (nb & 1) == 1)) continue; }
                    if constexpr (PHASES == 2) { if (t % 3 == 0 && (nb >> 1) == 1) continue; }
                    const SplitFrag wb = ld_split(lb + nb * 32 * SP, h, sw);
                    acc[nb] = mma_split(acc[nb], wb, xa, mx_sb, mx_sa);
                }
            }
            // ---- the NEXT step's weight tile (and, before a band switch, the next band) must have landed; what was
            // requested after them may stay in flight (loads complete in order): NBW, plus >= NA_MIN band pieces
            // lgkmcnt(0): this wave's fragment reads of the step have left LDS before anyone overwrites the buffers
            if (ablate & 1) continue;
            if constexpr (RING == 3) {
                if (t % 3 == 2) {
                    if constexpr (NBW == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                } else {
                    static_assert(RING != 3 || NBW + NA_MIN == 6 || NBW + NA_MIN == 8, "update the counted waits");
                    if constexpr (NBW + NA_MIN == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
            } else {
                // 2-slot ring: the tile requested first in THIS step is needed next; only band pieces may stay in flight
                static_assert(NA_MIN == 4, "update the counted waits");
                if (t % 3 == 0) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
    }
    if (ablate & 4) return;
    if constexpr (KSPLIT) {      // partial accumulators, register order: [tile][run][wave][block * 4 + quad][lane] x 16 B
        float4* o = (float4*)kws + ((((size_t)lin * ksplit + ks_i) * WM + wmi) * (NB * 4)) * 64 + lane;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) o[(nb * 4 + q) * 64] = make_float4(acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]);
    } else {
        if constexpr (!LSTM) epi_prefetch<NB, LSTM, GROUPED>(a, n0, h, pre, ec);
        epi_finish<NB, LSTM, GROUPED, true>(a, ec, n0, h, acc, pre, img_out);
    }
#endif
}

// Second half of a split-K launch of conv3x3_band_kernel: one block per tile, same lane -> (pixel, channel) mapping; the ksplit partial
// accumulator sets are summed in run order (deterministic), then the shared epilogue runs as if the main loop had just ended.
template <bool LSTM, bool GROUPED>
__global__ __launch_bounds__(256, 2) void conv_ksplit_epilogue_kernel(const ConvArgs* __restrict__ ap, float* __restrict__ img_out, int ksplit, const float* __restrict__ kws) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs& a = *ap;
    constexpr int NB = 4, WM = 4, TM = 32 * WM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wmi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = a.hm * a.wm, M = a.n * hw;      // the GEMM's own pixel grid (= the input grid for the stride-1 band kernel, half of it for the k5 s2 form)
    const int ntiles = a.cout / (32 * NB);
    const int lin = blockIdx.x;
    const int ntile = lin % ntiles, mtile = lin / ntiles;
    const int m0 = mtile * TM, n0 = ntile * 32 * NB;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[NB], dummy[NB];
    constexpr int PN = LSTM ? 1 : NB;
    f32x16 pre[PN];
    EpiCtx ec;
    epi_setup<NB, LSTM, GROUPED>(a, m0 + wmi * 32 + r, M, hw, n0, h, dummy, pre, ec, true, LSTM);      // (the bias sits in run 0's partials)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    for (int s = 0; s < ksplit; ++s) {
        const float4* o = (const float4*)kws + ((((size_t)lin * ksplit + s) * WM + wmi) * (NB * 4)) * 64 + lane;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = o[(nb * 4 + q) * 64];
                acc[nb][4 * q] += v.x; acc[nb][4 * q + 1] += v.y; acc[nb][4 * q + 2] += v.z; acc[nb][4 * q + 3] += v.w;
            }
    }
    if constexpr (!LSTM) epi_prefetch<NB, LSTM, GROUPED>(a, n0, h, pre, ec);
    epi_finish<NB, LSTM, GROUPED, true>(a, ec, n0, h, acc, pre, img_out);
#endif
}

// The same, four times as wide (round 5): at one sequence the deep layers are 24-96 tiles, and the one-block-per-tile form above spent
// 15-19 us per launch -- as long as the split main loop it follows -- on four (ksplit) dependent rounds of 16 loads per lane with 24
// blocks on the chip.  Here a block is one (tile, 32-pixel row block): its four waves each take ONE 32-column block (plain epilogues)
// or ONE 4-channel quad of every gate (ConvLSTM), so a lane sums 4 x ksplit float4 partials that are all requested up front (run
// index clamped, not branched: one round trip), then runs the shared epilogue for its slice.  Same run order of the sums as above
// (0 + run 0 + run 1 + ...): bit-identical results.  9 launches of a one-sequence frame: 16.6 -> ~6 us each.
template <bool LSTM, bool GROUPED, int KSMAX>
__global__ __launch_bounds__(256, (KSMAX > 4) ? 1 : 2) void conv_ksplit_epi4_kernel(const ConvArgs* __restrict__ ap, float* __restrict__ img_out, int ksplit, const float* __restrict__ kws) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs& a = *ap;
    constexpr int WM = 4, TM = 32 * WM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = a.hm * a.wm, M = a.n * hw;
    const int ntiles = a.cout / 128;
    const int lin = blockIdx.x >> 2, wmi = blockIdx.x & 3;
    const int ntile = lin % ntiles, mtile = lin / ntiles;
    const int m0 = mtile * TM, n0 = ntile * 128;
    const int r = lane & 31, h = lane >> 5;
    // partial accumulators: [tile][run][wave][block * 4 + quad][lane] x 16 B (conv3x3_band_kernel / conv_band_prog_kernel)
    const size_t run_stride = (size_t)WM * 16 * 64;
    const float4* base = (const float4*)kws + (((size_t)lin * ksplit) * WM + wmi) * 16 * 64 + lane;
    float4 v[KSMAX][4];
    if constexpr (LSTM) {
        const int q = wv;
        const int m = m0 + wmi * 32 + r;
        const bool mvalid = m < M;
        const unsigned lstm_o = (unsigned)(mvalid ? m : 0) * (unsigned)a.hidden + (unsigned)((n0 >> 2) + 4 * h);
        const f4 cprev = *(const f4*)(a.state + lstm_o + 8 * q);
#pragma unroll
        for (int s = 0; s < KSMAX; ++s) {
            const size_t so = (size_t)(s < ksplit ? s : ksplit - 1) * run_stride;
#pragma unroll
            for (int g = 0; g < 4; ++g) v[s][g] = base[so + (g * 4 + q) * 64];
        }
        f4 gate[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KSMAX; ++s)
                if (s < ksplit) { t[0] += v[s][g].x; t[1] += v[s][g].y; t[2] += v[s][g].z; t[3] += v[s][g].w; }
            if constexpr (ARITH == 3 || ARITH == 4) t *= a.acc_scale;
            gate[g] = t;
        }
        if (!mvalid) return;
        f4 cn, hn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {      // submodules.py:227-245, as epi_finish<4, true>
            const float gi = sigmoid_t<true>(gate[0][j]);
            const float gf = sigmoid_t<true>(gate[1][j]);
            const float go = sigmoid_t<true>(gate[2][j]);
            const float gc = tanh_t<true>(gate[3][j]);
            cn[j] = __fadd_rn(__fmul_rn(gf, cprev[j]), __fmul_rn(gi, gc));
            hn[j] = go * tanh_t<true>(cn[j]);
        }
        *(f4*)(a.state + lstm_o + 8 * q) = cn;
        if (!a.out_packed) *(f4*)(a.out + lstm_o + 8 * q) = hn;
        else store4_fmt<FMT>(a.out, (unsigned)m * (unsigned)a.hidden, (n0 >> 2) + 4 * h + 8 * q, hn);
    } else {
        const int nb = wv, n1 = n0 + nb * 32;
        f32x16 acc[1], dummy[1], pre[1];
        EpiCtx ec;
        epi_setup<1, false, GROUPED>(a, m0 + wmi * 32 + r, M, hw, n1, h, dummy, pre, ec, true, false);      // (the bias sits in run 0's partials)
#pragma unroll
        for (int s = 0; s < KSMAX; ++s) {
            const size_t so = (size_t)(s < ksplit ? s : ksplit - 1) * run_stride;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[s][q] = base[so + (nb * 4 + q) * 64];
        }
        epi_prefetch<1, false, GROUPED>(a, n1, h, pre, ec);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KSMAX; ++s)
                if (s < ksplit) { t[0] += v[s][q].x; t[1] += v[s][q].y; t[2] += v[s][q].z; t[3] += v[s][q].w; }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][4 * q + j] = t[j];
        }
        epi_finish<1, false, GROUPED, true>(a, ec, n1, h, acc, pre, img_out);
    }
#endif
}

// the epilogue launch of a split-K convolution (ks runs per tile)
constexpr int KSPLIT_RUNS_MAX = 8;      // what conv_ksplit_epi4_kernel<..., 8> sums; ksplit_cap clamps EVR_KSPLIT / EVR_KSPLIT_SMALL to it
template <bool LSTM, bool GROUPED>
static void launch_ksplit_epilogue(const ConvArgs* d_args, float* img, int total, int ks, const float* kws, hipStream_t stream) {
    static const int wide = getenv("EVR_KSPLIT_EPI4") ? atoi(getenv("EVR_KSPLIT_EPI4")) : 1;      // (A/B: 0 = one block per tile)
    constexpr bool can4 = !(LSTM && ARITH == 4);      // (P6 tensors are written as whole 16-channel groups: the ConvLSTM quad form has no writer)
    if constexpr (can4) if (wide) {
        if (ks <= 4) { hipLaunchKernelGGL((conv_ksplit_epi4_kernel<LSTM, GROUPED, 4>), dim3(total * 4), dim3(256), 0, stream, d_args, img, ks, kws); return; }
        if (ks <= KSPLIT_RUNS_MAX) { hipLaunchKernelGGL((conv_ksplit_epi4_kernel<LSTM, GROUPED, 8>), dim3(total * 4), dim3(256), 0, stream, d_args, img, ks, kws); return; }
        // (more runs than the wide form sums -- ksplit_cap never lets that happen -- fall through to the any-count epilogue)
    }
    hipLaunchKernelGGL((conv_ksplit_epilogue_kernel<LSTM, GROUPED>), dim3(total), dim3(256), 0, stream, d_args, img, ks, kws);
}

// most runs per tile: 4, and 8 for launches of at most 64 tiles (the residual convolutions and dec0 of ONE sequence: 24 / 48 tiles --
// eight runs still leave slots free).  Measured on one box, three runs each, +-0.1 %: a flat cap of 8 is +1.1 % at one sequence (2183 vs
// 2159 frames/s) and -1.0 % at four (5 runs for the 92-tile residual convolutions and enc2.rec: the 8-run epilogue kernel holds one block
// per CU); caps of 5 / 6: -3 % at one sequence; 5-6 runs for 65-102 tiles with a six-run epilogue kernel at two blocks per CU: 2206 vs 2245 at
// one sequence, equal at four.  EVR_KSPLIT sets the general cap, EVR_KSPLIT_SMALL the one for <= 64 tiles.
// Both switches are clamped to KSPLIT_RUNS_MAX = 8: conv_ksplit_epi4_kernel<..., 8> sums at most eight partial sets (with
// EVR_KSPLIT=16 the main kernel wrote sixteen and the epilogue silently summed the first eight).
static int ksplit_cap(int total, int ks_max) {
    static const int ks_small = getenv("EVR_KSPLIT_SMALL") ? atoi(getenv("EVR_KSPLIT_SMALL")) : (getenv("EVR_KSPLIT") ? 0 : 8);
    const int cap = (total <= 64 && ks_small > ks_max) ? ks_small : ks_max;
    return cap > KSPLIT_RUNS_MAX ? KSPLIT_RUNS_MAX : cap;
}
// workspace of the split-K launches: ConvArgs::ksplit_ws, KSPLIT_WS_BYTES owned by the handle (model / LPIPS) whose plan this is --
// allocated with the plan, freed with it, never touched in the launch path (no hipMalloc / synchronisation here: a step can be
// captured into a hipGraph, and two devices or two handles never share partial sums)
static float* ksplit_workspace(const ConvArgs& a, size_t bytes) { return (a.ksplit_ws && bytes <= KSPLIT_WS_BYTES) ? a.ksplit_ws : nullptr; }

template <int WM, int RING, bool LSTM, bool GROUPED = false, bool OVL = false, int PHASES = 0>
static int launch_band(const ConvArgs& a, const ConvArgs* d_args, hipStream_t stream, float* img) {
    const int M = a.n * a.hm * a.wm;
    const int tmv = OVL ? 32 * WM - 2 : 32 * WM;
    const int mtiles = (M + tmv - 1) / tmv;
    const int total = mtiles * (a.cout / 128);
    // split K when the launch leaves most of the 512 resident slots empty (small batches; the deep layers: a 346x260 sequence is 12
    // 128-pixel tiles at the bottom of the UNet) and K is long enough to cut: runs of >= 1 input-channel chunk, at most 4, never more
    // blocks than slots.  (A cost model that also split launches of 200-1000 tiles into more than one round measured slower: 1487 vs
    // 1599 frames/s at one sequence, 4098 vs 4356 at eight -- the partial accumulators are 64 KB per block each way.)
    int ks = 1;
    if constexpr (WM == 4 && !OVL) {
        static const int ks_max = getenv("EVR_KSPLIT") ? atoi(getenv("EVR_KSPLIT")) : 4;      // (0 / 1: never)
        const int nchunks = (a.c0 + (a.in_mode == IN_CAT ? a.c1 : 0)) / 32;
        if (ks_max > 1 && total <= 192 && nchunks >= 2 && !a.pred_w) {
            ks = 512 / total;
            if (ks > nchunks) ks = nchunks;
            if (ks > ksplit_cap(total, ks_max)) ks = ksplit_cap(total, ks_max);
            if (ks < 2) ks = 1;
        }
    }
    float* kws = nullptr;
    if (ks > 1) {
        kws = ksplit_workspace(a, (size_t)total * ks * 4 * 16 * 64 * sizeof(float4));
        if (!kws) ks = 1;
    }
    // (round 5: the 3-slot weight ring -- tiles requested TWO steps ahead, 84 KB of LDS -- for the split launches, which have one
    // block per CU and nobody to hide a DMA latency behind: 2095 vs 2206 frames/s at one sequence, 3856 vs 3944 at four.  Not kept.)
    if (ks > 1) hipLaunchKernelGGL((conv3x3_band_kernel<WM, RING, LSTM, GROUPED, OVL, PHASES, true>), dim3(total * ks), dim3(64 * WM), 0, stream, d_args, img, ks, kws);
    else hipLaunchKernelGGL((conv3x3_band_kernel<WM, RING, LSTM, GROUPED, OVL, PHASES, false>), dim3(total), dim3(64 * WM), 0, stream, d_args, img, 1, (float*)nullptr);
    EVR_LAUNCH_CHECK();
    if (ks > 1) {
        launch_ksplit_epilogue<LSTM, GROUPED>(d_args, img, total, ks, (const float*)kws, stream);
        EVR_LAUNCH_CHECK();
    }
    return EVR_OK;
}

// the band kernel takes: split arithmetic on PACKED inputs, 3x3 taps in row-major order, stride 1 on the input's own grid,
// N a multiple of 128, and enough pixels to fill the chip with 256-pixel tiles
static bool band_eligible(const ConvArgs& a, int kc) {
    static const bool off = getenv("EVR_NO_BAND") != nullptr;
    if (off) return false;
    if (!(a.x3 && a.in_packed && kc == 32 && a.tp.ntaps == 9 && a.stride == 1)) return false;
    if (a.hm != a.hin || a.wm != a.win || a.cout % 128 != 0) return false;
    if (a.tp.ngroups == 1) { if (a.os != 1 || a.hout != a.hm || a.wout != a.wm) return false; }
    else if (a.os != 2 || a.hout != 2 * a.hm || a.wout != 2 * a.wm || a.epi == EPI_LSTM) return false;     // transposed conv
    for (int t = 0; t < 9; ++t)
        if (a.tp.tap[t] != (((t / 3 - 1) & 0xffff) | ((t % 3 - 1) * 65536))) return false;
    const int64_t M = (int64_t)a.n * a.hm * a.wm;
    // (round 3: no fill threshold any more -- for launches that do not fill the chip the band form still beats the implicit GEMM,
    // whose 9-fold A re-fetch is the cost either way: +16-17 % end to end at 1 / 4 / 8 / 16 sequences.  EVR_BAND_MIN restores one.)
    static const int min_blocks = getenv("EVR_BAND_MIN") ? atoi(getenv("EVR_BAND_MIN")) : 1;
    return ((M + 127) / 128) * (a.cout / 128) >= min_blocks;     // blocks of the default 128-pixel tiles
}

// ---------------------------------------------------------------------------------------------------
// Band kernel for the 5x5 (and narrow 3x3) stride-1 'same' convolutions (UpsampleConvLayer decoders of the E2VID+ / SSL-E2VID / HyperE2VID /
// ET-Net layouts, submodules.py:69-97; LPIPS conv2) in the split arithmetics on PACKED / H2 inputs.
//
// The implicit GEMM fetches an A tile [128 px x 32 ch] for EVERY one of the 25 taps: per K step 16 KB (A) + 4..16 KB (B) go
// L2 -> LDS for 128..512 matrix cycles per wave, and with the 2..4 blocks a CU holds that is 62 / 94 / 156 B/clk/CU at
// N = 128 / 64 / 32 columns against a path of 64 B/clk -- the k5 decoders were bound by LDS-DMA, not by the matrix cores.
// Here, as in conv3x3_band_kernel, the TM + 4 source pixels of one (chunk, dy) are ONE contiguous run of pixel rows (a band,
// double-buffered) that the FIVE dx taps read at row offsets 0..4: 3.5 KB of band per step instead of 16 KB of A tile
// (38 / 45 / 58 B/clk/CU).  Walk: chunk x dy x dx; weight tiles through a 2-slot ring, requested one step ahead; counted
// vmcnt + bare s_barrier; invalid neighbours (image borders, neighbouring images of the batch) read a zero row.
// NB = 32-column blocks per wave (N tile = 32 NB): 4 / 2 / 1 for 128 / 64 / 32 output channels.
template <int KW, int NB>
__global__ __launch_bounds__(256, (NB == 4) ? 2 : 3) void conv_bandk_kernel(const ConvArgs* __restrict__ ap, float* __restrict__ img_out) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs& a = *ap;
    constexpr int WM = 4, SP = 8, HALO = KW / 2;
    static_assert(KW == 3 || KW == 5, "3x3 or 5x5 taps");
    constexpr int TM = 32 * WM;
    constexpr int A_ROWS = TM + 8;                  // TM + 4 source pixels needed; whole 8-row (1-KiB) DMA pieces
    constexpr int A_PIECES = A_ROWS / 8;            // 17
    constexpr int A_F4 = A_ROWS * SP, B_F4 = 32 * NB * SP;
    constexpr int NA_MAX = (A_PIECES + WM - 1) / WM, NA_MIN = A_PIECES / WM;   // 5 / 4 band pieces per wave
    constexpr int B_PIECES = B_F4 / 64;             // 16 / 8 / 4 weight-tile pieces
    constexpr int NBW = (B_PIECES + WM - 1) / WM;   // per wave: 4 / 2 / 1
    static_assert(NA_MIN == 4 && B_PIECES % WM == 0, "tile bookkeeping");
    __shared__ __attribute__((aligned(16))) float4 lds[2 * A_F4 + 2 * B_F4 + SP];   // [band 0 | band 1 | weight slot 0 | 1 | a zero row]
    constexpr int ZOFF = 2 * A_F4 + 2 * B_F4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wmi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.win, H = a.hin;
    const int hw = H * W;
    const int M = a.n * hw;
    const int ntiles = a.cout / (32 * NB);
    int lin;
    {   // XCD-aware bijective remap of the 1-D grid (block b runs on XCD b % 8)
        const int total = gridDim.x, bid = blockIdx.x;
        const int q = total >> 3, rr = total & 7, xcd = bid & 7, idx = bid >> 3;
        lin = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int ntile = lin % ntiles, mtile = lin / ntiles;
    const int m0 = mtile * TM, n0 = ntile * 32 * NB;       // band row 0 = source pixel m0 - 2 (+ dy*W)
    const int c0 = a.c0, c1 = a.c1;
    const int nchunks = (c0 + (a.in_mode == IN_CAT ? c1 : 0)) / 32;
    const int ktot = KW * KW * nchunks * 32;
    const unsigned in_pix = (unsigned)M;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in0, in_pix * (unsigned)c0 * 4u);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.in1 ? a.in1 : a.in0, in_pix * (unsigned)(a.in1 ? c1 : c0) * 4u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wgt, (unsigned)a.cout * ktot * 4u);

    int a_pix[NA_MAX]; unsigned a_v0[NA_MAX], a_v1[NA_MAX];
#pragma unroll
    for (int jj = 0; jj < NA_MAX; ++jj) {
        const int row = 8 * (wmi + jj * WM) + (lane >> 3);
        a_pix[jj] = m0 - HALO + row;
        const unsigned q = (unsigned)((((lane & 7) ^ swz<32>(row)) * 4));
        a_v0[jj] = ((unsigned)(a_pix[jj] * c0) + q) * 4u;
        a_v1[jj] = ((unsigned)(a_pix[jj] * c1) + q) * 4u;
    }
    unsigned b_off[NBW];
#pragma unroll
    for (int jj = 0; jj < NBW; ++jj) {
        const int row = 8 * (wmi + jj * WM) + (lane >> 3);
        b_off[jj] = (unsigned)((n0 + row) * ktot + (((lane & 7) ^ swz<32>(row)) * 4));
    }
    auto issue_band = [&](int cc, int dyi, int buf) {
        int coff = cc * 32;
        const bool second = coff >= c0;                                   // cat(x, h): the chunk comes from the second source
        const int csrc = second ? c1 : c0;
        if (second) coff -= c0;
        const int shift = (dyi - HALO) * W;
        const unsigned uni = (unsigned)((shift * csrc + coff) * 4);      // wave-uniform part of the byte offset
#pragma unroll
        for (int jj = 0; jj < NA_MAX; ++jj) {
            if (jj < NA_MIN || wmi + jj * WM < A_PIECES) {      // wave-uniform
                const int pix = a_pix[jj] + shift;
                unsigned voff = OOB_OFFSET;
                if ((unsigned)pix < in_pix) voff = (second ? a_v1[jj] : a_v0[jj]) + uni;
                lds_ptr_t dst = (lds_ptr_t)&lds[buf * A_F4 + (wmi + jj * WM) * 64];
                if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, voff, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, 0, 0, 0);
            }
        }
    };
    auto issue_w = [&](int t, int cc, int slot) {
        const unsigned kofs = (unsigned)((t * nchunks + cc) * 32);
#pragma unroll
        for (int jj = 0; jj < NBW; ++jj) {
            lds_ptr_t dst = (lds_ptr_t)&lds[2 * A_F4 + slot * B_F4 + (wmi + jj * WM) * 64];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, (b_off[jj] + kofs) * 4u, 0, 0, 0);
        }
    };

    const int r = lane & 31, h = lane >> 5;
    const int sw = swz<32>(r);
    f32x16 acc[NB];
    f32x16 pre[NB];
    EpiCtx ec;
    const int idx = wmi * 32 + r;                          // lane's row of the tile; its output pixel is m0 + idx
    epi_setup<NB, false, false>(a, m0 + idx, M, hw, n0, h, acc, pre, ec, true, false);   // operands are loaded in the epilogue

    unsigned vmask = 0;      // validity of the KW x KW neighbours of this lane's pixel (bit t = tap (t/KW - HALO, t%KW - HALO))
    {
        const int m = m0 + idx;
        if (m < M) {
            const int img = fdiv(m, a.div_hw_mul, a.div_hw_sh), rem = m - img * hw;
            const int py = fdiv(rem, a.div_w_mul, a.div_w_sh), px = rem - py * W;
#pragma unroll
            for (int t = 0; t < KW * KW; ++t) {
                const int yy = py + t / KW - HALO, xx = px + t % KW - HALO;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) vmask |= 1u << t;
            }
        }
    }
    const int mx_sa = a.mx_sa, mx_sb = a.mx_sb;
    if (tid < SP) lds[ZOFF + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    issue_band(0, 0, 0);
    issue_w(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (int c = 0; c < nchunks; ++c) {
        const int pa = c & 1;        // bands per chunk and taps per chunk are both odd: band parity (c + dy) & 1, ring slot (c + t) & 1
#pragma unroll
        for (int t = 0; t < KW * KW; ++t) {
            {   // the weight tile of step s + 1 (first, so that the counted wait below can leave the band in flight)
                const int t2 = (t + 1) % (KW * KW);
                int c2 = c + (t + 1) / (KW * KW);
                if (c2 >= nchunks) c2 = nchunks - 1;                 // tail: harmless re-load into the free slot
                issue_w(t2, c2, pa ^ ((t + 1) & 1));
            }
            if (t % KW == 0) {      // at the first tap of a band: request the NEXT band
                const int d2 = (t / KW + 1) % KW;
                int c2 = c + (t / KW + 1) / KW;
                if (c2 >= nchunks) c2 = nchunks - 1;
                issue_band(c2, d2, pa ^ ((t / KW + 1) & 1));
            }
            const int ab = pa ^ ((t / KW) & 1);
            const int i = idx + (t % KW);                   // band row of the lane's (dx) neighbour
            const int swi = swz<32>(i);
            const bool keep = (vmask >> t) & 1u;
            const float4* la = &lds[keep ? ab * A_F4 + i * SP : ZOFF];
            const float4* lb = &lds[2 * A_F4 + (pa ^ (t & 1)) * B_F4 + r * SP];
            const SplitFrag xa = ld_split(la, h, swi);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const SplitFrag wb = ld_split(lb + nb * 32 * SP, h, sw);
                acc[nb] = mma_split(acc[nb], wb, xa, mx_sb, mx_sa);
            }
            // the tile requested first in THIS step is needed next; only the band pieces requested after it may stay in flight
            if (t % KW == 0) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    epi_prefetch<NB, false, false>(a, n0, h, pre, ec);
    epi_finish<NB, false, false, true>(a, ec, n0, h, acc, pre, img_out);
#endif
}

template <int KW, int NB>
static int launch_bandk(const ConvArgs& a, const ConvArgs* d_args, hipStream_t stream, float* img) {
    const int M = a.n * a.hm * a.wm;
    const int total = ((M + 127) / 128) * (a.cout / (32 * NB));
    // (a.band_lds_pad: extra dynamic LDS per block -- caps the blocks a CU holds, for launches that share the chip with another stream)
    hipLaunchKernelGGL((conv_bandk_kernel<KW, NB>), dim3(total), dim3(256), (size_t)a.band_lds_pad, stream, d_args, img);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

// split arithmetic on PACKED inputs, kw x kw taps in row-major order, stride 1 on the input's own grid; every epilogue but the
// ConvLSTM one (which needs 128-column tiles: conv3x3_band_kernel / conv3x3_wide_kernel).  kw = 5: the upsample-conv decoders;
// kw = 3: the 3x3 layers whose N is not a multiple of 128 (ConvGRU layouts, HyperE2VID's bases_net, FireNet padded to 32 channels)
static bool bandk_eligible(const ConvArgs& a, int kc, int kw) {
    static const bool off = getenv("EVR_NO_BAND") != nullptr || (getenv("EVR_BAND5") && atoi(getenv("EVR_BAND5")) == 0);
    if (off) return false;
    if (!(a.x3 && a.in_packed && kc == 32 && a.tp.ntaps == kw * kw && a.stride == 1 && a.tp.ngroups == 1) || a.no_band5) return false;
    if (a.hm != a.hin || a.wm != a.win || a.os != 1 || a.hout != a.hm || a.wout != a.wm || a.cout % 32 != 0) return false;
    if (a.epi == EPI_LSTM) return false;
    if (kw == 3 && a.cout % 128 == 0) return false;       // the 128-column kernels take those
    for (int t = 0; t < kw * kw; ++t)
        if (a.tp.tap[t] != (((t / kw - kw / 2) & 0xffff) | ((t % kw - kw / 2) * 65536))) return false;
    const int nb = (a.cout % 128 == 0) ? 4 : (a.cout % 64 == 0) ? 2 : 1;
    if (a.pred_w && a.cout != 32 * nb) return false;      // a fused prediction needs a single N tile
    const int64_t M = (int64_t)a.n * a.hm * a.wm;
    static const int min_blocks = getenv("EVR_BAND_MIN") ? atoi(getenv("EVR_BAND_MIN")) : (getenv("EVR_BAND5_MIN") ? atoi(getenv("EVR_BAND5_MIN")) : 1);
    return ((M + 127) / 128) * (a.cout / (32 * nb)) >= min_blocks;
}

#if EVR_ARITH == 3
// ---------------------------------------------------------------------------------------------------
// FireNet's 3x3 layers (16 channels; ConvGRU gates over cat(x, h), residual-block convs) in the three-f16-product arithmetic.
//
// v_mfma_f32_32x32x16_f16 contracts 16 k -- exactly one 16-channel H2 group (64 B: 16 hi | 16 lo halves) per tap, so these layers
// need no channel padding: a (tap, source) step is three MFMAs (lo_w hi_x + hi_w lo_x + hi_w hi_x).  The whole weight matrix of
// an N tile is tiny (32 rows x 9 taps x <= 2 sources x 64 B = 36 KB), so the kernel is WEIGHT-STATIONARY and persistent: a block
// loads it into LDS once, then walks M tiles of 128 pixels; per tile 3 x (1 | 2) bands of TM + 2 pixel rows (64 B each) stream
// through two buffers, one barrier per band; the next tile's first band is requested before the current tile's epilogue (bias /
// ReLU / residual / ConvGRU update / fused prediction: the shared epi_finish), which therefore runs under that DMA.
// The exact-fp32 MFMA these layers used before runs at 1/16 of the f16 rate; the padded-to-32-channels route (EVR_FIRENET_PAD32)
// doubles every tensor's bytes in a network that is HBM-bound (28 tensor passes of 64 B per pixel per frame).
template <int NB, int NBUF>
__global__ __launch_bounds__(256, 2) void conv3x3_c16_kernel(const ConvArgs* __restrict__ ap, float* __restrict__ img_out) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs& a = *ap;
    constexpr int WM = 4, SP = 4, TM = 32 * WM;         // SP: 16-B slots per 64-B row (hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15)
    // a band = TM + 2 pixel rows of 64 B: eight 16-row DMA pieces and a ninth of which only two rows (8 lanes) are requested -- a
    // whole ninth piece made a two-buffer block of the two-source layers 55.4 KB, 1.6 KB too much for a third block per CU
    constexpr int A_ROWS = TM + 2, A_PIECES = (A_ROWS + 15) / 16, A_F4 = A_ROWS * SP;
    constexpr int W_ROWS = 32 * NB, W_F4 = W_ROWS * SP, W_PIECES = W_ROWS / 16;      // one (tap, source) weight tile
    constexpr int MAX_TC = 18;                           // 9 taps x 2 sources
    // Round 4: a RING of NBUF band buffers, requests DIST bands ahead of the one being multiplied (across tile boundaries).  These
    // layers are bound by HBM latency, not bandwidth: with two buffers a block had one 9-KB band in flight, 2 blocks x 256 CUs x 9 KB
    // = 4.7 MB on the whole chip -- at ~2 us per round trip that is the 2.2 TB/s the kernel measured (0.27 of the roof).
    // NBUF = 4 (two-source layers: the ConvGRU gates; 74 KB, two blocks per CU) or 2 (one-source layers: the residual convolutions,
    // whose 9 weight tiles leave room for FOUR blocks per CU at 37 KB -- they are bound by per-tile latency, not by bytes: 4.2 us per
    // 128-pixel tile and block for 0.4 us of MFMAs); the LDS is sized per launch (launch_c16).
    constexpr int DIST = NBUF - 1;
    extern __shared__ __attribute__((aligned(16))) float4 lds[];   // [band ring | weights (9 or 18 tiles) | a zero row]
    (void)MAX_TC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wmi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.win, H = a.hin;
    const int hw = H * W;
    const int M = a.n * hw;
    const int c0 = a.c0, c1 = (a.in_mode == IN_CAT) ? a.c1 : 0;
    const int nsrc = c1 ? 2 : 1;                         // 16-channel chunks of K per tap
    const int ntc = 9 * nsrc;
    const int ktot = ntc * 16;
    const int ntiles_n = a.cout / (32 * NB);
    const int mtiles = (M + TM - 1) / TM;
    const unsigned in_pix = (unsigned)M;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in0, in_pix * (unsigned)c0 * 4u);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.in1 ? a.in1 : a.in0, in_pix * 16u * 4u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wgt, (unsigned)a.cout * ktot * 4u);
    const int r = lane & 31, hh = lane >> 5;
    const int idx = wmi * 32 + r;
    // (a layer with at most 16 real output columns -- the ConvGRU candidate convolution -- keeps only weight rows 0..15 in LDS: lanes
    // of rows 16..31 read the zero row, their accumulator columns are padding nobody stores; 18 KB less, a third block per CU)
    const int wrows = (NB == 1 && a.n_valid <= 16) ? 16 : W_ROWS, w_f4 = wrows * SP, w_pieces = wrows / 16;
    const int WOFF = NBUF * A_F4, ZOFF = NBUF * A_F4 + ntc * w_f4;

    // N tile of this block (persistent over M): blocks b, b + ntiles_n, ... share it -- with one N tile (every FireNet layer) all do
    const int ntile = blockIdx.x % ntiles_n, n0 = ntile * 32 * NB;
    const int mstart = blockIdx.x / ntiles_n, mstep = gridDim.x / ntiles_n;
    // ---- weights: all (tap, source) tiles of the N tile, once
    (void)W_F4; (void)W_PIECES;
    for (int p = wmi; p < ntc * w_pieces; p += WM) {
        const int tc = p / w_pieces, row = (p % w_pieces) * 16 + (lane >> 2);
        const unsigned off = (unsigned)((n0 + row) * ktot + tc * 16 + (((lane & 3) ^ swz<16>(row)) * 4));
        lds_ptr_t dst = (lds_ptr_t)&lds[WOFF + tc * w_f4 + (p % w_pieces) * 64];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, off * 4u, 0, 0, 0);
    }
    if (tid < SP) lds[ZOFF + tid] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto issue_band = [&](int m0, int bi, int buf) {      // band bi = (source bi / 3, dy = bi % 3 - 1) of the tile at m0
        const bool second = bi >= 3;
        const int shift = (bi % 3 - 1) * W;
        for (int p = wmi; p < A_PIECES; p += WM) {        // wave 0: pieces 0, 4, 8; the others two each (npw below)
            const int row = p * 16 + (lane >> 2);
            const int pix = m0 - 1 + row + shift;
            unsigned voff = OOB_OFFSET;
            if ((unsigned)pix < in_pix) voff = ((unsigned)pix * 16u + (unsigned)(((lane & 3) ^ swz<16>(row)) * 4)) * 4u;
            lds_ptr_t dst = (lds_ptr_t)&lds[buf * A_F4 + p * 64];
            if (p * 16 + 16 <= A_ROWS || lane < (A_ROWS - p * 16) * 4) {      // (the last piece: its two real rows only -- the rest would land in the next buffer)
                if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, voff, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, 0, 0, 0);
            }
        }
    };
    const int nbands = 3 * nsrc;
    // the request stream runs DIST bands ahead of the multiply stream; both walk (tile, band) in the same order
    int mt = mstart;
    int rq_mt = mstart, rq_bi = 0, rq_buf = 0;
    int ahead = 0;                                        // bands requested and not yet multiplied
    auto request_next = [&]() {
        if (rq_mt < mtiles) { issue_band(rq_mt * TM, rq_bi, rq_buf); ++ahead; }
        rq_buf = (rq_buf + 1) & (NBUF - 1);
        if (++rq_bi == nbands) { rq_bi = 0; rq_mt += mstep; }
    };
#pragma unroll
    for (int d = 0; d < DIST; ++d) request_next();
    int cur = 0;                                          // ring slot of the band being multiplied
    const int sw = swz<16>(r);
    // requests of one wave per band: wave 0 carries pieces 0, 4, 8, the others two -- wave-uniform wait counts
    const int npw = (wmi == 0) ? 3 : 2;
    // the accumulators start at the bias of the block's N tile: fetched ONCE (a per-tile fetch sits behind the requested bands in the
    // in-order return stream, and waiting for it would drain the look-ahead at every tile)
    f32x16 bias0[NB];
    {
        f32x16 pre0[NB]; EpiCtx ec0_;
        epi_setup<NB, false, false>(a, 0, M, hw, n0, hh, bias0, pre0, ec0_, true, false);
    }
    while (mt < mtiles) {
        const int m0 = mt * TM;
        f32x16 acc[NB];
        f32x16 pre[NB];
        EpiCtx ec;
        epi_setup<NB, false, false>(a, m0 + idx, M, hw, n0, hh, acc, pre, ec, true, false);      // (its bias loads are dead: replaced below)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = bias0[nb];
        unsigned vmask = 0;
        {
            const int m = m0 + idx;
            if (m < M) {
                const int img = fdiv(m, a.div_hw_mul, a.div_hw_sh), rem = m - img * hw;
                const int py = fdiv(rem, a.div_w_mul, a.div_w_sh), px = rem - py * W;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
                    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) vmask |= 1u << t;
                }
            }
        }
        for (int bi = 0; bi < nbands; ++bi) {
            // band `cur` has landed once at most DIST - 1 later bands of this wave are outstanding (requests complete in order; anything
            // else in flight -- the previous tile's stores -- only makes the wait conservative); the barrier publishes every wave's
            // pieces and tells that everyone has left the band multiplied before this one, whose slot the next request reuses
            // (the stream's tail, where fewer than DIST bands are ahead: wait for everything)
            if (DIST == 1 || ahead < DIST) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (npw == 3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            static_assert(DIST == 1 || DIST == 3, "update the counted waits: (DIST - 1) bands of 3 | 2 requests");
            --ahead;
            // the epilogue's operands are requested BEFORE the last band's look-ahead request, so that waiting for them later does
            // not wait for the bands requested behind them
            if (bi + 1 == nbands) epi_prefetch<NB, false, false>(a, n0, hh, pre, ec);
            request_next();
            const int buf = cur;
            cur = (cur + 1) & (NBUF - 1);
            const int src = bi / 3, dy = bi % 3;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int t = dy * 3 + dx;
                const int i = idx + dx;
                const bool keep = (vmask >> t) & 1u;
                const float4* la = &lds[keep ? buf * A_F4 + i * SP : ZOFF];
                const int swi = swz<16>(i);
                const u32x4_t xh = __builtin_bit_cast(u32x4_t, la[hh ^ swi]), xl = __builtin_bit_cast(u32x4_t, la[(2 + hh) ^ swi]);
                const float4* lb = &lds[(NB > 1 || r < wrows) ? WOFF + (t * nsrc + src) * w_f4 + r * SP : ZOFF];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const u32x4_t wh = __builtin_bit_cast(u32x4_t, lb[nb * 32 * SP + (hh ^ sw)]), wl = __builtin_bit_cast(u32x4_t, lb[nb * 32 * SP + ((2 + hh) ^ sw)]);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wl), __builtin_bit_cast(f16x8, xh), acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, xl), acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, xh), acc[nb], 0, 0, 0);
                }
            }
        }
        epi_finish<NB, false, false, true>(a, ec, n0, hh, acc, pre, img_out);
        mt += mstep;
    }
#endif
}

template <int NB>
static int launch_c16(const ConvArgs& a, const ConvArgs* d_args, hipStream_t stream, float* img) {
    const int M = a.n * a.hm * a.wm;
    const int mtiles = (M + 127) / 128, ntiles_n = a.cout / (32 * NB);
    const int nsrc = (a.in_mode == IN_CAT && a.c1) ? 2 : 1;
    static const int c16_blocks = getenv("EVR_C16_BLOCKS") ? atoi(getenv("EVR_C16_BLOCKS")) : 3;      // (A/B: blocks per CU of the one-source form)
    // SHIPPED DEFAULTS: two band buffers and three blocks per CU for every layer.  One-source layers (residual convolutions): measured
    // 3 > 4 > 2 blocks (157 / 183 / 189 us); two-source layers (ConvGRU z|r gates, 32 real columns; the candidate convolution, <= 16
    // real columns = half the weight rows): three blocks with two buffers beat the ring of four at two blocks (273 vs 312-320 us,
    // 265 vs 308-381).  EVR_C16_ZR / EVR_C16_OUT=<nbuf>,<blocks> select the ring of four (nbuf = 4, 70 KB) for A/B runs.
    const bool half_w = NB == 1 && a.n_valid <= 16;
    static const int out_nbuf = [] { const char* e = getenv("EVR_C16_OUT"); return e ? atoi(e) : 2; }();
    static const int out_blocks = [] { const char* e = getenv("EVR_C16_OUT"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 3; }();
    static const int zr_nbuf = [] { const char* e = getenv("EVR_C16_ZR"); return e ? atoi(e) : 2; }();      // (the z|r gate launches: <nbuf>,<blocks>)
    static const int zr_blocks = [] { const char* e = getenv("EVR_C16_ZR"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 3; }();
    const int nbuf = nsrc == 1 ? 2 : (half_w ? (out_nbuf == 4 ? 4 : 2) : (zr_nbuf == 4 ? 4 : 2));
    const int wrows = half_w ? 16 : 32 * NB;
    const size_t lds_bytes = ((size_t)nbuf * (128 + 2) * 4 + (size_t)9 * nsrc * wrows * 4 + 4) * sizeof(float4);
    int per_cu = nsrc == 1 ? (c16_blocks > 0 ? c16_blocks : 3) : (half_w ? (out_blocks > 0 ? out_blocks : 3) : (zr_blocks > 0 ? zr_blocks : 3));
    int per_n = 256 * per_cu;                             // persistent: the resident blocks walk the M tiles
    if (per_n > mtiles) per_n = mtiles;
    // the LDS-size attribute is per device (the ring-of-four forms need 70 KB): remember it per (device, NB)
    static std::atomic<unsigned> attr_done[64];
    int dev = 0;
    EVR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_relaxed)) {
        EVR_HIP(hipFuncSetAttribute((const void*)conv3x3_c16_kernel<NB, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EVR_HIP(hipFuncSetAttribute((const void*)conv3x3_c16_kernel<NB, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev >= 0 && dev < 64) attr_done[dev].store(1, std::memory_order_relaxed);
    }
    if (nbuf == 2) hipLaunchKernelGGL((conv3x3_c16_kernel<NB, 2>), dim3((unsigned)(per_n * ntiles_n)), dim3(256), lds_bytes, stream, d_args, img);
    else hipLaunchKernelGGL((conv3x3_c16_kernel<NB, 4>), dim3((unsigned)(per_n * ntiles_n)), dim3(256), lds_bytes, stream, d_args, img);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

// ---------------------------------------------------------------------------------------------------
// The same layers, walked ROW BY ROW (round 6; profiles/r06_firenet_rows.txt).  conv3x3_c16_kernel above takes 128 LINEAR pixels per
// tile and fetches three shifted 130-pixel bands for it, reading every input row three times through L2.  Here a block owns a
// 128-column STRIP of one image and walks down a segment of its rows, RPB rows per step: the bands of row y are the image rows y-1, y,
// y+1 of the strip, so a step needs only RPB NEW bands -- every input byte is fetched once (plus a 2-row halo per segment) -- and the
// horizontal / vertical borders are the DMA's own zero fill (no tap masks).  The bands live in a ring of NBUF = 2 RPB + 2 slots per
// source: RPB + 2 under the multiplies, RPB being filled for the next step.  A step is: barrier; request the next step's rows; the
// multiplies (fragments read one tap ahead); ONE wait -- for those rows, the step's epilogue operands and the previous step's stores,
// all requested at least a multiply phase earlier; the epilogue's stores; the NEXT step's operand requests (residual / ConvGRU state:
// gru_prefetch16).  No round trip is waited for on its own.  4 RPB waves: RPB image rows x four 32-pixel column blocks.  Per pixel the
// MFMA sequence -- sources, then rows, then columns, (lo_w hi_x, hi_w lo_x, hi_w hi_x) each -- is the one of the tile kernel:
// bit-identical results (tests/test_gpu_fullsize.py).
// Measured (64 x 240x192, single stream): residual convolutions 157 -> 154 us, ConvGRU z|r 265-273 -> 247 us, candidate + update
// 265-308 -> 293 us; FireNet step 25.9 -> 27.4 k frames/s.  All three forms move 3.2-3.8 TB/s of their algorithmic bytes, half of them
// writes: what is left is the memory system's mixed read/write rate, not the schedule (a deeper ring, more blocks, other step heights:
// equal or worse).
// NSRC / HALFW (sources; only weight rows 0..15 kept, as in the tile kernel) are template parameters so that every tap's LDS offset is
// an immediate of its ds_read: as kernel arguments they cost 36 address registers, hoisted out of the step loop.
// (Rows requested TWO steps ahead -- a ring of 3 RPB + 2 slots, the end-of-step wait leaving the newest requests in flight -- measured
// no better on the one-source layers, the only ones with LDS for it: 165 vs 164 us, 25.8 vs 26.3 k frames/s.  Removed.)
template <int NB, int RPB, int NSRC, bool HALFW>
__global__ __launch_bounds__(256 * RPB, (RPB == 1 ? 3 : 2)) void conv3x3_c16_rows_kernel(const ConvArgs* __restrict__ ap, float* __restrict__ img_out,
                                                                                       int rseg, int nseg, int nstrips) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs& a = *ap;
    constexpr int NC = 4 * RPB, SP = 4, TM = 128, NBUF = 2 * RPB + 2;
    constexpr int A_ROWS = TM + 2, A_PIECES = (A_ROWS + 15) / 16, A_F4 = A_ROWS * SP;
    extern __shared__ __attribute__((aligned(16))) float4 lds[];   // [source 0 ring | source 1 ring | weights (9 or 18 tiles) | a zero row | bias]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.win, H = a.hin;
    const int hw = H * W;
    const int M = a.n * hw;
    constexpr int nsrc = NSRC, ntc = 9 * nsrc, ktot = ntc * 16;
    const unsigned in_pix = (unsigned)M;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in0, in_pix * 16u * 4u);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.in1 ? a.in1 : a.in0, in_pix * 16u * 4u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wgt, (unsigned)a.cout * ktot * 4u);
    const int r = lane & 31, hh = lane >> 5;
    constexpr int wrows = HALFW ? 16 : 32 * NB, w_f4 = wrows * SP, w_pieces = wrows / 16;
    constexpr int WOFF = nsrc * NBUF * A_F4, ZOFF = WOFF + ntc * w_f4, BOFF = ZOFF + SP;
    const int ntiles_n = a.cout / (32 * NB);
    const int ntile = blockIdx.x % ntiles_n, n0 = ntile * 32 * NB;
    const int istart = blockIdx.x / ntiles_n, istep = gridDim.x / ntiles_n;
    const int items = a.n * nseg * nstrips;

    for (int p = wv; p < ntc * w_pieces; p += NC) {      // the N tile's weights, once
        const int tc = p / w_pieces, row = (p % w_pieces) * 16 + (lane >> 2);
        const unsigned off = (unsigned)((n0 + row) * ktot + tc * 16 + (((lane & 3) ^ swz<16>(row)) * 4));
        lds_ptr_t dst = (lds_ptr_t)&lds[WOFF + tc * w_f4 + (p % w_pieces) * 64];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, off * 4u, 0, 0, 0);
    }
    if (tid < SP) lds[ZOFF + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    // (the accumulators start at the bias, read from LDS: a global fetch per step would queue behind the step's band requests)
    if (tid < 8 * NB) lds[BOFF + tid] = *(const float4*)(a.bias + n0 + 4 * tid);

    // piece q of a request of `nrows` image rows: (row q / (9 nsrc), source, 16-pixel piece); the waves take q = wv, wv + NC, ...
    // (swz<16>(16 p + rl) does not depend on the piece p: a lane's byte offset inside a piece is fixed.)  Pixels outside the image --
    // and whole rows above / below it -- are the descriptor's zero fill.
    const int rl = lane >> 2;
    const int lane_off = rl * 64 + (((lane & 3) ^ swz<16>(rl)) * 16);
    auto issue_rows = [&](int img, int yy0, int nrows, int slot0, int x0, int ylast) {
        const int per_row = A_PIECES * nsrc;
        for (int q = wv; q < nrows * per_row; q += NC) {
            const int rowi = q / per_row, rem = q - rowi * per_row;
            const int sidx = rem / A_PIECES, p = rem - sidx * A_PIECES;
            const int yy = yy0 + rowi;
            if (yy > ylast) break;                             // (rows below the segment's halo are never multiplied)
            int slot = slot0 + rowi;
            if (slot >= NBUF) slot -= NBUF;
            const bool ok = (unsigned)yy < (unsigned)H && (unsigned)(x0 - 1 + p * 16 + rl) < (unsigned)W;
            // (unsigned arithmetic: pixel index x 64 B reaches 2^32 at the 2^26 pixels c16_rows_eligible admits, and x0 - 1 is -1 in
            // the first strip -- both wrap to the right offset for every lane that is `ok`)
            const unsigned voff = ok ? ((unsigned)((img * H + yy) * W + x0 - 1) * 64u + (unsigned)(p * 1024 + lane_off)) : OOB_OFFSET;
            lds_ptr_t dst = (lds_ptr_t)&lds[(sidx * NBUF + slot) * A_F4 + p * 64];
            if (p * 16 + 16 <= A_ROWS) {
                if (sidx) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, voff, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, 0, 0, 0);
            } else if (lane < (A_ROWS - (A_PIECES - 1) * 16) * 4) {      // the last piece: its two real rows only
                if (sidx) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, voff, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, 0, 0, 0);
            }
        }
    };

    const bool gru = NB == 1 && (a.epi == EPI_GRU_ZR || a.epi == EPI_GRU_OUT);      // (hidden == 16: c16_rows_eligible)
    // timing ablation (EVR_ABLATE + EVR_ABLATE_ALL; results are garbage): 2 no band requests behind an item's first, 4 no epilogue
    // (operands and stores), 8 no end-of-step wait
    const int ablate = a.debug_ablate;
    const int sw = swz<16>(r);
    const int rr = RPB > 1 ? wv / 4 : 0;                  // image row of the step, column block
    // a lane's weight row in tile 0.  HALFW: the tile holds rows 0..15 only; lanes 16..31 read them again -- their accumulator columns
    // (output channels 16..31) are padding nobody stores
    const float4* const wbase = &lds[WOFF + (r & (wrows - 1)) * SP];
    const int idx = (wv & 3) * 32 + r;

    for (int item = istart; item < items; item += istep) {
        const int strip = item % nstrips;
        const int t_ = item / nstrips;
        const int seg = t_ % nseg, img = t_ / nseg;
        const int x0 = strip * TM, y0 = seg * rseg;
        const int y1 = (y0 + rseg < H) ? y0 + rseg : H;
        // (everyone has left the previous item's last rows -- and, the first time, the weights and the bias are in place)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue_rows(img, y0 - 1, RPB + 2, 0, x0, y1);      // rows y0-1 .. y0+RPB: the first step's
        int slot_m1 = 0;                                  // ring slot of row ys - 1
        const int x = x0 + idx;
        f32x16 pre[NB];
        EpiCtx ec;
        // the epilogue's operands (residual / ConvGRU state) are requested ONE STEP ahead, behind the previous step's stores: neither
        // their round trip nor the stores' is ever waited for on its own -- a step has ONE wait, for everything older than its stores
        auto setup_step = [&](int ys) {
            const int y = ys + rr;
            const bool lane_ok = x < W && y < y1;
            const int m = (img * H + (y < H ? y : H - 1)) * W + (x < W ? x : W - 1);
            f32x16 dead[NB];
            epi_setup<NB, false, false>(a, m, M, hw, n0, hh, dead, pre, ec, lane_ok, false);      // (its bias loads are dead: the accumulators start from LDS)
            if (gru) gru_prefetch16(a, hh, pre[0], ec);
            else epi_prefetch<NB, false, false>(a, n0, hh, pre, ec);
        };
        setup_step(y0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int ys = y0; ys < y1; ys += RPB) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            {   // the next step's new rows ys+RPB+1 .. ys+2 RPB into the slots of the rows the previous step has left
                int slot = slot_m1 + RPB + 2;
                if (slot >= NBUF) slot -= NBUF;
                if (ys + RPB < y1 && !(ablate & 2)) issue_rows(img, ys + RPB + 1, RPB, slot, x0, y1);
            }
            f32x16 acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = lds[BOFF + nb * 8 + 2 * q + hh];
                    acc[nb][4 * q] = b4.x; acc[nb][4 * q + 1] = b4.y; acc[nb][4 * q + 2] = b4.z; acc[nb][4 * q + 3] = b4.w;
                }
            // the taps' fragments (x hi / lo of the pixel row, w hi / lo of the lane's output row) are read ONE TAP AHEAD of the MFMAs
            // that use them, into the other half of f: left to itself hipcc reuses three register quads and every MFMA waits a full
            // LDS latency for a read issued one MFMA earlier
            static_assert(NB == 1, "the tap pipeline below is written for one 32-column block");
            u32x4_t f[2][4];
            // (per lane: the x fragments' offsets inside a band for the three columns, the w fragments' inside a weight tile -- lanes of
            // rows the tile does not hold read the zero row through a negative tile offset)
            auto load_tap = [&](int src, int tp, u32x4_t (&d)[4]) {      // src, tp = dy * 3 + dx: compile-time after unrolling
                const int dy = tp / 3, dx = tp - dy * 3;
                int slot = slot_m1 + rr + dy;
                if (slot >= NBUF) slot -= NBUF;
                const float4* la = &lds[(src * NBUF + slot) * A_F4 + (idx + dx) * SP];
                const int swi = swz<16>(idx + dx);
                d[0] = __builtin_bit_cast(u32x4_t, la[hh ^ swi]); d[1] = __builtin_bit_cast(u32x4_t, la[(2 + hh) ^ swi]);
                const float4* lb = wbase + (tp * nsrc + src) * w_f4;
                d[2] = __builtin_bit_cast(u32x4_t, lb[hh ^ sw]); d[3] = __builtin_bit_cast(u32x4_t, lb[(2 + hh) ^ sw]);
            };
            load_tap(0, 0, f[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int src = 0; src < nsrc; ++src) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int cur = (src * 9 + tp) & 1;
                    if (tp + 1 < 9) load_tap(src, tp + 1, f[cur ^ 1]);
                    else if (src + 1 < nsrc) load_tap(src + 1, 0, f[cur ^ 1]);
                    __builtin_amdgcn_sched_barrier(0);      // (the next tap's reads go out BEFORE this tap's MFMAs: 96 matrix cycles cover the LDS latency)
                    const u32x4_t xh = f[cur][0], xl = f[cur][1], wh = f[cur][2], wl = f[cur][3];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wl), __builtin_bit_cast(f16x8, xh), acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, xl), acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, xh), acc[0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the next step's rows (requested a whole multiply phase ago), this step's operands and the previous step's stores
            if (!(ablate & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(ablate & 4)) {
                epi_finish<NB, false, false, true, NB == 1>(a, ec, n0, hh, acc, pre, img_out);
                if (ys + RPB < y1) setup_step(ys + RPB);
            } else if (acc[0][0] == 123.456f) img_out[0] = acc[0][1];
            slot_m1 += RPB;
            if (slot_m1 >= NBUF) slot_m1 -= NBUF;
        }
    }
#endif
}

// rows per segment: about two items per resident block, even, at least 8 (the halo rows are fetched once more per segment)
static void c16_rows_plan(const ConvArgs& a, int blocks, int* rseg, int* nseg, int* nstrips) {
    const int H = a.hin, W = a.win;
    *nstrips = (W + 127) / 128;
    int want = (2 * blocks + a.n * *nstrips - 1) / (a.n * *nstrips);
    if (want < 1) want = 1;
    int rs = (H + want - 1) / want;
    rs = (rs + 1) & ~1;
    if (rs < 8) rs = 8;
    *rseg = rs; *nseg = (H + rs - 1) / rs;
}

// EVR_C16_ROWS: 0 = the tile kernel everywhere, 2 = the row kernel whatever the size
static int c16_rows_mode() {
    static const int v = getenv("EVR_C16_ROWS") ? atoi(getenv("EVR_C16_ROWS")) : 1;
    return v;
}
static bool c16_rows_eligible(const ConvArgs& a) {
    if (c16_rows_mode() <= 0) return false;
    if ((int64_t)a.n * a.hin * a.win >= (1LL << 26)) return false;      // (32-bit byte offsets of 64-B pixels)
    if ((a.epi == EPI_GRU_ZR || a.epi == EPI_GRU_OUT) && a.hidden != 16) return false;      // (gru_prefetch16)
    // (a small batch has too few (strip, segment) items to fill the chip: the tile kernel's 128-pixel tiles stay; EVR_C16_ROWS=2
    // lifts the threshold -- the parity tests' small shapes)
    return c16_rows_mode() >= 2 || (int64_t)a.n * ((a.win + 127) / 128) * ((a.hin + 7) / 8) >= 1024;
}

template <int NB, int RPB, int NSRC, bool HALFW>
static int launch_c16_rows_t(const ConvArgs& a, const ConvArgs* d_args, hipStream_t stream, float* img, int per_cu) {
    constexpr int NBUF = 2 * RPB + 2;
    constexpr int wrows = HALFW ? 16 : 32 * NB;
    const size_t lds_bytes = ((size_t)NSRC * NBUF * (128 + 2) * 4 + (size_t)9 * NSRC * wrows * 4 + 4 + 8 * NB) * sizeof(float4);
    const int ntiles_n = a.cout / (32 * NB);
    int rseg, nseg, nstrips;
    c16_rows_plan(a, 256 * per_cu, &rseg, &nseg, &nstrips);
    int64_t items = (int64_t)a.n * nseg * nstrips;
    int blocks = 256 * per_cu;
    if (blocks > items) blocks = (int)items;
    static std::atomic<unsigned> attr_done[64];
    int dev = 0;
    EVR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_relaxed)) {
        EVR_HIP(hipFuncSetAttribute((const void*)conv3x3_c16_rows_kernel<NB, RPB, NSRC, HALFW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev >= 0 && dev < 64) attr_done[dev].store(1, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((conv3x3_c16_rows_kernel<NB, RPB, NSRC, HALFW>), dim3((unsigned)(blocks * ntiles_n)), dim3(256 * RPB), lds_bytes, stream,
                       d_args, img, rseg, nseg, nstrips);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

template <int NB>
static int launch_c16_rows(const ConvArgs& a, const ConvArgs* d_args, hipStream_t stream, float* img) {
    const int nsrc = (a.in_mode == IN_CAT && a.c1) ? 2 : 1;
    const bool half_w = NB == 1 && a.n_valid <= 16;
    // rows per step and blocks per CU: EVR_C16_ROWS_1=<rows>,<blocks> for the one-source layers, EVR_C16_ROWS_2 for the two-source ones
    // (A/B).  LDS: one source, 1 row = 4 x 8.3 + 18 (9) KB -> three blocks; two sources, 2 rows = 12 x 8.3 + 36 (18) KB -> one block of 8 waves
    static const int r1 = [] { const char* e = getenv("EVR_C16_ROWS_1"); return e ? atoi(e) : 1; }();
    static const int b1 = [] { const char* e = getenv("EVR_C16_ROWS_1"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 0; }();
    static const int r2 = [] { const char* e = getenv("EVR_C16_ROWS_2"); return e ? atoi(e) : 2; }();
    static const int b2 = [] { const char* e = getenv("EVR_C16_ROWS_2"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 0; }();
    const int rows = nsrc == 1 ? r1 : 2, bl = nsrc == 1 ? b1 : b2;      // (two sources: always two rows per step -- one row per step leaves four waves on the CU)
    (void)r2;
    const int per_cu = bl > 0 ? bl : (rows == 2 ? 1 : (nsrc == 1 ? 3 : 1));
    if (rows == 2) {
        if (nsrc == 1) return half_w ? launch_c16_rows_t<NB, 2, 1, true>(a, d_args, stream, img, per_cu) : launch_c16_rows_t<NB, 2, 1, false>(a, d_args, stream, img, per_cu);
        return half_w ? launch_c16_rows_t<NB, 2, 2, true>(a, d_args, stream, img, per_cu) : launch_c16_rows_t<NB, 2, 2, false>(a, d_args, stream, img, per_cu);
    }
    return half_w ? launch_c16_rows_t<NB, 1, 1, true>(a, d_args, stream, img, per_cu) : launch_c16_rows_t<NB, 1, 1, false>(a, d_args, stream, img, per_cu);
}

static bool c16_eligible(const ConvArgs& a, int kc) {
    if (!(a.x3 == 3 && a.in_packed && kc == 16 && a.tp.ntaps == 9 && a.stride == 1 && a.tp.ngroups == 1)) return false;
    if (a.c0 != 16 || !(a.in_mode == IN_SINGLE || a.c1 == 16)) return false;
    if (a.hm != a.hin || a.wm != a.win || a.os != 1 || a.hout != a.hm || a.wout != a.wm || a.cout != 32) return false;      // (one N tile of 32 columns: every FireNet layer)
    if (a.epi == EPI_LSTM) return false;
    for (int t = 0; t < 9; ++t)
        if (a.tp.tap[t] != (((t / 3 - 1) & 0xffff) | ((t % 3 - 1) * 65536))) return false;
    return true;
}
#endif   // EVR_ARITH == 3

// ---------------------------------------------------------------------------------------------------
// Band kernel, 256 pixels x 256 columns per block ("wide band").
//
// With the split arithmetic at 2/3 of the matrix cycles, conv3x3_band_kernel's 32-pixel x 128-column wave tile reads
// 20 KiB of fragments from LDS per 512 matrix cycles.  Here a wave owns TWO 32-pixel blocks (64 px x 128 columns, 8
// accumulators: a weight fragment read once feeds both -> 24 KiB per 1024 cycles), and the block is 4 (M) x 2 (N) waves
// over ONE pixel band: 256 pixels x 256 columns, 2 bands of 264 rows x 128 B + a 2-slot ring of 256 x 128 B weight
// tiles = 131 KiB, one block of 8 waves per CU (two waves per SIMD).  The band is fetched once for 256 columns instead
// of 128.  The walk (chunk x dy x dx, counted vmcnt, validity masks) is conv3x3_band_kernel's.  Epilogue operands
// (cell state / residual / fused skip) are loaded in the epilogue, one 32-pixel block at a time: with 128 accumulator
// registers there is no room to park them during the main loop.
// WN = 2: the 256 x 256 block described above.  WN = 1 ("twin"): 256 pixels x 128 columns, 4 waves, ONE band buffer
// (33.8 KB + 2 x 16 KB of weight tiles = 66 KB): TWO independent blocks per CU, so a SIMD's two waves belong to
// different blocks and do not stall at the same barrier; the price is the band switch every third step -- barrier,
// request the next band, wait for it, barrier -- whose DMA latency the other block's MFMAs have to cover.
// GROUPED / PHASES: the transposed decoders, as in conv3x3_band_kernel (phase-major N, (tap, phase) pairs the transposed
// kernel does not connect skipped: whole steps when no phase of the tile uses the tap, blocks at compile time via PHASES).
template <bool LSTM, int WN, bool GROUPED = false, int PHASES = 0>
__global__ __launch_bounds__(256 * WN, 2) void conv3x3_wide_kernel(const ConvArgs* __restrict__ ap, float* __restrict__ img_out) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs& a = *ap;
    constexpr int WM = 4, NW = WM * WN, MB = 2, NB = 4, SP = 8;
    constexpr bool SINGLE = (WN == 1);              // one band buffer
    constexpr int NBUF = SINGLE ? 1 : 2;
    constexpr int TM = 32 * MB * WM, TN = 32 * NB * WN;   // 256 x 256
    constexpr int A_ROWS = TM + 8;                  // TM + 2 source pixels needed; whole 8-row (1-KiB) DMA pieces
    constexpr int A_PIECES = A_ROWS / 8;            // 33
    constexpr int A_F4 = A_ROWS * SP, B_F4 = TN * SP;
    constexpr int NA_MAX = (A_PIECES + NW - 1) / NW, NA_MIN = A_PIECES / NW;   // 5 / 4 (9 / 8) band pieces per wave
    constexpr int NBW = (B_F4 / 64) / NW;           // 4 weight-tile pieces per wave
    static_assert(NBW == 4 && (SINGLE || NA_MIN == 4), "update the counted waits");
    __shared__ __attribute__((aligned(16))) float4 lds[NBUF * A_F4 + 2 * B_F4 + SP];   // [band 0 | band 1 | weight slot 0 | 1 | a zero row]
    constexpr int ZROW = (NBUF * A_F4 + 2 * B_F4) / SP;     // row index of the zero row, counted from lds[0]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wmi = wv & 3, wni = (WN == 1) ? 0 : (wv >> 2);
    const int W = a.win, H = a.hin;
    const int hw = H * W;
    const int M = a.n * hw;
    const int ntiles = a.cout / TN;
    int lin;
    {   // XCD-aware bijective remap of the 1-D grid (block b runs on XCD b % 8)
        const int total = gridDim.x, bid = blockIdx.x;
        const int q = total >> 3, rr = total & 7, xcd = bid & 7, idx = bid >> 3;
        lin = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int ntile = lin % ntiles, mtile = lin / ntiles;
    const int m0 = mtile * TM, n0 = ntile * TN;            // band row 0 = source pixel m0 - 1 (+ dy*W)
    const int n0w = n0 + wni * 32 * NB;                    // this wave's 128 columns
    const int c0 = a.c0, c1 = a.c1;
    const int nchunks = (c0 + (a.in_mode == IN_CAT ? c1 : 0)) / 32;
    const int ktot = 9 * nchunks * 32;
    const unsigned in_pix = (unsigned)M;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in0, in_pix * (unsigned)c0 * 4u);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.in1 ? a.in1 : a.in0, in_pix * (unsigned)(a.in1 ? c1 : c0) * 4u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wgt, (unsigned)a.cout * ktot * 4u);

    // DMA pieces (piece j = wv + jj*NW; lane -> row 8j + lane/8, 16-B slot lane%8, source slot swizzled).  Pieces of one
    // wave are 64 (32) rows apart, so their swizzle ((row >> 1) & 7) is the same and ONE pixel / offset register serves all
    // of them (an array per piece costs 11 more registers, which this kernel does not have)
    const int row0 = 8 * wv + (lane >> 3);
    const int a_pix0 = m0 - 1 + row0;
    const unsigned a_q0 = (unsigned)((((lane & 7) ^ swz<32>(row0)) * 4));
    const unsigned b_off0 = (unsigned)((n0 + row0) * ktot) + a_q0;
    // byte offset of piece 0's quad at shift 0 in either source (may wrap before the tensor; the range test excludes it):
    // a request adds wave-uniform terms only -- no per-request multiply in the loop
    const unsigned a_v0 = ((unsigned)(a_pix0 * c0) + a_q0) * 4u, a_v1 = ((unsigned)(a_pix0 * c1) + a_q0) * 4u;
    auto issue_band = [&](int cc, int dyi, int buf) {
        int coff = cc * 32;
        const bool second = coff >= c0;
        const int csrc = second ? c1 : c0;
        if (second) coff -= c0;
        const int shift = (dyi - 1) * W;
#pragma unroll
        for (int jj = 0; jj < NA_MAX; ++jj) {
            if (jj < NA_MIN || wv + jj * NW < A_PIECES) {      // wave-uniform
                const int pix = a_pix0 + 8 * NW * jj + shift;
                unsigned voff = OOB_OFFSET;
                if ((unsigned)pix < in_pix) voff = (second ? a_v1 : a_v0) + (unsigned)(((8 * NW * jj + shift) * csrc + coff) * 4);
                lds_ptr_t dst = (lds_ptr_t)&lds[buf * A_F4 + (wv + jj * NW) * 64];
                if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, voff, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, 0, 0, 0);
            }
        }
    };
    auto issue_w = [&](int t, int cc, int slot) {
        const unsigned kofs = (unsigned)((t * nchunks + cc) * 32);
#pragma unroll
        for (int jj = 0; jj < NBW; ++jj) {
            // (twin form, PHASES = 1: a wave's piece jj lies in 32-column block jj -- pieces are 32 rows apart -- and a block whose phase
            // does not connect tap t is never multiplied: its quarter of the weight tile is not fetched either)
            if constexpr (PHASES == 1 && WN == 1) { if ((t / 3 == 0 && (jj >> 1) == 1) || (t % 3 == 0 && (jj & 1) == 1)) continue; }
            lds_ptr_t dst = (lds_ptr_t)&lds[NBUF * A_F4 + slot * B_F4 + (wv + jj * NW) * 64];
            // (soffset carries the wave-uniform part: row block jj and the K offset of the step)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, b_off0 * 4u, (kofs + (unsigned)(8 * NW * jj * ktot)) * 4u, 0, 0);
        }
    };

    const int r = lane & 31, h = lane >> 5;
    const int sw = swz<32>(r);
    // (two named accumulator sets, not acc[MB][NB]: with the large plain epilogue hipcc leaves an array of arrays in scratch)
    f32x16 acc0[NB], acc1[NB];
    constexpr int PN = LSTM ? 1 : NB;
    f32x16 late[PN];             // the epilogue's operands, loaded there
    EpiCtx ec0, ec1;
    const int mA = m0 + wmi * 64 + r, mB = mA + 32;
    epi_setup<NB, LSTM, GROUPED>(a, mA, M, hw, n0w, h, acc0, late, ec0, true, false);   // prefetch = false: `late` untouched
    epi_setup<NB, LSTM, GROUPED>(a, mB, M, hw, n0w, h, acc1, late, ec1, true, false);
    int tile_groups = 0;         // phases of this wave's 128 columns, and per tap those that use it (conv3x3_band_kernel)
    if constexpr (GROUPED) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) tile_groups |= 1 << col_group(a.tp, n0w + nb * 32);
    }
    auto tap_use = [&](int t) -> int { return GROUPED ? (a.tp.tap_groups[t] & tile_groups) : 1; };
    auto neighbours = [&](int m) -> unsigned {   // validity of the pixel's 9 neighbours (bit t = tap (t/3 - 1, t%3 - 1))
        unsigned vm = 0;
        if (m < M) {
            const int img = fdiv(m, a.div_hw_mul, a.div_hw_sh), rem = m - img * hw;   // (W == a.wm: stride 1 on the input's grid)
            const int py = fdiv(rem, a.div_w_mul, a.div_w_sh), px = rem - py * W;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) vm |= 1u << t;
            }
        }
        return vm;
    };
    const unsigned vmask0 = neighbours(mA), vmask1 = neighbours(mB);
    const int mx_sa = a.mx_sa, mx_sb = a.mx_sb;
    // timing ablation (results are garbage when non-zero), COMPILE-TIME only (-DEVR_WIDE_ABLATE=mask, tools/ablate_wide.sh: a run-time
    // mask costs this kernel registers it does not have): bit 0 no waits / barriers in the loop, bit 1 no LDS-DMA requests in the
    // loop, bit 2 no epilogue, bit 5 no MFMAs (bits 4 / 6: no epilogue operand loads / no epilogue stores, EPI_ABLATE above)
#ifdef EVR_WIDE_ABLATE
    constexpr int ablate = EVR_WIDE_ABLATE;
#else
    constexpr int ablate = 0;
#endif

    // prologue: band 0 and the first weight tile (bare s_barrier: __syncthreads() carries a fence hipcc lowers to vmcnt(0))
    if (tid < SP) lds[NBUF * A_F4 + 2 * B_F4 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    issue_band(0, 0, 0);
    issue_w(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (int c = 0; c < nchunks; ++c) {
        const int pa = c & 1;        // parity of band index 3c + t/3 is (c + t/3) & 1; ring slot of step 9c + t is (c + t) & 1
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // requests of this step: the weight tile of step s + 1 (NBW pieces) and, at the first tap of a band, the NEXT
            // band (spreading them over the step's four MFMA groups measured no faster and costs registers)
            const int t2 = (t + 1) % 9;
            int cw = c + (t + 1) / 9;
            if (cw >= nchunks) cw = nchunks - 1;                     // tail: harmless re-load into the free slot
            const int d2 = (t / 3 + 1) % 3;
            int cb = c + (t / 3 + 1) / 3;
            if (cb >= nchunks) cb = nchunks - 1;
            if (!(ablate & 2)) issue_w(t2, cw, pa ^ ((t + 1) & 1));       // first, so that the counted wait below can leave the band in flight
            if (!SINGLE && t % 3 == 0 && !(ablate & 2)) issue_band(cb, d2, pa ^ ((t / 3 + 1) & 1));
            __builtin_amdgcn_sched_barrier(0);
            const int ab = SINGLE ? 0 : (pa ^ ((t / 3) & 1));
            const float4* lb = &lds[NBUF * A_F4 + (pa ^ (t & 1)) * B_F4 + (wni * 32 * NB + r) * SP];
            // band rows of the lane's dx neighbours, counted from lds[0]; a pixel that is not a real neighbour (image
            // border, neighbouring image of the batch) reads the zero row instead: one select per block, not 16 per fragment
            int l0 = wmi * 64 + r + (t % 3);                           // row inside the band (its swizzle is the DMA's)
            asm volatile("" : "+v"(l0));     // opaque: the slot addresses of the unrolled taps are recomputed, not kept live
            const int l1 = l0 + 32;
            int i0 = ab * A_ROWS + l0, i1 = ab * A_ROWS + l1;
            if (t != 4) { i0 = ((vmask0 >> t) & 1u) ? i0 : ZROW; i1 = ((vmask1 >> t) & 1u) ? i1 : ZROW; }
            const SplitFrag xa0 = ld_split(&lds[i0 * SP], h, swz<32>(l0));
            const SplitFrag xa1 = ld_split(&lds[i1 * SP], h, swz<32>(l1));
            // one block's weight fragments at a time (the scheduler would otherwise hoist all four: +48 registers, spills;
            // reading one block ahead -- left to the scheduler or pinned with scheduling barriers -- measured no faster)
            if (!GROUPED || tap_use(t)) {      // wave-uniform: a tap no phase of these columns uses is skipped whole
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if constexpr (PHASES == 1) { if ((t / 3 == 0 && (nb >> 1) == 1) || (t % 3 == 0 && (nb & 1) == 1)) continue; }
                if constexpr (PHASES == 2) { if (t % 3 == 0 && (nb >> 1) == 1) continue; }
                const SplitFrag wb = ld_split(lb + nb * 32 * SP, h, sw);
                if (ablate & 32) { acc0[nb][0] += __uint_as_float(wb.h0[0] ^ xa0.h0[0]); acc1[nb][0] += __uint_as_float(wb.h1[1] ^ xa1.f0[0]); }
                else {
                acc0[nb] = mma_split(acc0[nb], wb, xa0, mx_sb, mx_sa);
                acc1[nb] = mma_split(acc1[nb], wb, xa1, mx_sb, mx_sa);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            // the tile requested first in THIS step is needed next; only the band pieces requested after it may stay in
            // flight (loads complete in order).  lgkmcnt(0): this wave's fragment reads have left LDS before anyone
            // overwrites the buffers
            if (ablate & 1) continue;
            if constexpr (SINGLE) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (t % 3 == 2 && !(t == 8 && c == nchunks - 1)) {   // everyone has left the band: fetch the next one into the same buffer
                    if (!(ablate & 2)) issue_band(cb, d2, 0);
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                }
            } else {
                if (t % 3 == 0) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
    }
    if (ablate & 4) { if (acc0[0][0] + acc1[1][0] + acc0[2][3] + acc1[3][5] == 123.456f) img_out[tid] = acc0[1][1] + acc1[0][2] + acc0[3][0] + acc1[2][0]; return; }
    if constexpr (LSTM) {      // the cell update needs the four gate blocks together
        // (both pixel blocks' cell states are requested before the first block's gate arithmetic: the second request's latency
        // hides under it instead of following it)
        f32x16 late2[PN];
        epi_prefetch<NB, true, false>(a, n0w, h, late, ec0);
        epi_prefetch<NB, true, false>(a, n0w, h, late2, ec1);
        epi_finish<NB, true, false, true>(a, ec0, n0w, h, acc0, late, img_out);
        epi_finish<NB, true, false, true>(a, ec1, n0w, h, acc1, late2, img_out);
    } else {
        if constexpr (GROUPED) {
            // The last decoder with the prediction layer fused and every 32-column block = one sub-pixel phase (E2VID: dec2): a block
            // closes one output pixel per lane pair, so the epilogue is a reduction of the accumulators and nothing has to be parked.
            // All eight (pixel block, phase) skip terms are requested FIRST and consumed after the eight dot products: the general
            // path below ran eight load -> sigmoid -> store chains one after the other (timing ablation, round 4: 331 of this
            // layer's 657 us were its epilogue; 657 -> 428 us with this form).  Same operation order per value as epi_finish:
            // results are bit-identical.
            if (a.pred_w && a.tp.grp_cols == 32 && a.out == nullptr && a.post_add == nullptr && a.n_valid >= 32 &&
                (a.epi == EPI_BIAS || a.epi == EPI_BIAS_RELU)) {
                const bool relu = a.epi != EPI_BIAS;
                const float sc = (ARITH == 3 || ARITH == 4) ? a.acc_scale : 1.0f;
                unsigned opx[2][NB]; int oy[2][NB], ox[2][NB]; float skipv[2][NB];
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        int cgb;
                        const EpiCtx& ec = pb ? ec1 : ec0;
                        out_addr<true>(a, ec, n0w, h, nb, opx[pb][nb], cgb, oy[pb][nb], ox[pb][nb]);
                        skipv[pb][nb] = (a.pred_skip_dot && ec.mvalid) ? a.pred_skip_dot[opx[pb][nb]] : 0.f;
                    }
                f4 pwq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) pwq[q] = *(const f4*)(a.pred_w + 4 * h + 8 * q);
                float part[2][NB];
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        float pp = 0.f;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            float v = (pb ? acc1[nb][i] : acc0[nb][i]) * sc;
                            v = relu ? fmaxf(v, 0.f) : v;
                            pp = fmaf(v, pwq[i >> 2][i & 3], pp);
                        }
                        part[pb][nb] = pp;
                    }
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const EpiCtx& ec = pb ? ec1 : ec0;
                        float pp = part[pb][nb];
                        pp += __shfl_xor(pp, 32, 64);
                        if (h == 0 && ec.mvalid) {
                            const int y = oy[pb][nb] - a.crop_y0, x = ox[pb][nb] - a.crop_x0;
                            float sres = pp + a.pred_b;
                            if (a.pred_skip_dot) sres += skipv[pb][nb];
                            if (a.pred_sigmoid) sres = sigmoid_t<false>(sres);
                            if (a.prev_rec) a.prev_rec[opx[pb][nb]] = sres;
                            if ((unsigned)y < (unsigned)a.crop_h && (unsigned)x < (unsigned)a.crop_w)
                                img_out[(unsigned)((ec.e_img * a.crop_h + y) * a.crop_w + x)] = sres;
                        }
                    }
                return;
            }
        }
        // plain epilogues: one 32-column block at a time, and the second pixel block's 64 accumulators wait in LDS (idle
        // now: the last step ended with vmcnt(0) + barrier) -- 128 live accumulators plus the epilogue's operand /
        // conversion registers do not fit in 256, and hipcc's own spill goes to scratch memory
        // (round 4: the operands of a pixel block's FOUR column blocks are requested together -- two exposed load latencies per tile
        // instead of eight: the timing ablation put 100 of dec1's 590 us on these loads.  Not in the fp6 build: its group conversion
        // needs the registers, hipcc spilled 5 of them.)
        float4* park = &lds[wv * 1024];        // 16 KB per wave
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                park[(nb * 4 + q) * 64 + lane] = make_float4(acc1[nb][4 * q], acc1[nb][4 * q + 1], acc1[nb][4 * q + 2], acc1[nb][4 * q + 3]);
        if constexpr (FMT != 3) {
            {
                f32x16 op4[NB];
                epi_prefetch<NB, false, GROUPED>(a, n0w, h, op4, ec0);
                epi_finish<NB, false, GROUPED, true>(a, ec0, n0w, h, acc0, op4, img_out);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                f32x16 op4[NB];
                epi_prefetch<NB, false, GROUPED>(a, n0w, h, op4, ec1);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = park[(nb * 4 + q) * 64 + lane];
                        acc0[nb][4 * q] = v.x; acc0[nb][4 * q + 1] = v.y; acc0[nb][4 * q + 2] = v.z; acc0[nb][4 * q + 3] = v.w;
                    }
                epi_finish<NB, false, GROUPED, true>(a, ec1, n0w, h, acc0, op4, img_out);
            }
        } else {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x16 one[1], op[1];
                one[0] = acc0[nb];
                epi_prefetch<1, false, GROUPED>(a, n0w + 32 * nb, h, op, ec0);
                epi_finish<1, false, GROUPED, true>(a, ec0, n0w + 32 * nb, h, one, op, img_out);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x16 one[1], op[1];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = park[(nb * 4 + q) * 64 + lane];
                    one[0][4 * q] = v.x; one[0][4 * q + 1] = v.y; one[0][4 * q + 2] = v.z; one[0][4 * q + 3] = v.w;
                }
                epi_prefetch<1, false, GROUPED>(a, n0w + 32 * nb, h, op, ec1);
                epi_finish<1, false, GROUPED, true>(a, ec1, n0w + 32 * nb, h, one, op, img_out);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#endif
}

template <bool LSTM, int WN, bool GROUPED = false, int PHASES = 0>
static int launch_wide(const ConvArgs& a, const ConvArgs* d_args, hipStream_t stream, float* img) {
    const int M = a.n * a.hm * a.wm;
    const int total = ((M + 255) / 256) * (a.cout / (128 * WN));
    hipLaunchKernelGGL((conv3x3_wide_kernel<LSTM, WN, GROUPED, PHASES>), dim3(total), dim3(256 * WN), 0, stream, d_args, img);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

// ---------------------------------------------------------------------------------------------------
// Programmed band kernel: the k5 stride-2 encoder convolutions in split arithmetic on PACKED activations.
//
// Space-to-depth turns conv(k5, s2) into a 3x3 stride-1 convolution over 2x2 pixel blocks with 4*Cin channels
// (phase-major): in(2y + ky - 2, 2x + kx - 2) = block(y + dy, x + dx), phase (py, px) with 2*dy + py = ky - 2.  Of the
// 36 (tap, phase) pairs 25 exist; the others are not zero-padded but simply absent from the step PROGRAM the host
// builds (model.cpp): a list of (tap, weight-tile offset) steps grouped into bands = (phase, channel chunk, dy).
// The band of one (chunk, dy) is again ONE run of consecutive blocks -- here addressed as (row, x) with the
// row pitch of two input rows -- loaded once and read by its dx taps at row offsets 0/1/2, exactly as in
// conv3x3_band_kernel; the implicit GEMM re-fetches the A tile for all 25 taps (2.4x the L2 -> LDS bytes).
// Weight tiles: 2-slot ring, requested one step ahead; counted vmcnt + bare s_barrier.
template <int NB, bool KSPLIT = false>
__global__ __launch_bounds__(256, 2) void conv_band_prog_kernel(const ConvArgs* __restrict__ ap, float* __restrict__ img_out, int ksplit_arg, float* __restrict__ kws) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs& a = *ap;
    constexpr int WM = 4, SP = 8, TM = 32 * WM;
    constexpr int A_ROWS = TM + 8, A_PIECES = A_ROWS / 8, A_F4 = A_ROWS * SP, B_F4 = 32 * NB * SP;
    constexpr int NA_MAX = (A_PIECES + WM - 1) / WM, NA_MIN = A_PIECES / WM;
    constexpr int NBW = (B_F4 / 64) / WM;
    static_assert(NA_MIN == 4 && (B_F4 / 64) % WM == 0, "tile bookkeeping");
    __shared__ __attribute__((aligned(16))) float4 lds[2 * A_F4 + 2 * B_F4 + SP];   // [band 0 | band 1 | weight slot 0 | 1 | a zero row]
    constexpr int ZOFF = 2 * A_F4 + 2 * B_F4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wmi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Wb = a.wm, Hb = a.hm;                 // block grid = output grid
    const int hw = Hb * Wb;
    const int M = a.n * hw;
    const int nrows = a.n * Hb;
    const int ntiles = a.cout / (32 * NB);
    int lin;
    {   // XCD-aware bijective remap of the 1-D grid (block b runs on XCD b % 8)
        const int total = gridDim.x, bid = blockIdx.x;
        const int q = total >> 3, rr = total & 7, xcd = bid & 7, idx = bid >> 3;
        lin = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    // (split K as in conv3x3_band_kernel: ksplit consecutive blocks share a tile, each walks a run of whole BANDS of the step program)
    const int ksplit = KSPLIT ? ksplit_arg : 1;
    const int ks_i = KSPLIT ? lin % ksplit : 0;
    if (KSPLIT) lin /= ksplit;
    const int ntile = lin % ntiles, mtile = lin / ntiles;
    const int m0 = mtile * TM, n0 = ntile * 32 * NB;
    const int C = a.c0;
    const int nch2 = 4 * C / 32;                    // K chunks per tap in space-to-depth form
    const int ktot = 9 * nch2 * 32;
    const unsigned row_pitch = 2u * (unsigned)a.win * (unsigned)C, pix_pitch = 2u * (unsigned)C;   // floats
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in0, (unsigned)a.n * a.hin * a.win * (unsigned)C * 4u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wgt2, (unsigned)a.cout * ktot * 4u);
    // the whole step program in two VGPRs (lane l: entries l and 64 + l), read back with v_readlane
    const unsigned prog_lo = a.prog[lane], prog_hi = a.prog[64 + lane];
    auto entry = [&](int s) -> unsigned {
        return s < 64 ? (unsigned)__builtin_amdgcn_readlane((int)prog_lo, s) : (unsigned)__builtin_amdgcn_readlane((int)prog_hi, s - 64);
    };

    // per-lane constants of the DMA pieces this wave issues: block (row, x) of band row 8j + lane/8 for dy = 0
    int a_rho[NA_MAX]; unsigned a_xq[NA_MAX];
#pragma unroll
    for (int jj = 0; jj < NA_MAX; ++jj) {
        const int row = 8 * (wmi + jj * WM) + (lane >> 3);
        const int mb = m0 - 1 + row;
        int rho = 0x20000000, x = 0;
        if (mb >= 0 && mb < M) { rho = fdiv(mb, a.div_w_mul, a.div_w_sh); x = mb - rho * Wb; }
        a_rho[jj] = rho;
        // byte offset at dy = 0, phase 0, chunk 0 (garbage for the sentinel row, which the range test excludes): a request
        // adds a wave-uniform term only, no per-request multiply
        a_xq[jj] = ((unsigned)rho * row_pitch + (unsigned)x * pix_pitch + (unsigned)((((lane & 7) ^ swz<32>(row)) * 4))) * 4u;
    }
    unsigned b_off[NBW];
#pragma unroll
    for (int jj = 0; jj < NBW; ++jj) {
        const int row = 8 * (wmi + jj * WM) + (lane >> 3);
        b_off[jj] = (unsigned)((n0 + row) * ktot + (((lane & 7) ^ swz<32>(row)) * 4));
    }
    auto issue_band = [&](int src, int dy, int buf) {
        const int py = src & 1, px = (src >> 1) & 1, cc = src >> 2;
        const unsigned choff = (unsigned)((py * a.win + px) * C + cc * 32);
#pragma unroll
        for (int jj = 0; jj < NA_MAX; ++jj) {
            if (jj < NA_MIN || wmi + jj * WM < A_PIECES) {      // wave-uniform
                const int rho = a_rho[jj] + dy;
                unsigned voff = OOB_OFFSET;
                if ((unsigned)rho < (unsigned)nrows) voff = a_xq[jj] + (unsigned)(dy * (int)row_pitch + (int)choff) * 4u;
                lds_ptr_t dst = (lds_ptr_t)&lds[buf * A_F4 + (wmi + jj * WM) * 64];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, 0, 0, 0);
            }
        }
    };
    auto issue_w = [&](int kofs, int slot) {
#pragma unroll
        for (int jj = 0; jj < NBW; ++jj) {
            lds_ptr_t dst = (lds_ptr_t)&lds[2 * A_F4 + slot * B_F4 + (wmi + jj * WM) * 64];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, (b_off[jj] + (unsigned)kofs) * 4u, 0, 0, 0);
        }
    };

    const int r = lane & 31, h = lane >> 5;
    const int sw = swz<32>(r);
    const int idx = wmi * 32 + r;
    f32x16 acc[NB];
    f32x16 pre[NB];
    EpiCtx ec;
    epi_setup<NB, false, false>(a, m0 + idx, M, hw, n0, h, acc, pre, ec, true, false);   // operands are loaded in the epilogue
    if (KSPLIT && ks_i > 0) {      // the bias belongs to the first run only
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    }

    unsigned vmask = 0;      // validity of the 9 neighbour blocks of this lane's output pixel
    {
        const int m = m0 + idx;
        if (m < M) {
            const int img = fdiv(m, a.div_hw_mul, a.div_hw_sh), rem = m - img * hw;
            const int py = fdiv(rem, a.div_w_mul, a.div_w_sh), px = rem - py * Wb;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
                if ((unsigned)yy < (unsigned)Hb && (unsigned)xx < (unsigned)Wb) vmask |= 1u << t;
            }
        }
    }

    const int nsteps_all = a.prog_steps;
    const int mx_sa = a.mx_sa, mx_sb = a.mx_sb;
    auto kofs_of = [&](unsigned e) -> int { return (int)(((e & 15u) * (unsigned)nch2 + ((e >> 8) & 255u)) * 32u); };
    // this block's steps [s_first, nsteps]: everything, or (KSPLIT) the run from the first band start at or after its share's
    // beginning up to the step before the next run's
    int s_first = 1, nsteps = nsteps_all;
    if constexpr (KSPLIT) {
        auto band_start = [&](int s) { while (s <= nsteps_all && !(entry(s) & 16u)) ++s; return s; };
        s_first = band_start(1 + (ks_i * nsteps_all) / ksplit);
        nsteps = (ks_i + 1 == ksplit) ? nsteps_all : band_start(1 + ((ks_i + 1) * nsteps_all) / ksplit) - 1;
    }
    {
        const unsigned e1 = entry(s_first);
        if (tid < SP) lds[ZOFF + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s_first == 1) {
            const unsigned e0 = entry(0);
            issue_band((int)((e0 >> 16) & 1023u), (int)((e0 >> 26) & 3u) - 1, 0);
        } else {      // the run's first band, decoded from its own first step: K chunk cc = phase * (C / 32) + channel chunk, dy from the tap
            const int cc = (int)((e1 >> 8) & 255u), nch = C / 32, phase = cc / nch;
            issue_band((phase >> 1) | ((phase & 1) << 1) | ((cc - phase * nch) << 2), (int)(e1 & 15u) / 3 - 1, 0);
        }
        issue_w(kofs_of(e1), 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    int abuf = 0;
    for (int s = s_first; s <= nsteps; ++s) {
        const unsigned e = entry(s);
        const int slot = (s - s_first) & 1;
        if (s < nsteps) issue_w(kofs_of(entry(s + 1)), slot ^ 1);
        if ((e & 16u) && s > s_first) abuf ^= 1;
        if (e & 32u) issue_band((int)((e >> 16) & 1023u), (int)((e >> 26) & 3u) - 1, abuf ^ 1);
        const int t = (int)(e & 15u);
        const int dxi = t - (t / 3) * 3;
        const int i = idx + dxi;
        const int swi = swz<32>(i);
        const bool keep = (vmask >> t) & 1u;
        const float4* la = &lds[keep ? abuf * A_F4 + i * SP : ZOFF];      // invalid neighbour blocks read the zero row
        const float4* lb = &lds[2 * A_F4 + slot * B_F4 + r * SP];
        const SplitFrag xa = ld_split(la, h, swi);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const SplitFrag wb = ld_split(lb + nb * 32 * SP, h, sw);
            acc[nb] = mma_split(acc[nb], wb, xa, mx_sb, mx_sa);
        }
        // the next step's weight tile was requested first in this step: it must have landed; a band requested after it
        // may stay in flight only if this band has further steps (bit 6)
        if (e & 64u) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if constexpr (KSPLIT) {      // partial accumulators for conv_ksplit_epilogue_kernel (NB == 4)
        float4* o = (float4*)kws + ((((size_t)lin * ksplit + ks_i) * WM + wmi) * (NB * 4)) * 64 + lane;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) o[(nb * 4 + q) * 64] = make_float4(acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]);
    } else {
        epi_prefetch<NB, false, false>(a, n0, h, pre, ec);
        epi_finish<NB, false, false, true>(a, ec, n0, h, acc, pre, img_out);
    }
#endif
}

// The same convolution on 256-pixel tiles (round 5): conv3x3_wide_kernel's twin form for the programmed walk.  A step of the
// 128-pixel kernel above is 24 (NB = 4) or 12 (NB = 2) MFMA groups per wave between two barriers, and a step costs a block about
// 2000 cycles whatever it holds (weight-tile DMA issue and latency, fragment reads, the barrier): these layers sat at 0.34-0.40
// matrix-pipe-busy where the ConvLSTM twin kernel, with 48 groups per step, reaches 0.72 (measured at 64 sequences: enc0.conv 493
// -> 469 us, enc1.conv 461 -> 461 -- the step cadence was not the whole story --, headline +0.7 %).  Here a wave owns TWO 32-pixel blocks
// (64 pixels x 32 NB columns, a weight fragment read once feeds both), four waves = 256 pixels, ONE band buffer of 264 block rows
// (33.8 KB) + the 2-slot weight ring: 66 KB (NB = 4) / 50 KB (NB = 2) -> two / three blocks per CU.  The band is refetched behind
// the last step that reads it (barrier, request, wait, barrier -- the other block's MFMAs cover it), so the program's "request the
// next band now" bits are read at the band's FIRST step (where they sit) and honoured at its LAST.  Plain epilogues only, and no
// split K: this form is for launches that fill the chip (launch_band_prog).
template <int NB>
__global__ __launch_bounds__(256, 2) void conv_band_prog_wide_kernel(const ConvArgs* __restrict__ ap, float* __restrict__ img_out) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs& a = *ap;
    constexpr int WM = 4, MB = 2, SP = 8, TM = 32 * MB * WM;          // 256 pixels
    constexpr int A_ROWS = TM + 8, A_PIECES = A_ROWS / 8, A_F4 = A_ROWS * SP, B_F4 = 32 * NB * SP;
    constexpr int NA_MAX = (A_PIECES + WM - 1) / WM, NA_MIN = A_PIECES / WM;      // 9 / 8 band pieces per wave
    constexpr int NBW = (B_F4 / 64) / WM;                               // 4 / 2 weight-tile pieces per wave
    static_assert((B_F4 / 64) % WM == 0 && NA_MIN == 8, "tile bookkeeping");
    __shared__ __attribute__((aligned(16))) float4 lds[A_F4 + 2 * B_F4 + SP];      // [band | weight slot 0 | 1 | a zero row]
    constexpr int ZROW = (A_F4 + 2 * B_F4) / SP;                        // row index of the zero row, counted from lds[0]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Wb = a.wm, Hb = a.hm;                 // block grid = output grid
    const int hw = Hb * Wb;
    const int M = a.n * hw;
    const int nrows = a.n * Hb;
    const int ntiles = a.cout / (32 * NB);
    int lin;
    {   // XCD-aware bijective remap of the 1-D grid (block b runs on XCD b % 8)
        const int total = gridDim.x, bid = blockIdx.x;
        const int q = total >> 3, rr = total & 7, xcd = bid & 7, idx = bid >> 3;
        lin = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int ntile = lin % ntiles, mtile = lin / ntiles;
    const int m0 = mtile * TM, n0 = ntile * 32 * NB;
    const int C = a.c0;
    const int nch2 = 4 * C / 32;                    // K chunks per tap in space-to-depth form
    const int ktot = 9 * nch2 * 32;
    const unsigned row_pitch = 2u * (unsigned)a.win * (unsigned)C, pix_pitch = 2u * (unsigned)C;   // floats
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in0, (unsigned)a.n * a.hin * a.win * (unsigned)C * 4u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wgt2, (unsigned)a.cout * ktot * 4u);
    const unsigned prog_lo = a.prog[lane], prog_hi = a.prog[64 + lane];      // the step program in two VGPRs (conv_band_prog_kernel)
    auto entry = [&](int s) -> unsigned {
        return s < 64 ? (unsigned)__builtin_amdgcn_readlane((int)prog_lo, s) : (unsigned)__builtin_amdgcn_readlane((int)prog_hi, s - 64);
    };

    // DMA pieces: piece j = wv + 4 jj covers band rows 8j .. 8j + 7 (lane -> row 8j + lane / 8, 16-B slot lane % 8).  A wave's pieces
    // are 32 rows apart: one swizzle serves all of them; their (row, x) block coordinates are decoded per request -- nine multiply-
    // highs per band, a band every two or three steps -- instead of living in eighteen registers this kernel does not have
    const int row0 = 8 * wv + (lane >> 3);
    const unsigned a_q0 = (unsigned)((((lane & 7) ^ swz<32>(row0)) * 4));
    const unsigned b_off0 = (unsigned)((n0 + row0) * ktot) + a_q0;
    auto issue_band = [&](int src, int dy) {
        const int py = src & 1, px = (src >> 1) & 1, cc = src >> 2;
        const unsigned choff = (unsigned)((py * a.win + px) * C + cc * 32) + a_q0;
#pragma unroll
        for (int jj = 0; jj < NA_MAX; ++jj) {
            if (jj < NA_MIN || wv + jj * WM < A_PIECES) {      // wave-uniform
                const int mb = m0 - 1 + row0 + 32 * jj;
                unsigned voff = OOB_OFFSET;
                if ((unsigned)mb < (unsigned)M) {
                    const int rho = fdiv(mb, a.div_w_mul, a.div_w_sh), x = mb - rho * Wb;
                    if ((unsigned)(rho + dy) < (unsigned)nrows) voff = ((unsigned)(rho + dy) * row_pitch + (unsigned)x * pix_pitch + choff) * 4u;
                }
                lds_ptr_t dst = (lds_ptr_t)&lds[(wv + jj * WM) * 64];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, 0, 0, 0);
            }
        }
    };
    auto issue_w = [&](int kofs, int slot) {
#pragma unroll
        for (int jj = 0; jj < NBW; ++jj) {
            lds_ptr_t dst = (lds_ptr_t)&lds[A_F4 + slot * B_F4 + (wv + jj * WM) * 64];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, b_off0 * 4u, ((unsigned)kofs + (unsigned)(8 * WM * jj * ktot)) * 4u, 0, 0);
        }
    };

    const int r = lane & 31, h = lane >> 5;
    const int sw = swz<32>(r);
    f32x16 acc0[NB], acc1[NB];
    f32x16 late[NB];
    EpiCtx ec0, ec1;
    const int mA = m0 + wv * 64 + r, mB = mA + 32;
    epi_setup<NB, false, false>(a, mA, M, hw, n0, h, acc0, late, ec0, true, false);   // operands are loaded in the epilogue
    epi_setup<NB, false, false>(a, mB, M, hw, n0, h, acc1, late, ec1, true, false);
    auto neighbours = [&](int m) -> unsigned {   // validity of the 9 neighbour blocks of the lane's output pixel
        unsigned vm = 0;
        if (m < M) {
            const int img = fdiv(m, a.div_hw_mul, a.div_hw_sh), rem = m - img * hw;
            const int py = fdiv(rem, a.div_w_mul, a.div_w_sh), px = rem - py * Wb;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
                if ((unsigned)yy < (unsigned)Hb && (unsigned)xx < (unsigned)Wb) vm |= 1u << t;
            }
        }
        return vm;
    };
    const unsigned vmask0 = neighbours(mA), vmask1 = neighbours(mB);
    const int nsteps = a.prog_steps;
    const int mx_sa = a.mx_sa, mx_sb = a.mx_sb;
    auto kofs_of = [&](unsigned e) -> int { return (int)(((e & 15u) * (unsigned)nch2 + ((e >> 8) & 255u)) * 32u); };
    {
        const unsigned e0 = entry(0);
        if (tid < SP) lds[A_F4 + 2 * B_F4 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        issue_band((int)((e0 >> 16) & 1023u), (int)((e0 >> 26) & 3u) - 1);
        issue_w(kofs_of(entry(1)), 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    unsigned next_band = 0;      // bits 16-27 of the current band's first step: the band after it
    for (int s = 1; s <= nsteps; ++s) {
        const unsigned e = entry(s);
        const int slot = (s - 1) & 1;
        const unsigned e_next = s < nsteps ? entry(s + 1) : 0u;
        if (s < nsteps) issue_w(kofs_of(e_next), slot ^ 1);
        if (e & 16u) next_band = e;
        __builtin_amdgcn_sched_barrier(0);
        const int t = (int)(e & 15u);
        const int dxi = t - (t / 3) * 3;
        int l0 = wv * 64 + r + dxi;                 // band row of the lane's (dx) neighbour block
        asm volatile("" : "+v"(l0));
        const int l1 = l0 + 32;
        const int i0 = ((vmask0 >> t) & 1u) ? l0 : ZROW, i1 = ((vmask1 >> t) & 1u) ? l1 : ZROW;      // invalid neighbours read the zero row
        const float4* lb = &lds[A_F4 + slot * B_F4 + r * SP];
        const SplitFrag xa0 = ld_split(&lds[i0 * SP], h, swz<32>(l0));
        const SplitFrag xa1 = ld_split(&lds[i1 * SP], h, swz<32>(l1));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const SplitFrag wb = ld_split(lb + nb * 32 * SP, h, sw);
            acc0[nb] = mma_split(acc0[nb], wb, xa0, mx_sb, mx_sa);
            acc1[nb] = mma_split(acc1[nb], wb, xa1, mx_sb, mx_sa);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (e_next & 16u) {      // this was the band's last step and everyone has left it: fetch the next one into the same buffer
            issue_band((int)((next_band >> 16) & 1023u), (int)((next_band >> 26) & 3u) - 1);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    // plain epilogues, one pixel block at a time; the second block's accumulators wait in LDS (idle now), as in conv3x3_wide_kernel
    float4* park = &lds[wv * (NB * 4 * 64)];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            park[(nb * 4 + q) * 64 + lane] = make_float4(acc1[nb][4 * q], acc1[nb][4 * q + 1], acc1[nb][4 * q + 2], acc1[nb][4 * q + 3]);
    if constexpr (FMT != 3) {
        {
            f32x16 op4[NB];
            epi_prefetch<NB, false, false>(a, n0, h, op4, ec0);
            epi_finish<NB, false, false, true>(a, ec0, n0, h, acc0, op4, img_out);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            f32x16 op4[NB];
            epi_prefetch<NB, false, false>(a, n0, h, op4, ec1);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = park[(nb * 4 + q) * 64 + lane];
                    acc0[nb][4 * q] = v.x; acc0[nb][4 * q + 1] = v.y; acc0[nb][4 * q + 2] = v.z; acc0[nb][4 * q + 3] = v.w;
                }
            epi_finish<NB, false, false, true>(a, ec1, n0, h, acc0, op4, img_out);
        }
    } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x16 one[1], op[1];
            one[0] = acc0[nb];
            epi_prefetch<1, false, false>(a, n0 + 32 * nb, h, op, ec0);
            epi_finish<1, false, false, true>(a, ec0, n0 + 32 * nb, h, one, op, img_out);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x16 one[1], op[1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = park[(nb * 4 + q) * 64 + lane];
                one[0][4 * q] = v.x; one[0][4 * q + 1] = v.y; one[0][4 * q + 2] = v.z; one[0][4 * q + 3] = v.w;
            }
            epi_prefetch<1, false, false>(a, n0 + 32 * nb, h, op, ec1);
            epi_finish<1, false, false, true>(a, ec1, n0 + 32 * nb, h, one, op, img_out);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#endif
}

template <int NB>
static int launch_band_prog(const ConvArgs& a, const ConvArgs* d_args, hipStream_t stream, float* img) {
    const int M = a.n * a.hm * a.wm;
    const int total = ((M + 127) / 128) * (a.cout / (32 * NB));
    int ks = 1;
    if constexpr (NB == 4) {      // split K for under-filled launches (launch_band's rule; runs are whole bands of the step program)
        static const int ks_max = getenv("EVR_KSPLIT") ? atoi(getenv("EVR_KSPLIT")) : 4;
        const int nbands = a.prog_steps / 2;      // (>= 2 steps per band on average: an upper bound that keeps every run non-empty)
        if (ks_max > 1 && total <= 192 && nbands >= 4) {
            ks = 512 / total;
            if (ks > nbands / 2) ks = nbands / 2;
            if (ks > ksplit_cap(total, ks_max)) ks = ksplit_cap(total, ks_max);
            if (ks < 2) ks = 1;
        }
    }
    {   // 256-pixel tiles once a launch has enough of them to fill the chip (two or three blocks per CU; EVR_PROG_WIDE=0: never)
        static const int pw_on = getenv("EVR_PROG_WIDE") ? atoi(getenv("EVR_PROG_WIDE")) : 1;
        static const int pw_min = getenv("EVR_PROG_WIDE_MIN") ? atoi(getenv("EVR_PROG_WIDE_MIN")) : 600;
        const int total2 = ((M + 255) / 256) * (a.cout / (32 * NB));
        if (pw_on && total2 >= pw_min && ks == 1) {
            hipLaunchKernelGGL((conv_band_prog_wide_kernel<NB>), dim3(total2), dim3(256), 0, stream, d_args, img);
            EVR_LAUNCH_CHECK();
            return EVR_OK;
        }
    }
    float* kws = nullptr;
    if (ks > 1) { kws = ksplit_workspace(a, (size_t)total * ks * 4 * 16 * 64 * sizeof(float4)); if (!kws) ks = 1; }
    if constexpr (NB == 4) {
        if (ks > 1) {
            hipLaunchKernelGGL((conv_band_prog_kernel<NB, true>), dim3(total * ks), dim3(256), 0, stream, d_args, img, ks, kws);
            EVR_LAUNCH_CHECK();
            launch_ksplit_epilogue<false, false>(d_args, img, total, ks, (const float*)kws, stream);
            EVR_LAUNCH_CHECK();
            return EVR_OK;
        }
    }
    hipLaunchKernelGGL((conv_band_prog_kernel<NB, false>), dim3(total), dim3(256), 0, stream, d_args, img, 1, (float*)nullptr);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

static bool band_prog_eligible(const ConvArgs& a, int kc) {
    static const bool off = getenv("EVR_NO_BAND") != nullptr;
    if (off || !a.prog || !a.wgt2) return false;
    if (!(a.x3 && a.in_packed && kc == 32 && a.tp.ngroups == 1 && a.tp.ntaps == 25 && a.stride == 2 && a.os == 1)) return false;
    if (a.in_mode != IN_SINGLE || a.hin != 2 * a.hm || a.win != 2 * a.wm || a.hout != a.hm || a.wout != a.wm) return false;
    if (a.cout % 64 != 0 || a.pred_w) return false;
    // measured (64 sequences of 352x264): 64 columns (enc0) 595 -> 510 us with three bf16 products; at 128/256 columns the
    // band form only tied the implicit GEMM then, but with 2/3 of the matrix cycles the implicit GEMM's 25-fold A re-fetch
    // (64 B/clk/CU from L2) is the limit: 420/378 -> 395/360 us.  EVR_BAND_PROG_ALL=0 restores the implicit GEMM there.
    static const bool all = getenv("EVR_BAND_PROG_ALL") ? atoi(getenv("EVR_BAND_PROG_ALL")) != 0 : true;
    if (a.cout % 128 == 0 && !all) return false;
    const int nb = (a.cout % 128 == 0) ? 4 : 2;
    const int64_t M = (int64_t)a.n * a.hm * a.wm;
    static const int min_blocks = getenv("EVR_BAND_MIN") ? atoi(getenv("EVR_BAND_MIN")) : 1;
    return ((M + 127) / 128) * (a.cout / (32 * nb)) >= min_blocks;
}

template <int KC, int WM, int NB, bool LSTM, bool GROUPED, bool REGSTAGE = false, int X3 = 0>
static int launch_t(const ConvArgs& a, const ConvArgs* d_args, hipStream_t stream, float* img) {
    const int M = a.n * a.hm * a.wm;
    const int mtiles = (M + 32 * WM - 1) / (32 * WM);
    const int ntiles = a.cout / (32 * NB);
    const int total = mtiles * ntiles;
    hipLaunchKernelGGL((conv_igemm_kernel<KC, WM, NB, LSTM, GROUPED, REGSTAGE, X3>), dim3(total), dim3(64 * WM), 0, stream, d_args, img);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

}  // namespace EVR_ANS
using namespace EVR_ANS;

int EVR_LAUNCH_NAME(const ConvArgs& a, const ConvArgs* d_args, int kc, int wm, int nb, hipStream_t stream, float* img) {
    EVR_REQUIRE(!a.pred_w || (a.cout == 32 * nb && img), "conv_igemm: fused prediction needs a single N tile and an image pointer");
    EVR_REQUIRE(a.tp.grp_cols % 32 == 0 && a.tp.ngroups <= MAX_PHASES &&
                (a.tp.inter ? (a.tp.ngroups == 4 && a.tp.grp_cols == 32 && a.cout % 128 == 0 && !a.pred_w) : a.tp.ngroups * a.tp.grp_cols == a.cout),
                "conv_igemm: bad column groups");
    EVR_REQUIRE(a.cout % (32 * nb) == 0, "conv_igemm: cout %d not a multiple of %d", a.cout, 32 * nb);
    EVR_REQUIRE(a.c0 % kc == 0 && (a.in_mode != IN_CAT || a.c1 % kc == 0), "conv_igemm: channels %d/%d not multiples of %d", a.c0, a.c1, kc);
    EVR_REQUIRE(a.epi != EPI_LSTM || nb == 4, "conv_igemm: the LSTM epilogue needs nb == 4");
    EVR_REQUIRE((int64_t)a.n * a.hm * a.wm < (1LL << 31), "conv_igemm: M too large");
#if EVR_ARITH == 3
    if (c16_eligible(a, kc)) {
        EVR_REQUIRE(a.acc_scale > 0.f && a.div_hw_sh < 32, "conv3x3_c16: plan without accumulator scale / fastdiv");
        if (c16_rows_eligible(a)) return launch_c16_rows<1>(a, d_args, stream, img);
        return launch_c16<1>(a, d_args, stream, img);
    }
#endif
    EVR_REQUIRE(!a.x3 || kc == 32, "conv_igemm: the split path needs 32-channel chunks");
    EVR_REQUIRE(a.n_valid % 4 == 0 && a.cout_total % 4 == 0, "conv_igemm: output channels %d/%d not multiples of 4 (16-B epilogue accesses)", a.n_valid, a.cout_total);
    EVR_REQUIRE(a.epi != EPI_LSTM || (a.os == 1 && a.hout == a.hm && a.wout == a.wm && a.hidden % 32 == 0),
                "conv_igemm: the ConvLSTM epilogue writes the state grid itself (stride 1, hidden %% 32 == 0)");
    EVR_REQUIRE((a.epi != EPI_GRU_ZR && a.epi != EPI_GRU_OUT) || a.hidden % 4 == 0, "conv_igemm: ConvGRU hidden %d not a multiple of 4", a.hidden);
    const bool packed_io = a.in_packed || a.out_packed || a.res_packed || a.padd_packed || a.state_packed;
    EVR_REQUIRE(!packed_io || a.x3, "conv_igemm: PACKED tensors need the split mode");
    EVR_REQUIRE(!a.x3 || a.in_packed, "conv_igemm: the split kernels take PACKED inputs");
    EVR_REQUIRE(a.x3 != 2 || (a.mx_sa > 0 && a.mx_sb > 0), "conv_igemm: split mode without block scales");
    EVR_REQUIRE(a.x3 != 3 || a.acc_scale > 0.f, "conv_igemm: three-product mode without the accumulator scale");
    EVR_REQUIRE(a.x3 == 0 || a.x3 == 2 || a.x3 == 3 || a.x3 == 4, "conv_igemm: unknown arithmetic mode %d", a.x3);
#if EVR_ARITH == 3
    EVR_REQUIRE(a.x3 == 3, "conv_igemm: this object carries the three-f16-product kernels only (mode %d requested)", a.x3);
#elif EVR_ARITH == 4
    EVR_REQUIRE(a.x3 == 4, "conv_igemm: this object carries the f16 + fp6 kernels only (mode %d requested)", a.x3);
    EVR_REQUIRE(a.acc_scale > 0.f, "conv_igemm: f16 + fp6 mode without the accumulator scale");
    // P6 tensors have a scale per 16-channel group: they are written as whole groups only
    EVR_REQUIRE(a.group_store && a.epi != EPI_GRU_ZR && a.epi != EPI_GRU_OUT && a.epi != EPI_BIAS_TANH && !(a.pred_w && a.out && a.out_packed),
                "conv_igemm: the f16 + fp6 mode writes whole 64-B groups only (epilogue %d)", a.epi);
    EVR_REQUIRE(!a.out_packed || (a.n_valid % 16 == 0 && a.cout_total % 16 == 0), "conv_igemm: P6 output needs channel counts that are multiples of 16");
#else
    EVR_REQUIRE(a.x3 != 3 && a.x3 != 4, "conv_igemm: the three-f16-product / f16 + fp6 kernels live in the other objects");
#endif
    EVR_REQUIRE(a.div_hw_sh < 32 && a.div_w_sh < 32 && (a.div_hw_mul || (unsigned)(a.hm * a.wm) == (1u << a.div_hw_sh)) && (a.div_w_mul || (unsigned)a.wm == (1u << a.div_w_sh)),
                "conv_igemm: plan without set_fastdiv()");
    EVR_REQUIRE(!a.out_packed || (a.n_valid % 8 == 0 && a.cout_total % 8 == 0), "conv_igemm: PACKED output needs channel counts that are multiples of 8");
    const int mode = a.x3 ? 2 : 0;
    if (band_prog_eligible(a, kc)) {
        if (a.cout % 128 == 0) return launch_band_prog<4>(a, d_args, stream, img);
        return launch_band_prog<2>(a, d_args, stream, img);
    }
    if (bandk_eligible(a, kc, 5)) {
        if (a.cout % 128 == 0) return launch_bandk<5, 4>(a, d_args, stream, img);
        if (a.cout % 64 == 0) return launch_bandk<5, 2>(a, d_args, stream, img);
        return launch_bandk<5, 1>(a, d_args, stream, img);
    }
    if (bandk_eligible(a, kc, 3)) {
        if (a.cout % 64 == 0) return launch_bandk<3, 2>(a, d_args, stream, img);
        return launch_bandk<3, 1>(a, d_args, stream, img);
    }
    if (band_eligible(a, kc)) {
        // 128 x 128 tiles: 4 waves x 2-slot ring, two blocks per CU (one block's epilogue and barrier bubbles hide under the
        // other's MFMAs).  The 8-wave and overlapped-tile / 3-slot-ring configurations of earlier rounds measured slower
        // (profiles/r02_tile_variants.txt) and are gone: with the zero row their 80 KiB no longer leave two blocks per CU.
        // 256-pixel block tiles when N allows and there are enough of them to fill the chip (EVR_WIDE=0: never)
        static const int wide = getenv("EVR_WIDE") ? atoi(getenv("EVR_WIDE")) : 1;
        static const int wide_min = getenv("EVR_WIDE_MIN") ? atoi(getenv("EVR_WIDE_MIN")) : 600;   // (round 5: 600 tiles, was 1024 -- +2 % at 8 / 16 / 32 sequences, equal at 4 and 64; 300: -1 % at 16 / 32.  Tests lower it)
        // two 256 x 128 blocks per CU (one band buffer each) while K is short, one 256 x 256 block otherwise: measured
        // 1486 / 1449 / 1405 us against 1585 / 1489 / 1401 us for 128 / 256 / 512 input channels (EVR_WIDE=2 / 3 force one)
        const bool twin = (wide == 2) || (wide == 1 && a.c0 + a.c1 <= 256);
        // (the twin form already pays from ~1.4 rounds of its 512 slots on: per-layer times at 4 / 8 sequences, enc0.rec 144 -> 138 us
        // with 726 twin tiles, enc1.rec 262 -> 249 us with 728; below one round it loses -- enc1.rec 136 -> 151 us with 364 tiles -- and
        // the 256 x 256 form loses up to its own threshold: enc2.rec 256 -> 279 us at 8 sequences.  Explicit EVR_WIDE_MIN: one threshold)
        static const int wide_min_twin = getenv("EVR_WIDE_MIN") ? wide_min : 350;
        const bool wide_ok = wide && a.tp.ngroups == 1 && a.cout % 256 == 0 && !a.pred_w &&
                             (((int64_t)a.n * a.hm * a.wm + 255) / 256) * (a.cout / 256) >= (twin ? wide_min_twin : wide_min);
        if (wide_ok && a.epi == EPI_LSTM) {
            return twin ? launch_wide<true, 1>(a, d_args, stream, img) : launch_wide<true, 2>(a, d_args, stream, img);
        }
        // plain 3x3 layers (residual blocks): the twin form once a launch has enough 256 x 128 tiles (not at 64 sequences of
        // 346x260: 726 tiles for 512 slots; from 128 sequences or 640x480 up)
        // (their own threshold: at 64 sequences the residual convolutions are 726 such tiles -- 1.42 rounds of 512 slots -- and run
        // 281 us on this form against 258 us on the 128 x 128 tiles; the ConvLSTM threshold above went down to 600 in round 5)
        static const int wide_min_plain = getenv("EVR_WIDE_MIN_PLAIN") ? atoi(getenv("EVR_WIDE_MIN_PLAIN")) : (wide_min > 1024 ? wide_min : (getenv("EVR_WIDE_MIN") ? wide_min : 1024));
        if (wide && a.tp.ngroups == 1 && a.cout % 128 == 0 && !a.pred_w &&
            (a.epi == EPI_BIAS || a.epi == EPI_BIAS_RELU || a.epi == EPI_RESIDUAL_RELU) &&
            (((int64_t)a.n * a.hm * a.wm + 255) / 256) * (a.cout / 128) >= wide_min_plain)
            return launch_wide<false, 1>(a, d_args, stream, img);
        if (a.epi == EPI_LSTM) {
            return launch_band<4, 2, true>(a, d_args, stream, img);
        }
        if (a.tp.ngroups > 1) {
            {
                // does the tap/phase connectivity follow the k5 s2 p2 rule the PHASES variants compile in?
                bool rule = a.tp.ngroups == 4;
                for (int t = 0; t < 9 && rule; ++t) {
                    int want = 0;
                    for (int g = 0; g < 4; ++g) if (!((g >> 1) == 1 && t / 3 == 0) && !((g & 1) == 1 && t % 3 == 0)) want |= 1 << g;
                    rule = a.tp.tap_groups[t] == want;
                }
                for (int g = 0; g < 4 && rule; ++g) rule = a.tp.grp_ofy[g] == (g >> 1) && a.tp.grp_ofx[g] == (g & 1);
                // 256 x 128 tiles, two blocks per CU (the twin form of conv3x3_wide_kernel), once there are enough of them:
                // half the LDS-DMA bytes per MFMA cycle of the 128 x 128 tiles (EVR_WIDE_DEC=0: never)
                static const int wide_dec = getenv("EVR_WIDE_DEC") ? atoi(getenv("EVR_WIDE_DEC")) : 1;
                // (dec2 -- 32 output channels, 18 steps -- gains from 363 such tiles on: 49 -> 39 us at 4 sequences; dec1 at 364 loses, 83 -> 92 us)
                static const int wide_min_dec32 = getenv("EVR_WIDE_MIN") ? wide_min : 350;
                const bool twin_ok = wide_dec && rule && (((int64_t)a.n * a.hm * a.wm + 255) / 256) * (a.cout / 128) >= (a.tp.grp_cols == 32 ? wide_min_dec32 : wide_min) &&
                                     (a.epi == EPI_BIAS || a.epi == EPI_BIAS_RELU) && (!a.pred_w || a.tp.grp_cols == 32);   // (a fused
                // prediction must close inside one 32-column block: the epilogue runs block by block)
                if (twin_ok && a.tp.grp_cols == 32) return launch_wide<false, 1, true, 1>(a, d_args, stream, img);
                if (twin_ok && a.tp.grp_cols == 64) return launch_wide<false, 1, true, 2>(a, d_args, stream, img);
                if (twin_ok && a.tp.grp_cols % 128 == 0) return launch_wide<false, 1, true, 0>(a, d_args, stream, img);
                if (rule && a.tp.grp_cols == 32) return launch_band<4, 2, false, true, false, 1>(a, d_args, stream, img);
                if (rule && a.tp.grp_cols == 64) return launch_band<4, 2, false, true, false, 2>(a, d_args, stream, img);
                return launch_band<4, 2, false, true>(a, d_args, stream, img);
            }
            return launch_band<4, 2, false, true>(a, d_args, stream, img);
        }
        return launch_band<4, 2, false>(a, d_args, stream, img);
    }
    if (a.epi == EPI_LSTM) {
        EVR_REQUIRE(kc == 32, "conv_igemm: ConvLSTM needs 32-channel chunks");
        if (mode == 2) {
            if (wm == 8) return launch_t<32, 8, 4, true, false, true, 2>(a, d_args, stream, img);
            if (wm == 4) return launch_t<32, 4, 4, true, false, true, 2>(a, d_args, stream, img);
            if (wm == 2) return launch_t<32, 2, 4, true, false, false, 2>(a, d_args, stream, img);
            return launch_t<32, 1, 4, true, false, false, 2>(a, d_args, stream, img);
        }
#if EVR_ARITH == 2
        if (wm == 8) return launch_t<32, 8, 4, true, false>(a, d_args, stream, img);
        // register staging measured +1..4 % over LDS-DMA for this kernel (EVR_LSTM_DMA=1 selects the DMA loader)
        if (wm == 4 && !getenv("EVR_LSTM_DMA")) return launch_t<32, 4, 4, true, false, true>(a, d_args, stream, img);
        if (wm == 4) return launch_t<32, 4, 4, true, false>(a, d_args, stream, img);
        if (wm == 2) return launch_t<32, 2, 4, true, false>(a, d_args, stream, img);
        return launch_t<32, 1, 4, true, false>(a, d_args, stream, img);
#endif
    }
    if (a.tp.ngroups > 1) {   // transposed conv: column groups = sub-pixel phases
#if EVR_ARITH == 2
#define EVR_CASEG(WM_, NB_) if (kc == 32 && wm == WM_ && nb == NB_) { if (mode == 2) return launch_t<32, WM_, NB_, false, true, false, 2>(a, d_args, stream, img); return launch_t<32, WM_, NB_, false, true>(a, d_args, stream, img); }
#else
#define EVR_CASEG(WM_, NB_) if (kc == 32 && wm == WM_ && nb == NB_) { return launch_t<32, WM_, NB_, false, true, false, 2>(a, d_args, stream, img); }
#endif
        EVR_CASEG(4, 4) EVR_CASEG(2, 4) EVR_CASEG(1, 4) EVR_CASEG(4, 2) EVR_CASEG(2, 2) EVR_CASEG(1, 2)
        EVR_CASEG(4, 1) EVR_CASEG(2, 1) EVR_CASEG(1, 1)
#undef EVR_CASEG
        set_error("conv_igemm: no grouped kernel for kc=%d wm=%d nb=%d", kc, wm, nb);
        return EVR_ERR_UNSUPPORTED;
    }
    if (mode != 0) {
#define EVR_CASEX(WM_, NB_) if (wm == WM_ && nb == NB_) return launch_t<32, WM_, NB_, false, false, false, 2>(a, d_args, stream, img);
        EVR_CASEX(4, 4) EVR_CASEX(2, 4) EVR_CASEX(1, 4) EVR_CASEX(4, 2) EVR_CASEX(2, 2) EVR_CASEX(1, 2)
        EVR_CASEX(4, 1) EVR_CASEX(2, 1) EVR_CASEX(1, 1)
#undef EVR_CASEX
    }
#if EVR_ARITH == 2
#define EVR_CASE(KC_, WM_, NB_) if (kc == KC_ && wm == WM_ && nb == NB_) return launch_t<KC_, WM_, NB_, false, false>(a, d_args, stream, img);
    EVR_CASE(32, 4, 4) EVR_CASE(32, 2, 4) EVR_CASE(32, 1, 4)
    EVR_CASE(32, 4, 2) EVR_CASE(32, 2, 2) EVR_CASE(32, 1, 2)
    EVR_CASE(32, 4, 1) EVR_CASE(32, 2, 1) EVR_CASE(32, 1, 1)
    EVR_CASE(16, 4, 1) EVR_CASE(16, 2, 1) EVR_CASE(16, 1, 1)
#undef EVR_CASE
#endif
    set_error("conv_igemm: no kernel for kc=%d wm=%d nb=%d", kc, wm, nb);
    return EVR_ERR_UNSUPPORTED;
}

}  // namespace evr

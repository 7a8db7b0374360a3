// Winograd F(2x2, 3x3) form of the 3x3 stride-1 'same' convolutions in exact fp32 -- the ConvLSTM gate convolutions
// (model/submodules.py:227-245, 65 % of E2VID's FLOPs) and the residual blocks' convolutions (:169-184) of the library's
// exact-fp32 mode (EVR_FP32=1, the saturation re-run twin): 16 multiplies per 2x2 output tile and (cin, cout) pair instead
// of 36, i.e. 2.25x fewer v_mfma_f32_32x32x2_f32 -- the one instruction that mode is bound by (0.81 matrix-busy in round 5).
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs   (Lavin & Gray 2016)
//
//   GEMM view   16 independent GEMMs (one per transform-domain position) with M = n*th*tw tiles of 2x2 output pixels
//               (flattened), N = output channels, K = input channels (cat(x, h) for the ConvLSTM)
//   weights     U = G g G^T computed in fp64 on the host at model creation (wino_pack_weights), stored in the exact order
//               the kernel's LDS image wants: [N/64][K/8][pos 16][wc 2][kp 2][row 32][4 k] -- a block's K chunk is ONE
//               contiguous 32 KB run, fetched by LDS-DMA, and a wave's A fragment is a lane-linear ds_read_b128
//   block       256 threads = 4 waves = 2 (tile halves) x 2 (column halves): 64 tiles x 64 columns; a wave owns 32 tiles x
//               32 columns x 16 positions = 16 accumulators of 16 registers = the 256 AGPRs of a one-wave-per-SIMD kernel
//   input       per K chunk (8 channels) every thread owns one (tile, channel pair): 16 8-byte loads of its 4x4 patch (zero
//               padding and ragged M from the buffer descriptor's range check), B^T d B in registers (32 packed adds), 16
//               8-byte LDS stores into the V image [pos][wt][kp][tile 32][4 k]; the loads run two chunks ahead of the MFMAs
//   ConvLSTM    the wave's 32 columns are the 4 gates of 8 hidden channels, rows ordered so that a lane (= one tile) holds
//               all four gates of 4 channels: output transform and cell update are a register epilogue
//   grid        1-D, XCD-aware remap, column block fastest (the blocks that share input tiles run together)
#include "conv.h"
#include <cstdlib>
#include <type_traits>

namespace evr {

// G g G^T for every (row, input channel) of w = [n_gemm][9][cin] (prep_conv2d's layout: tap = ky*3 + kx), in fp64, laid out
// as the kernel streams it.  lstm_hidden > 0: the rows of w are the ConvLSTM permutation (c/32)*128 + gate*32 + c%32.
void wino_pack_weights(const std::vector<float>& w, int n_gemm, int cin, int lstm_hidden, std::vector<float>& out) {
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int ncb = n_gemm / 64, nch = cin / 8;
    out.assign((size_t)n_gemm * cin * 16, 0.f);
    for (int cb = 0; cb < ncb; ++cb)
        for (int wc = 0; wc < 2; ++wc)
            for (int rho = 0; rho < 32; ++rho) {
                int src;
                if (lstm_hidden > 0) {
                    const int gate = rho >> 3, ch = cb * 16 + wc * 8 + (rho & 7);
                    src = (ch / 32) * 128 + gate * 32 + (ch % 32);
                } else {
                    src = cb * 64 + wc * 32 + rho;
                }
                for (int k = 0; k < cin; ++k) {
                    double g[3][3];
                    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = (double)w[((size_t)src * 9 + t) * cin + k];
                    double tmp[4][3];
                    for (int xi = 0; xi < 4; ++xi)
                        for (int kx = 0; kx < 3; ++kx) tmp[xi][kx] = G[xi][0] * g[0][kx] + G[xi][1] * g[1][kx] + G[xi][2] * g[2][kx];
                    const int c = k / 8, kp = (k % 8) / 4, kk = k % 4;
                    for (int xi = 0; xi < 4; ++xi)
                        for (int nu = 0; nu < 4; ++nu) {
                            const double u = tmp[xi][0] * G[nu][0] + tmp[xi][1] * G[nu][1] + tmp[xi][2] * G[nu][2];
                            const int pos = xi * 4 + nu;
                            out[((((((size_t)cb * nch + c) * 16 + pos) * 2 + wc) * 2 + kp) * 32 + rho) * 4 + kk] = (float)u;
                        }
                }
            }
}

bool wino_enabled() {
    static const bool on = getenv("EVR_WINO") ? atoi(getenv("EVR_WINO")) != 0 : true;
    return on;
}

// host-side test of a launch plan (model.cpp sets wgt_wino only for layers that pass the shape part of this)
bool wino_eligible(const ConvArgs& a) {
    if (!a.wgt_wino || a.x3 != 0) return false;
    if (a.wino_s2d) {
        // a k5 stride-2 convolution as the 3x3 stride-1 convolution over 2x2 pixel blocks it is (model.cpp prep_s2d): 4 cin channels
        // (phase-major) on the hm x wm block grid -- 16 instead of 25 multiplies per output and input channel
        if (a.tp.ntaps != 25 || a.stride != 2 || a.tp.ngroups != 1 || a.os != 1 || a.in_mode != IN_SINGLE) return false;
        if (a.hin != 2 * a.hm || a.win != 2 * a.wm || a.hout != a.hm || a.wout != a.wm) return false;
        if (a.c0 % 8 || a.wino_s2d != 1 + __builtin_ctz((unsigned)(a.c0 / 8)) || (a.c0 / 8) != (1 << (a.wino_s2d - 1))) return false;
        if (a.cout % 64 || a.n_valid != a.cout || a.cout_total != a.cout || a.pred_w || !a.out || a.residual) return false;
        if (a.in_packed || a.out_packed || a.padd_packed) return false;
        if ((int64_t)a.n * a.hout * a.wout * a.cout_total * 4 > 0xBFFF0000LL) return false;
        return a.epi == EPI_BIAS || a.epi == EPI_BIAS_RELU;
    }
    if (a.tp.ntaps != 9 || a.stride != 1) return false;
    if (a.hm != a.hin || a.wm != a.win) return false;
    // a transposed convolution (k5 s2: four sub-pixel phases of <= 3x3 taps on the INPUT grid, prep_tconv) is one 3x3 convolution with
    // 4 x cout columns whose 32-column blocks each belong to one phase: same transforms, the epilogue scatters to (2y + py, 2x + px)
    const bool tconv = a.tp.ngroups == 4 && a.os == 2;
    if (tconv) {
        if (a.hout != 2 * a.hm || a.wout != 2 * a.wm || a.cout != 4 * a.cout_total || a.n_valid != a.cout_total || a.cout_total % 32) return false;
        if (!(a.tp.inter ? a.tp.grp_cols == 32 : a.tp.grp_cols == a.cout_total)) return false;
        if (a.in_mode != IN_SINGLE || a.residual) return false;
    } else {
        if (a.tp.ngroups != 1 || a.os != 1 || a.hout != a.hm || a.wout != a.wm || a.n_valid != a.cout) return false;
    }
    if (a.c0 % 8 || (a.in_mode == IN_CAT && a.c1 != a.c0) || a.cout % 64) return false;
    if (((a.c0 + (a.in_mode == IN_CAT ? a.c1 : 0)) / 8) % 2) return false;      // (the chunk loop is unrolled by its two LDS buffers)
    if (a.in_packed || a.out_packed || a.res_packed || a.padd_packed || a.state_packed) return false;
    if ((int64_t)a.n * a.hout * a.wout * (a.epi == EPI_LSTM ? a.hidden : a.cout_total) * 4 > 0xBFFF0000LL) return false;      // (EPI_OOB, wino.hip)
    if (a.epi == EPI_LSTM) return !tconv && a.hidden % 16 == 0 && a.cout == 4 * a.hidden;
    // (a fused 1x1 prediction layer: only where a wave's 32 columns are ALL of a pixel's channels -- the last transposed decoder; its skip
    // term is the whole skip tensor (post_add: the exact-fp32 head kernel writes no per-pixel dot product) or one float per pixel)
    if (a.pred_w) return tconv && a.cout_total == 32 && (a.epi == EPI_BIAS || a.epi == EPI_BIAS_RELU);
    if (!a.out) return false;
    if (tconv) return a.epi == EPI_BIAS || a.epi == EPI_BIAS_RELU;
    return (a.epi == EPI_BIAS || a.epi == EPI_BIAS_RELU || a.epi == EPI_RESIDUAL_RELU) && a.cout_total == a.cout;
}

namespace wino {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int fdiv(int n, unsigned mul, unsigned sh) { return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> sh); }
// Gate activations: FAST (the default here) = v_exp_f32 / v_rcp_f32 forms, absolute error ~2e-7 -- a fifth of what the fp32 summation
// order of a 4608-term convolution leaves (1e-6); EVR_WINO_FASTACT=0: libm-grade (3200 instead of 350 instructions per tile and lane)
template <bool FAST> __device__ __forceinline__ float sigmoid_t(float x) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    else return 1.0f / (1.0f + expf(-x));
}
template <bool FAST> __device__ __forceinline__ float tanh_t(float x) {
    if constexpr (FAST) return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
    else return tanhf(x);
}

// two channels per instruction (hipcc splits the <2 x float> adds of the input transform into scalar v_add_f32 pairs)
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) { f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f2 pk_sub(f2 a, f2 b) { f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned OOB_OFFSET = 0xFFFFFFF0u;   // >= num_records of every descriptor -> the load returns 0
// the epilogue ADDS up to 96 bytes to a pixel's offset (channel runs of one lane): its marker must not wrap past 2^32 -- 0xFFFFFFF0 + 32
// is byte 16 of the tensor, and the dropped store of a pixel that does not exist became a store into pixel 0.  Output tensors beyond
// 3 GiB stay on the direct form (wino_eligible)
constexpr unsigned EPI_OOB = 0xC0000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
#endif

constexpr int UV_F4 = 2048;      // float4 per 32-KB image: [pos 16][half 2][lane 64]

// VAR (timing experiments, EVR_WINO_VAR; bit 0 -- the MFMAs of two positions interleaved so that no two consecutive ones share an
// accumulator -- measured no different and is gone): bit 1 = no side work at all (no DMA, transform or patch loads after the first chunk: results are garbage); bit 2 = side work on
// cache-hot addresses (always chunk 0's weights and patch: garbage) -- separates the side work's issue cost from its memory latency;
// bits 3 / 4 / 5 = no weight DMA / no patch loads / no transform and V stores after an item's first chunk
template <bool LSTM, bool FAST, int VAR = 0, bool PRED = false, bool S2D = false>
__global__ __launch_bounds__(256) void wino_f32_kernel(const ConvArgs* __restrict__ ap, int total, float* __restrict__ img_out) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs& a = *ap;
    // four separate LDS objects: hipcc's waitcnt insertion then knows that an LDS-DMA into one U buffer does not alias the fragment
    // reads of the other (with one array it puts s_waitcnt vmcnt(0) in front of every chunk's first ds_read, i.e. waits for the
    // weight image it has just requested)
    __shared__ __attribute__((aligned(16))) float4 ldsU0[UV_F4], ldsU1[UV_F4], ldsV0[UV_F4], ldsV1[UV_F4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wv & 1, wc = wv >> 1;
    const int s2d = S2D ? a.wino_s2d : 0;      // (0, or 1 + log2(cin / 8): the input is read in space-to-depth form -- its own instantiation: the others have no register for it)
    const int H = s2d ? a.hm : a.hin, W = s2d ? a.wm : a.win, tw = a.wino_tw, tpi = a.wino_th * tw;
    const int Mt = a.n * tpi;
    const int ncb = a.cout >> 6;
    // PERSISTENT blocks (one per CU: 128 KB of LDS, 512 registers per lane): block b runs on XCD b % 8; every XCD owns a contiguous
    // range of the (tile block, column block) items, column block fastest, and its blocks walk it with the stride of their count --
    // at any time the CUs of an XCD work on neighbouring tile blocks x all column blocks (shared input lines in the XCD's L2).
    // A new work-group per item cost ~10 us of launch, cold-start latency and drain per 27-us item at 128 input channels.
    int item, item_end, stride;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7;
        const int q = total >> 3, r = total & 7;
        const int start = xcd * q + (xcd < r ? xcd : r);
        item_end = start + q + (xcd < r ? 1 : 0);
        stride = (nblk - xcd + 7) >> 3;
        item = start + (bid >> 3);
    }
    if (item >= item_end) return;
    const int c0 = a.c0;
    const int nch = s2d ? (c0 >> 1) : ((c0 + (a.in_mode == IN_CAT ? a.c1 : 0)) >> 3);      // (even: checked at launch; s2d: 4 phases x c0 / 8)
    const int nch0 = s2d ? nch : (c0 >> 3);
    const unsigned in_bytes = (unsigned)a.n * (unsigned)a.hin * (unsigned)a.win * (unsigned)c0 * 4u;
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wgt_wino, (unsigned)a.cout * (unsigned)(nch * 8) * 64u);
    const unsigned wdt_mul = a.wdiv_t_mul, wdt_sh = a.wdiv_t_sh, wdw_mul = a.wdiv_tw_mul, wdw_sh = a.wdiv_tw_sh;

    // ---- input-transform role: thread = (tile tt of the block's 64, channel pair cp of the chunk's 8 channels)
    const int tt = tid >> 2, cp = tid & 3;
    unsigned poff[16];
    // (s2d: grid pixel (y, x) is the 2x2 block at (2y, 2x) of the real image; a chunk's phase (py, px) moves the descriptor's base)
    constexpr int psc = S2D ? 2 : 1;
    const unsigned pixb = (unsigned)(c0 * psc) * 4u, rowb = (unsigned)(a.win * psc) * (unsigned)c0 * 4u;
    auto set_patch_offsets = [&](int mt) {
        const int m = mt * 64 + tt;
        const bool tvalid = m < Mt;
        const int mm = tvalid ? m : 0;
        const int img = fdiv(mm, wdt_mul, wdt_sh), rem = mm - img * tpi;
        const int ty = fdiv(rem, wdw_mul, wdw_sh), tx = rem - ty * tw;
        const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
        // (one multiply chain, then adds: with a product per pixel hipcc guards each of the 16 offsets with its own branch)
        const unsigned base = (unsigned)((img * a.hin + psc * y0) * a.win + psc * x0) * (unsigned)c0 * 4u + (unsigned)(cp * 8);      // (wraps for y0 / x0 = -1: never used then)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ok = tvalid && (unsigned)(y0 + r) < (unsigned)H && (unsigned)(x0 + c) < (unsigned)W;
                const unsigned o = base + (unsigned)r * rowb + (unsigned)c * pixb;
                poff[r * 4 + c] = ok ? o : OOB_OFFSET;
            }
    };
    // V image: [pos][wt][kp][tile 32][4 k] floats; this thread's 8-B slot at pos 0
    const int vslot = (((tt >> 5) * 2 + (cp >> 1)) * 32 + (tt & 31)) * 4 + (cp & 1) * 2;

    f2 S[16];      // the staged 4x4 patch of one chunk (two channels), row-major
    f2 T[16];      // its column pass
    const float* const in0p = a.in0; const float* const in1p = a.in1 ? a.in1 : a.in0;      // (kept in SGPRs: no scalar load per chunk)
    auto patch_rsrc = [&](int c) {
        if constexpr (S2D) {      // chunk c = phase c >> (s2d - 1) (py * 2 + px), channels 8 (c & (c0 / 8 - 1)) ...
            const int ph = c >> (s2d - 1), cc = c & ((1 << (s2d - 1)) - 1);
            return make_rsrc(in0p + (size_t)(((ph >> 1) * a.win + (ph & 1)) * c0 + cc * 8), in_bytes);
        }
        const bool second = c >= nch0;
        const float* src = (second ? in1p : in0p) + (size_t)((second ? c - nch0 : c) * 8);
        return make_rsrc(src, in_bytes);
    };
    // B^T d B in two passes: rows of d (over r) per patch column c, then columns per transform row xi -> the V image
    // (each pass in two halves, so that a step can spread them over the gaps between its MFMAs)
    auto col_pass_a = [&](int c) { T[0 * 4 + c] = pk_sub(S[0 * 4 + c], S[2 * 4 + c]); T[1 * 4 + c] = pk_add(S[1 * 4 + c], S[2 * 4 + c]); };
    auto col_pass_b = [&](int c) { T[2 * 4 + c] = pk_sub(S[2 * 4 + c], S[1 * 4 + c]); T[3 * 4 + c] = pk_sub(S[1 * 4 + c], S[3 * 4 + c]); };
    auto col_pass = [&](int c) { col_pass_a(c); col_pass_b(c); };
    f2 R[4];       // one transform row on its way to LDS
    auto row_pass_a = [&](int xi) { R[0] = pk_sub(T[xi * 4 + 0], T[xi * 4 + 2]); R[1] = pk_add(T[xi * 4 + 1], T[xi * 4 + 2]); };
    auto row_pass_b = [&](int xi) { R[2] = pk_sub(T[xi * 4 + 2], T[xi * 4 + 1]); R[3] = pk_sub(T[xi * 4 + 1], T[xi * 4 + 3]); };
    auto row_store = [&](int xi, float4* vimg4) {
        float* vimg = (float*)vimg4 + vslot;
        *(f2*)(vimg + (xi * 4 + 0) * 512) = R[0];
        *(f2*)(vimg + (xi * 4 + 1) * 512) = R[1];
        *(f2*)(vimg + (xi * 4 + 2) * 512) = R[2];
        *(f2*)(vimg + (xi * 4 + 3) * 512) = R[3];
    };
    auto row_pass_store = [&](int xi, float4* vimg4) { row_pass_a(xi); row_pass_b(xi); row_store(xi, vimg4); };
    // the 32-KB weight image of (column block cbx, chunk c): 32 lane-linear 1-KB pieces, 8 per wave (piece j of this wave)
    const unsigned u_voff = (unsigned)(lane * 16 + wv * 8192);
    auto issue_u = [&](int cbx, int c, int j, float4* ubuf) {
        const unsigned soff = (unsigned)((cbx * nch + c) * 32768);
        lds_ptr_t dst = (lds_ptr_t)&ubuf[(wv * 8 + j) * 64];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, u_voff + (unsigned)(j * 1024), soff, 0, 0);
    };

    f32x16 acc[16];
    // item -> (column block, tile block).  Items are ordered so that 32 consecutive ones -- what the 32 CUs of an XCD hold at a time --
    // are 8 tile blocks x 4 column blocks (EVR_WINO_ORDER=0 / ncb % 4 != 0: 2 x 16, column block fastest): a weight image is then
    // shared by 8 CUs and an input patch by 4, instead of 2 and 16 -- 2.6x fewer bytes from the Infinity Cache per launch
    const int ntb = (Mt + 63) >> 6;
    const bool order2d = a.wino_order != 0 && (ncb & 3) == 0;
    auto decode = [&](int it, int& cbo, int& mto) {
        if (!order2d) { cbo = it % ncb; mto = it / ncb; return; }
        const int per = 8 * ncb, grp = it / per, rem = it - grp * per;
        const int gt = (ntb - 8 * grp) < 8 ? (ntb - 8 * grp) : 8;
        const int cq = rem / (4 * gt), rem2 = rem - cq * 4 * gt;
        cbo = cq * 4 + (rem2 & 3); mto = grp * 8 + (rem2 >> 2);
    };
    int cb, mt;      // the item being computed
    decode(item, cb, mt);
    int cbn = cb, mtn = mt; bool has_next = false;      // the item after it (prefetched by the last chunk's step)

    // epilogue operands requested by the last chunk's step, a whole step before they are used.  Every epilogue access is a buffer
    // operation with a 32-bit byte offset: pixels that do not exist (ragged last tile block, the odd last row / column of the grid)
    // carry EPI_OOB -- loads return 0, stores are dropped by the range check -- so there is no branch and no 64-bit address
    const int hl = lane >> 5;
    unsigned eoff[4];  // byte offset of the lane's first channel at the tile's 4 output pixels (py * 2 + px)
    f4 eb[4];          // bias runs of the lane's 4 x 4 rows
    f4 ecp[4];         // ConvLSTM: cell state of the lane's 4 channels at those pixels
    const bool tconv = !LSTM && a.os == 2;      // (four sub-pixel phases: wino_eligible)
    const unsigned out_bytes = (unsigned)a.n * (unsigned)a.hout * (unsigned)a.wout * (unsigned)(LSTM ? a.hidden : a.cout_total) * 4u;
    unsigned bcol0 = 0;      // the lane's first GEMM column (its bias run; = its first output channel except for a transposed convolution)
    auto epi_prefetch = [&]() {
        const int me = mt * 64 + wt * 32 + (lane & 31);
        const bool evalid = me < Mt;
        const int mme = evalid ? me : 0;
        const int eimg = fdiv(mme, wdt_mul, wdt_sh), erem = mme - eimg * tpi;
        const int ety = fdiv(erem, wdw_mul, wdw_sh), etx = erem - ety * tw;
        const unsigned cpp = (unsigned)(LSTM ? a.hidden : a.cout_total) * 4u;
        unsigned chan0 = LSTM ? (unsigned)(cb * 16 + wc * 8 + 4 * hl) : (unsigned)(cb * 64 + wc * 32 + 4 * hl);
        bcol0 = chan0;
        // output pixel of the tile's input pixel (Y, X): (Y, X) itself, or (2 Y + py, 2 X + px) of the wave's phase (its 32 columns
        // are one phase: ConvTaps::inter, or phase-major groups of >= 32 columns)
        int osc = 1, ofy = 0, ofx = 0;
        if (tconv) {
            const int col0 = cb * 64 + wc * 32;
            const int g = a.tp.inter ? ((col0 >> 5) & 3) : col0 / a.tp.grp_cols;
            chan0 = (unsigned)((a.tp.inter ? (((col0 >> 7) << 5) + (col0 & 31)) : col0 - g * a.tp.grp_cols) + 4 * hl);
            osc = 2; ofy = a.tp.grp_ofy[g]; ofx = a.tp.grp_ofx[g];
        }
        const unsigned Wo = (unsigned)a.wout;
        const unsigned ebase = (unsigned)((eimg * a.hout + osc * 2 * ety + ofy) * a.wout + osc * 2 * etx + ofx) * cpp + chan0 * 4u;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool ok = evalid && 2 * ety + (p >> 1) < H && 2 * etx + (p & 1) < W;
            const unsigned o = ebase + (unsigned)((p >> 1) * osc) * Wo * cpp + (unsigned)((p & 1) * osc) * cpp;
            eoff[p] = ok ? o : EPI_OOB;
        }
        const __amdgpu_buffer_rsrc_t rsb = make_rsrc(a.bias, (unsigned)a.cout * 4u);
        if constexpr (LSTM) {
            const unsigned brow = ((chan0 >> 5) * 128u + (chan0 & 31u)) * 4u;
#pragma unroll
            for (int q = 0; q < 4; ++q) eb[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsb, brow + 128u * q, 0, 0));
            const __amdgpu_buffer_rsrc_t rss = make_rsrc(a.state, out_bytes);
#pragma unroll
            for (int p = 0; p < 4; ++p) ecp[p] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rss, eoff[p], 0, 0));
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if constexpr (!PRED) eb[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsb, bcol0 * 4u + 32u * q, 0, 0));
        }
    };

    // One K chunk: 64 MFMAs (16 positions x 4) on the images of buffer B, and -- spread over the positions so that it issues in
    // the MFMAs' shadow -- the next chunk's weight DMA (two pieces at each of positions 0-3), its input transform (column pass at
    // 0-3, row pass + V stores at 4-7) and the patch loads of the chunk after it (four at each of positions 4-7, right behind
    // the column pass that frees the staging registers: they have 12 positions ~ 3000 cycles to land).
    // MODE 2: both follow; 1: only the next chunk follows (no patch loads); 0: the item's last chunk -- it requests the epilogue's
    // operands and, when the block has a further item, that item's first weight image and patch (the buffers of parity 0 are free
    // from this step's barrier on), so that the next item's cold start hides under this item's epilogue.
    // FIRST: an item's first chunk.  Its weight image was requested BEFORE the chunk-0 patch this thread has already consumed
    // (loads complete in order), so only the block-wide barrier is needed; the explicit count of the other steps would be wrong
    // behind an epilogue's stores.
    auto step = [&](auto bufc, auto modec, auto firstc, int c) {
        constexpr int B = decltype(bufc)::value, MODE = decltype(modec)::value;
        constexpr bool FIRST = decltype(firstc)::value != 0;
        float4* ucur = B ? ldsU1 : ldsU0; float4* unext = B ? ldsU0 : ldsU1;
        float4* vcur = B ? ldsV1 : ldsV0; float4* vnext = B ? ldsV0 : ldsV1;
        // U[c] (LDS-DMA) must have landed; the patch loads requested behind its last piece (memory operations complete in order) may
        // stay in flight -- the step before the last one requests none
        if constexpr (FIRST) asm volatile("" ::: "memory");
        else if constexpr (MODE == 0 || (VAR & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // V[c] and U[c] are whole; every wave has left the other buffers
        asm volatile("" ::: "memory");
        const float4* lu = ucur + wc * 64 + lane;
        const float4* lv = vcur + wt * 64 + lane;
        __amdgpu_buffer_rsrc_t rsp = rsw;
        if constexpr (MODE == 2) rsp = patch_rsrc((VAR & 4) ? 0 : c + 2);
        if constexpr (MODE == 0) rsp = patch_rsrc(0);
        constexpr bool SIDE = !(VAR & 2);
        // Side work of (position p, gap g): gap g is the shadow of the position's MFMA g (64 cycles of the matrix pipe; a wave issues
        // in order, so what sits between two MFMAs runs beside the first and delays the second once it is longer than that -- with
        // the same work bunched behind the first two MFMAs of positions 0-7 a step took 4930 instead of 4096 cycles).
        //   chunk c + 1: weight DMA pieces 2p, 2p + 1 at (p < 4, gaps 1 and 3); column pass of patch column p - 4 at (4 <= p < 8, gaps 2
        //   and 3); row pass of transform row p - 12 at (p >= 12, gaps 1 and 2) and its V stores at gap 3
        //   chunk c + 2: patch load (p - 8) * 4 + g at (8 <= p < 12, gap g) -- behind every DMA piece, so the wait at the top of the
        //   next step can leave all 16 in flight; they are read 12 positions (~3000 cycles) later
        //   last chunk of an item: epilogue operands at (0, 1), the next item's offsets at (1, 1), its first weight image at
        //   (2 <= p < 10, gap 1), its first patch at (10 <= p < 14, gap g)
        auto side = [&](int p, int g) {
            if constexpr (!SIDE) { if constexpr (MODE == 0) { if (p == 0 && g == 1) epi_prefetch(); } return; }
            if constexpr (MODE >= 1) {
                if constexpr (!(VAR & 8)) { if (p < 4 && (g & 1)) issue_u(cb, (VAR & 4) ? 0 : c + 1, 2 * p + (g >> 1), unext); }
                if constexpr (!(VAR & 32)) {
                    if (p >= 4 && p < 8) { if (g == 2) col_pass_a(p - 4); if (g == 3) col_pass_b(p - 4); }
                    if (p >= 12) { if (g == 1) row_pass_a(p - 12); if (g == 2) row_pass_b(p - 12); if (g == 3) row_store(p - 12, vnext); }
                }
            }
            if constexpr (MODE == 2 && !(VAR & 16)) {
                if (p >= 8 && p < 12) S[(p - 8) * 4 + g] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rsp, poff[(p - 8) * 4 + g], 0, 0));
            }
            if constexpr (MODE == 0) {
                if (p == 0 && g == 1) epi_prefetch();
                if (p == 1 && g == 1 && has_next) set_patch_offsets(mtn);
                if (p >= 2 && p < 10 && g == 1 && has_next) issue_u(cbn, 0, p - 2, unext);
                if (p >= 10 && p < 14 && has_next) S[(p - 10) * 4 + g] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rsp, poff[(p - 10) * 4 + g], 0, 0));
            }
        };
        auto mma = [&](int p, float x, float y, bool first) {
            if (FIRST && first) {      // (the accumulators of a new item start at zero: the first MFMA of every position takes a zero C operand)
                f32x16 z;
#pragma unroll
                for (int j = 0; j < 16; ++j) z[j] = 0.f;
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, z, 0, 0, 0);
            } else acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[p], 0, 0, 0);
        };
        float4 u = lu[0], v = lv[0];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            float4 un = u, vn = v;
            mma(p, u.x, v.x, true);
            if (p < 15) { un = lu[(p + 1) * 128]; vn = lv[(p + 1) * 128]; }
            side(p, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(p, u.y, v.y, false); side(p, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(p, u.z, v.z, false); side(p, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(p, u.w, v.w, false); side(p, 3);
            __builtin_amdgcn_sched_barrier(0);
            u = un; v = vn;
        }
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2;

    // A^T m A of accumulator register j: y[py * 2 + px]
    auto otrans = [&](auto jc, float (&y)[4]) {
        constexpr int j = decltype(jc)::value;
        float s0[4], s1[4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            // (read where they are used: as plain values hipcc copies all 256 accumulator registers out of the AGPRs right behind
            // each position's last MFMA of the last step -- 256 live VGPRs, 388 B of scratch per lane)
            float m0, m1, m2, m3;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m0) : "a"(acc[xi * 4 + 0][j]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m1) : "a"(acc[xi * 4 + 1][j]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m2) : "a"(acc[xi * 4 + 2][j]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m3) : "a"(acc[xi * 4 + 3][j]));
            s0[xi] = (m0 + m1) + m2;
            s1[xi] = (m1 - m2) - m3;
        }
        y[0] = (s0[0] + s0[1]) + s0[2];
        y[2] = (s0[1] - s0[2]) - s0[3];
        y[1] = (s1[0] + s1[1]) + s1[2];
        y[3] = (s1[1] - s1[2]) - s1[3];
        __builtin_amdgcn_sched_barrier(0);      // (one register's 16 reads and 24 adds at a time: otherwise the adds sink and 128 values are live)
    };

    // prologue of the block's first item: U[0] requested, patch 0 transformed into V0, patch 1 staged
    set_patch_offsets(mt);
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_u(cb, 0, j, ldsU0);
    {
        const __amdgpu_buffer_rsrc_t rs = patch_rsrc(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) S[i] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs, poff[i], 0, 0));
    }
    for (;;) {
        // (re-entered per item: S holds the item's chunk-0 patch -- requested by the prologue above or by the previous item's last step)
#pragma unroll
        for (int c = 0; c < 4; ++c) col_pass(c);
        {
            const __amdgpu_buffer_rsrc_t rs1 = patch_rsrc(1);
#pragma unroll
            for (int i = 0; i < 16; ++i) S[i] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs1, poff[i], 0, 0));
        }
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) row_pass_store(xi, ldsV0);
        {
            const int nxt = item + stride;
            has_next = nxt < item_end;
            cbn = cb; mtn = mt;
            if (has_next) decode(nxt, cbn, mtn);
        }
        step(I0{}, I2{}, I1{}, 0);
        step(I1{}, I2{}, I0{}, 1);
        int c = 2;
        for (; c + 3 < nch; c += 2) { step(I0{}, I2{}, I0{}, c); step(I1{}, I2{}, I0{}, c + 1); }
        if (c + 2 < nch) { step(I0{}, I2{}, I0{}, c); step(I1{}, I2{}, I0{}, c + 1); c += 2; }      // (not taken: nch - 2 is even)
        step(I0{}, I1{}, I0{}, c);
        step(I1{}, I0{}, I0{}, c + 1);

        // ---- epilogue: lane = tile (lane & 31), register j of an accumulator = row 8*(j>>2) + 4*(lane>>5) + (j&3) of the wave's 32
        if constexpr (LSTM) {
            // the wave's 32 rows = 4 gates (in, remember, out, cell: submodules.py:231) x 8 hidden channels; this lane: 4 channels,
            // finished two at a time (8-byte stores: the 4 x 4 transpose of whole 16-byte runs costs 16 registers this loop does not have)
            const __amdgpu_buffer_rsrc_t rss = make_rsrc(a.state, out_bytes), rso = make_rsrc(a.out, out_bytes);
            auto chan2 = [&](auto ic) {
                constexpr int i0 = decltype(ic)::value;
                f2 cn[4], hn[4];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    float yi[4], yf[4], yo[4], yc[4];
                    if (ii == 0) {
                        otrans(std::integral_constant<int, 0 + i0>{}, yi); otrans(std::integral_constant<int, 4 + i0>{}, yf);
                        otrans(std::integral_constant<int, 8 + i0>{}, yo); otrans(std::integral_constant<int, 12 + i0>{}, yc);
                    } else {
                        otrans(std::integral_constant<int, 1 + i0>{}, yi); otrans(std::integral_constant<int, 5 + i0>{}, yf);
                        otrans(std::integral_constant<int, 9 + i0>{}, yo); otrans(std::integral_constant<int, 13 + i0>{}, yc);
                    }
                    const int i = i0 + ii;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float gi = sigmoid_t<FAST>(yi[p] + eb[0][i]);
                        const float gf = sigmoid_t<FAST>(yf[p] + eb[1][i]);
                        const float go = sigmoid_t<FAST>(yo[p] + eb[2][i]);
                        const float gc = tanh_t<FAST>(yc[p] + eb[3][i]);
                        const float cv = __fadd_rn(__fmul_rn(gf, ecp[p][i]), __fmul_rn(gi, gc));      // submodules.py:242
                        cn[p][ii] = cv;
                        hn[p][ii] = go * tanh_t<FAST>(cv);                                          // submodules.py:243
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, cn[p]), rss, eoff[p] + 4u * i0, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, hn[p]), rso, eoff[p] + 4u * i0, 0, 0);
                }
            };
            // (one channel pair at a time: left alone, hipcc hoists all 256 accumulator reads and spills)
            chan2(I0{}); __builtin_amdgcn_sched_barrier(0);
            chan2(I2{}); __builtin_amdgcn_sched_barrier(0);
        } else {
            const int epi = a.epi;
            const bool res = (epi == EPI_RESIDUAL_RELU), relu = (epi != EPI_BIAS);
            const bool padd = a.post_add != nullptr;
            const __amdgpu_buffer_rsrc_t rso = make_rsrc(a.out, out_bytes);
            const __amdgpu_buffer_rsrc_t rsr = make_rsrc(res ? a.residual : a.out, res ? out_bytes : 0u);
            const __amdgpu_buffer_rsrc_t rsp = make_rsrc(padd ? a.post_add : a.out, padd ? out_bytes : 0u);      // (0 records: every load returns 0)
            auto group = [&](auto qc) {
                constexpr int q = decltype(qc)::value;
                f4 rv[4], sv[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    rv[p] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsr, eoff[p] + 32u * q, 0, 0));
                    sv[p] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsp, eoff[p] + 32u * q, 0, 0));
                }
                float y0[4], y1[4], y2[4], y3[4];
                otrans(std::integral_constant<int, 4 * q + 0>{}, y0);
                otrans(std::integral_constant<int, 4 * q + 1>{}, y1);
                otrans(std::integral_constant<int, 4 * q + 2>{}, y2);
                otrans(std::integral_constant<int, 4 * q + 3>{}, y3);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f4 v = {y0[p], y1[p], y2[p], y3[p]};
                    v += eb[q];
                    v += rv[p];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu ? fmaxf(v[i], 0.f) : v[i];
                    v += sv[p];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rso, eoff[p] + 32u * q, 0, 0);
                }
            };
            if constexpr (PRED) {
                // fused prediction layer (model/unet.py:136-138; conv.hip epi_finish's `pw` path): the wave's 32 rows are the 32 channels of
                // one sub-pixel phase, so an output pixel's dot product closes inside the lane pair -- nothing is stored but the image
                float part[4] = {0.f, 0.f, 0.f, 0.f};
                const __amdgpu_buffer_rsrc_t rspw = make_rsrc(a.pred_w, (unsigned)a.cout_total * 4u);
                // (the tile's first output pixel and the lane's first channel, decoded again here: six registers the main loop does not have)
                const int col0 = cb * 64 + wc * 32;
                const int pg = a.tp.inter ? ((col0 >> 5) & 3) : col0 / a.tp.grp_cols;
                const __amdgpu_buffer_rsrc_t rsbias = make_rsrc(a.bias, (unsigned)a.cout * 4u);
                const unsigned pchan0 = (unsigned)((a.tp.inter ? (((col0 >> 7) << 5) + (col0 & 31)) : col0 - pg * a.tp.grp_cols) + 4 * hl);
                auto pgroup = [&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    // (bias and prediction weights of the run fetched here, not a step ahead: 256 cache-hot bytes, and 16 registers the last step keeps)
                    const f4 pw = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rspw, pchan0 * 4u + 32u * q, 0, 0));
                    const f4 bq = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsbias, (unsigned)(col0 + 4 * hl) * 4u + 32u * q, 0, 0));
                    f4 sv[4];      // skip_sum fused into the producer (model_util.py:4-5): added after the activation, as in group()
#pragma unroll
                    for (int p = 0; p < 4; ++p) sv[p] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsp, eoff[p] + 32u * q, 0, 0));
                    float y0[4], y1[4], y2[4], y3[4];
                    otrans(std::integral_constant<int, 4 * q + 0>{}, y0);
                    otrans(std::integral_constant<int, 4 * q + 1>{}, y1);
                    otrans(std::integral_constant<int, 4 * q + 2>{}, y2);
                    otrans(std::integral_constant<int, 4 * q + 3>{}, y3);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        f4 v = {y0[p], y1[p], y2[p], y3[p]};
                        v += bq;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { const float t = (relu ? fmaxf(v[i], 0.f) : v[i]) + sv[p][i]; part[p] = fmaf(t, pw[i], part[p]); }
                    }
                };
                pgroup(I0{}); __builtin_amdgcn_sched_barrier(0);
                pgroup(I1{}); __builtin_amdgcn_sched_barrier(0);
                pgroup(I2{}); __builtin_amdgcn_sched_barrier(0);
                pgroup(std::integral_constant<int, 3>{}); __builtin_amdgcn_sched_barrier(0);
                // (buffer operations with 32-bit offsets, pixels that do not exist / the lane pair's upper half / pixels outside the crop carry
                // EPI_OOB: no branch, no 64-bit address per pixel -- this epilogue has no registers for them)
                const unsigned npix = (unsigned)a.n * (unsigned)a.hout * (unsigned)a.wout;
                const __amdgpu_buffer_rsrc_t rsd = make_rsrc(a.pred_skip_dot ? a.pred_skip_dot : a.pred_w, a.pred_skip_dot ? npix * 4u : 0u);
                const __amdgpu_buffer_rsrc_t rsv = make_rsrc(a.prev_rec ? a.prev_rec : img_out, a.prev_rec ? npix * 4u : 0u);
                const __amdgpu_buffer_rsrc_t rsi = make_rsrc(img_out, (unsigned)a.n * (unsigned)a.crop_h * (unsigned)a.crop_w * 4u);
                const int me = mt * 64 + wt * 32 + (lane & 31);
                const int pimg = fdiv(me < Mt ? me : 0, wdt_mul, wdt_sh), erem = (me < Mt ? me : 0) - pimg * tpi;
                const int ety = fdiv(erem, wdw_mul, wdw_sh), etx = erem - ety * tw;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float pp = part[p];
                    pp += __shfl_xor(pp, 32, 64);
                    const int oy = 2 * (2 * ety + (p >> 1)) + a.tp.grp_ofy[pg], ox = 2 * (2 * etx + (p & 1)) + a.tp.grp_ofx[pg];
                    const bool live = hl == 0 && eoff[p] != EPI_OOB;
                    const unsigned po = live ? (unsigned)((pimg * a.hout + oy) * a.wout + ox) * 4u : EPI_OOB;
                    float sres = pp + a.pred_b;
                    sres += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsd, po, 0, 0));
                    if (a.pred_sigmoid) sres = sigmoid_t<false>(sres);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sres), rsv, po, 0, 0);
                    const int y = oy - a.crop_y0, x = ox - a.crop_x0;
                    const bool in = live && (unsigned)y < (unsigned)a.crop_h && (unsigned)x < (unsigned)a.crop_w;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sres), rsi, in ? (unsigned)((pimg * a.crop_h + y) * a.crop_w + x) * 4u : EPI_OOB, 0, 0);
                }
            } else {
            group(I0{}); __builtin_amdgcn_sched_barrier(0);
            group(I1{}); __builtin_amdgcn_sched_barrier(0);
            group(I2{}); __builtin_amdgcn_sched_barrier(0);
            group(std::integral_constant<int, 3>{}); __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!has_next) break;
        item += stride; cb = cbn; mt = mtn;
    }
#endif
}

}  // namespace wino

int launch_conv_wino(const ConvArgs& a, const ConvArgs* d_args, hipStream_t stream, float* img) {
    EVR_REQUIRE(wino_eligible(a), "conv_wino: the plan is not a 3x3 stride-1 fp32 convolution this kernel covers");
    EVR_REQUIRE(!a.pred_w || img, "conv_wino: a fused prediction needs the image buffer");
    EVR_REQUIRE(a.wino_th == ((a.wino_s2d ? a.hm : a.hin) + 1) / 2 && a.wino_tw == ((a.wino_s2d ? a.wm : a.win) + 1) / 2 && a.wdiv_t_sh < 32 && a.wdiv_tw_sh < 32, "conv_wino: plan without tile grid");
    EVR_REQUIRE((int64_t)a.n * a.hin * a.win * a.c0 * 4 < 0xFFFFFF00LL, "conv_wino: input tensor exceeds the buffer-descriptor range");
    const int64_t Mt = (int64_t)a.n * a.wino_th * a.wino_tw;
    const int64_t total = ((Mt + 63) / 64) * (a.cout / 64);
    EVR_REQUIRE(total < (1LL << 31), "conv_wino: too many work items");
    // persistent blocks, one per CU (EVR_WINO_BLOCKS overrides the count); libm-grade gate activations with EVR_WINO_FASTACT=0
    static const int nblocks = getenv("EVR_WINO_BLOCKS") ? atoi(getenv("EVR_WINO_BLOCKS")) : 256;
    static const bool fast_act = getenv("EVR_WINO_FASTACT") ? atoi(getenv("EVR_WINO_FASTACT")) != 0 : true;
    const unsigned grid = (unsigned)(total < nblocks ? total : nblocks);
    static const int var = getenv("EVR_WINO_VAR") ? atoi(getenv("EVR_WINO_VAR")) : 0;
    if (a.epi == EPI_LSTM) {
        if (!fast_act) hipLaunchKernelGGL((wino::wino_f32_kernel<true, false>), dim3(grid), dim3(256), 0, stream, d_args, (int)total, img);
        else if (var == 2) hipLaunchKernelGGL((wino::wino_f32_kernel<true, true, 2>), dim3(grid), dim3(256), 0, stream, d_args, (int)total, img);
        else if (var == 4) hipLaunchKernelGGL((wino::wino_f32_kernel<true, true, 4>), dim3(grid), dim3(256), 0, stream, d_args, (int)total, img);
        else hipLaunchKernelGGL((wino::wino_f32_kernel<true, true>), dim3(grid), dim3(256), 0, stream, d_args, (int)total, img);
    } else if (a.wino_s2d) {
        hipLaunchKernelGGL((wino::wino_f32_kernel<false, false, 0, false, true>), dim3(grid), dim3(256), 0, stream, d_args, (int)total, img);
    } else if (a.pred_w) {
        hipLaunchKernelGGL((wino::wino_f32_kernel<false, false, 0, true>), dim3(grid), dim3(256), 0, stream, d_args, (int)total, img);
    } else {
        hipLaunchKernelGGL((wino::wino_f32_kernel<false, false>), dim3(grid), dim3(256), 0, stream, d_args, (int)total, img);
    }
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

}  // namespace evr

// ET-Net (model/eitr/ of the reference) -- the token-side kernels of its transformer encoders/decoders:
//
//   layernorm     nn.LayerNorm(256) of the pre-norm layers (transformer_encoder.py:66,71; transformer_decoder.py:68-80)
//   attention     nn.MultiheadAttention's core: softmax(q k^T / sqrt(d)) v per (sequence, head), d = 32, computed
//                 flash-style in fp32 (online softmax over 64-key tiles staged in LDS; one query per lane, so a
//                 row's max / sum / output never leave its registers -- no cross-lane reductions)
//   add_pos       words + sine position table (u_trans.py:93-104; TransformerEncoder.with_embed)
//   mean6         (hs0 + hs1 + hs2 + hc0 + hc1 + hc2) / 6 (u_trans.py:112)
// The projections / FFN layers are 1x1 convolutions on the matrix cores (conv.hip); tokens stay fp32 on the residual
// path and are written PACKED (conv.h) only where a matrix-core GEMM consumes them.
#include "conv.h"
#include "packed.h"
#include <cstdlib>

namespace evr {

// one wave per token row of C = 256 channels (4 per lane)
__global__ __launch_bounds__(256) void layernorm256_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                            float* __restrict__ out, int64_t rows, int out_packed) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4 v = *(const float4*)(x + row * 256 + lane * 4);
    float s = (v.x + v.y) + (v.z + v.w);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * (1.0f / 256.0f);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-5f);
    const float4 g = *(const float4*)(w + lane * 4), be = *(const float4*)(b + lane * 4);
    const float4 r = make_float4(dx * rstd * g.x + be.x, dy * rstd * g.y + be.y, dz * rstd * g.z + be.z, dw * rstd * g.w + be.w);
    st4_any(out + row * 256, lane * 4, r, out_packed);
}

// q: [n, Lq, ldq] at column offset qo + h*32, k / v likewise with Lk rows; out: [n, Lq, 256] (PLAIN or PACKED), column h*32.
// Block = 4 waves = 4 blocks of 64 queries of one (sequence, head); the K / V tiles are shared through LDS.
constexpr int AT_D = 32, AT_KT = 64;
__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float sk[AT_KT][AT_D];
    __shared__ __attribute__((aligned(16))) float sv[AT_KT][AT_D];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int h = blockIdx.y, n = blockIdx.z;
    const int qi = (blockIdx.x * 4 + wv) * 64 + lane;
    const bool qok = qi < a.Lq;
    const float scale = 0.17677669529663687f;      // 1 / sqrt(32)
    float q[AT_D], o[AT_D];
    {
        const float* qp = a.q + ((int64_t)n * a.Lq + (qok ? qi : 0)) * a.ldq + a.qo + h * AT_D;
#pragma unroll
        for (int d = 0; d < AT_D; d += 4) {
            const float4 t = *(const float4*)(qp + d);
            q[d] = t.x * scale; q[d + 1] = t.y * scale; q[d + 2] = t.z * scale; q[d + 3] = t.w * scale;
        }
    }
#pragma unroll
    for (int d = 0; d < AT_D; ++d) o[d] = 0.f;
    float m = -1e30f, l = 0.f;
    const float* kb = a.k + (int64_t)n * a.Lk * a.ldk + a.ko + h * AT_D;
    const float* vb = a.v + (int64_t)n * a.Lk * a.ldv + a.vo + h * AT_D;
    for (int k0 = 0; k0 < a.Lk; k0 += AT_KT) {
        __syncthreads();
        for (int i = tid; i < AT_KT * AT_D / 4; i += 256) {     // 512 float4 per matrix
            const int r = i / (AT_D / 4), c4 = (i % (AT_D / 4)) * 4;
            const int kr = k0 + r;
            float4 tk = make_float4(0.f, 0.f, 0.f, 0.f), tv = tk;
            if (kr < a.Lk) { tk = *(const float4*)(kb + (int64_t)kr * a.ldk + c4); tv = *(const float4*)(vb + (int64_t)kr * a.ldv + c4); }
            *(float4*)&sk[r][c4] = tk; *(float4*)&sv[r][c4] = tv;
        }
        __syncthreads();
        const int kn = min(AT_KT, a.Lk - k0);
        for (int j0 = 0; j0 < kn; j0 += 16) {
            float s[16];
            float mx = m;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int d = 0; d < AT_D; d += 4) {
                    const float4 kk = *(const float4*)&sk[j0 + j][d];      // wave-uniform address: LDS broadcast
                    acc = fmaf(q[d], kk.x, acc); acc = fmaf(q[d + 1], kk.y, acc); acc = fmaf(q[d + 2], kk.z, acc); acc = fmaf(q[d + 3], kk.w, acc);
                }
                s[j] = (j0 + j < kn) ? acc : -1e30f;
                mx = fmaxf(mx, s[j]);
            }
            const float alpha = __expf(m - mx);
            l *= alpha;
#pragma unroll
            for (int d = 0; d < AT_D; ++d) o[d] *= alpha;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float p = (j0 + j < kn) ? __expf(s[j] - mx) : 0.f;
                l += p;
#pragma unroll
                for (int d = 0; d < AT_D; d += 4) {
                    const float4 vv = *(const float4*)&sv[j0 + j][d];
                    o[d] = fmaf(p, vv.x, o[d]); o[d + 1] = fmaf(p, vv.y, o[d + 1]); o[d + 2] = fmaf(p, vv.z, o[d + 2]); o[d + 3] = fmaf(p, vv.w, o[d + 3]);
                }
            }
            m = mx;
        }
    }
    if (qok) {
        const float inv = 1.0f / l;
        float* op = a.out + ((int64_t)n * a.Lq + qi) * 256;
#pragma unroll
        for (int d = 0; d < AT_D; d += 4)
            st4_any(op, h * AT_D + d, make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv), a.out_packed);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same attention on the matrix cores (round 3; EVR_ATTN_VALU=1 keeps the kernel above).  Flash-style over 64-key tiles;
// a wave owns 32 queries of one (sequence, head).  Both GEMMs run TRANSPOSED so that a lane is a query (conv.hip's C^T trick):
//   S^T [key][query] = K . Q^T        A operand = K rows (keys), B operand = Q (queries): acc lane = query, registers = keys
//   O^T [d][query]  += V^T . P^T      A operand = V^T rows (d),   B operand = P (queries): acc lane = query, registers = d
// so a query's running max / sum / rescale are per-lane scalars (its 32 keys of a block live in the lane pair r, r + 32: one
// shuffle for the max), and the probabilities go from the S accumulator straight into the B operand of the second GEMM -- the
// k order inside an MFMA is free as long as both operands agree, so V^T is staged in LDS with its keys in the accumulator's
// order (registers 8s .. 8s+7 of half h = keys 16s + 4h + {0..3}, 16s + 8 + 4h + {0..3} of the 32-key block).
// Arithmetic: every factor (q / sqrt(d), k, v, p) is hi + lo in two IEEE halves and every product three f16 MFMAs
// (hi hi + hi lo + lo hi, fp32 accumulate) -- 22 bits per factor, as in the fp32-grade convolution mode; exp in fp32.
typedef _Float16 at_h8 __attribute__((ext_vector_type(8)));
typedef float at_f16 __attribute__((ext_vector_type(16)));
typedef unsigned at_u4 __attribute__((ext_vector_type(4)));
typedef _Float16 at_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void at_split8(const float (&v)[8], at_u4& hi, at_u4& lo) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const at_h2 a = {(_Float16)v[2 * k], (_Float16)v[2 * k + 1]};
        const at_h2 b = {(_Float16)(v[2 * k] - (float)a[0]), (_Float16)(v[2 * k + 1] - (float)a[1])};
        hi[k] = __builtin_bit_cast(unsigned, a); lo[k] = __builtin_bit_cast(unsigned, b);
    }
}
__device__ __forceinline__ at_f16 at_mma3(at_f16 acc, at_u4 ah, at_u4 al, at_u4 bh, at_u4 bl) {
#if defined(__HIP_DEVICE_COMPILE__)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(at_h8, al), __builtin_bit_cast(at_h8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(at_h8, ah), __builtin_bit_cast(at_h8, bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(at_h8, ah), __builtin_bit_cast(at_h8, bh), acc, 0, 0, 0);
#endif
    return acc;
}
__global__ __launch_bounds__(256) void attention_mfma_kernel(const AttnArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    // K tile [key][32 d] and V^T tile [d][64 keys in accumulator order], each as a hi and a lo plane of halves (4 KB per plane)
    __shared__ __attribute__((aligned(16))) _Float16 sKh[AT_KT * AT_D], sKl[AT_KT * AT_D], sVh[AT_D * AT_KT], sVl[AT_D * AT_KT];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, n = blockIdx.z;
    const int qi = (blockIdx.x * 4 + wv) * 32 + r;
    const bool qok = qi < a.Lq;
    const float scale = 0.17677669529663687f;      // 1 / sqrt(32)
    // Q fragments (B operand of S^T): lane (query r, half hh) supplies d = 16 s + 8 hh .. + 7 of slab s
    at_u4 qh[2], ql[2];
    {
        const float* qp = a.q + ((int64_t)n * a.Lq + (qok ? qi : 0)) * a.ldq + a.qo + h * AT_D;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float4 t0 = *(const float4*)(qp + 16 * s + 8 * hh), t1 = *(const float4*)(qp + 16 * s + 8 * hh + 4);
            const float v[8] = {t0.x * scale, t0.y * scale, t0.z * scale, t0.w * scale, t1.x * scale, t1.y * scale, t1.z * scale, t1.w * scale};
            at_split8(v, qh[s], ql[s]);
        }
    }
    at_f16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    float m = -1e30f, l = 0.f;                       // (l: this lane's half of the row sum; the pair is added at the end)
    const float* kb = a.k + (int64_t)n * a.Lk * a.ldk + a.ko + h * AT_D;
    const float* vb = a.v + (int64_t)n * a.Lk * a.ldv + a.vo + h * AT_D;
    // staging: thread -> key tid / 4 of the tile, d segment (tid % 4) * 8
    const int skey = tid >> 2, sd = (tid & 3) * 8;
    const int kk = skey & 31;                        // position of the key inside its 32-key block -> its slot in accumulator order
    const int vslot = (skey >> 5) * 32 + ((kk >> 4) * 16) + (((kk >> 2) & 1) * 8) + (((kk >> 3) & 1) * 4) + (kk & 3);
    for (int k0 = 0; k0 < a.Lk; k0 += AT_KT) {
        __syncthreads();
        {
            const int kr = k0 + skey;
            float kv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, vv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (kr < a.Lk) {
                const float4 k0v = *(const float4*)(kb + (int64_t)kr * a.ldk + sd), k1v = *(const float4*)(kb + (int64_t)kr * a.ldk + sd + 4);
                const float4 v0v = *(const float4*)(vb + (int64_t)kr * a.ldv + sd), v1v = *(const float4*)(vb + (int64_t)kr * a.ldv + sd + 4);
                kv[0] = k0v.x; kv[1] = k0v.y; kv[2] = k0v.z; kv[3] = k0v.w; kv[4] = k1v.x; kv[5] = k1v.y; kv[6] = k1v.z; kv[7] = k1v.w;
                vv[0] = v0v.x; vv[1] = v0v.y; vv[2] = v0v.z; vv[3] = v0v.w; vv[4] = v1v.x; vv[5] = v1v.y; vv[6] = v1v.z; vv[7] = v1v.w;
            }
            at_u4 khi, klo;
            at_split8(kv, khi, klo);
            *(at_u4*)&sKh[skey * AT_D + sd] = khi; *(at_u4*)&sKl[skey * AT_D + sd] = klo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 vh = (_Float16)vv[e];
                sVh[(sd + e) * AT_KT + vslot] = vh;
                sVl[(sd + e) * AT_KT + vslot] = (_Float16)(vv[e] - (float)vh);
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 2; ++b) {                 // the tile's two 32-key blocks
            if (k0 + 32 * b >= a.Lk) break;           // (block-uniform)
            at_f16 sacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {             // A = K rows: lane (key r, half hh) supplies d = 16 s + 8 hh .. + 7
                const at_u4 ah = *(const at_u4*)&sKh[(32 * b + r) * AT_D + 16 * s + 8 * hh];
                const at_u4 al = *(const at_u4*)&sKl[(32 * b + r) * AT_D + 16 * s + 8 * hh];
                sacc = at_mma3(sacc, ah, al, qh[s], ql[s]);
            }
            // register i of half hh = key 8 (i >> 2) + 4 hh + (i & 3) of the block
            float mx = m;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = k0 + 32 * b + 8 * (i >> 2) + 4 * hh + (i & 3);
                sacc[i] = (key < a.Lk) ? sacc[i] : -1e30f;
                mx = fmaxf(mx, sacc[i]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));   // the other half of the query's keys
            const float alpha = __expf(m - mx);
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] *= alpha;
            float p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = k0 + 32 * b + 8 * (i >> 2) + 4 * hh + (i & 3);
                p[i] = (key < a.Lk) ? __expf(sacc[i] - mx) : 0.f;
                l += p[i];
            }
            m = mx;
#pragma unroll
            for (int s = 0; s < 2; ++s) {             // B = P: registers 8 s .. 8 s + 7 are this lane's 8 k-values of slab s
                const float pv[8] = {p[8 * s], p[8 * s + 1], p[8 * s + 2], p[8 * s + 3], p[8 * s + 4], p[8 * s + 5], p[8 * s + 6], p[8 * s + 7]};
                at_u4 ph, pl;
                at_split8(pv, ph, pl);
                const at_u4 vh = *(const at_u4*)&sVh[r * AT_KT + 32 * b + 16 * s + 8 * hh];      // A = V^T rows: lane (d r, half hh)
                const at_u4 vl = *(const at_u4*)&sVl[r * AT_KT + 32 * b + 16 * s + 8 * hh];
                o = at_mma3(o, vh, vl, ph, pl);
            }
        }
    }
    l += __shfl_xor(l, 32, 64);
    if (qok) {
        const float inv = 1.0f / l;
        float* op = a.out + ((int64_t)n * a.Lq + qi) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q)                   // register 4 q + j = d 8 q + 4 hh + j
            st4_any(op, h * AT_D + 8 * q + 4 * hh, make_float4(o[4 * q] * inv, o[4 * q + 1] * inv, o[4 * q + 2] * inv, o[4 * q + 3] * inv), a.out_packed);
    }
#endif
}

__global__ __launch_bounds__(256) void add_pos_kernel(const float* __restrict__ x, const float* __restrict__ pos, float* __restrict__ out,
                                                       int64_t total4, int L, int x_packed) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one 4-channel run
    if (i >= total4) return;
    const int c4 = (int)(i % 64) * 4;
    const int64_t row = i / 64;                                     // n*L + l
    const int l = (int)(row % L);
    const float4 v = ld4_any(x + row * 256, c4, x_packed), p = *(const float4*)(pos + (int64_t)l * 256 + c4);
    *(float4*)(out + row * 256 + c4) = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
}

__global__ __launch_bounds__(256) void mean6_kernel(const float* __restrict__ a0, const float* __restrict__ a1, const float* __restrict__ a2,
                                                     const float* __restrict__ a3, const float* __restrict__ a4, const float* __restrict__ a5,
                                                     float* __restrict__ out, int64_t total4, int out_packed) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i % 64) * 4;
    const int64_t o = (i / 64) * 256;
    const float4 x0 = *(const float4*)(a0 + o + c4), x1 = *(const float4*)(a1 + o + c4), x2 = *(const float4*)(a2 + o + c4);
    const float4 x3 = *(const float4*)(a3 + o + c4), x4 = *(const float4*)(a4 + o + c4), x5 = *(const float4*)(a5 + o + c4);
    auto f = [](float p, float q, float r, float s, float t, float u) { return (((((p + q) + r) + s) + t) + u) / 6.0f; };
    st4_any(out + o, c4, make_float4(f(x0.x, x1.x, x2.x, x3.x, x4.x, x5.x), f(x0.y, x1.y, x2.y, x3.y, x4.y, x5.y),
                                     f(x0.z, x1.z, x2.z, x3.z, x4.z, x5.z), f(x0.w, x1.w, x2.w, x3.w, x4.w, x5.w)), out_packed);
}

int launch_layernorm256(const float* x, const float* w, const float* b, float* out, int64_t rows, int out_packed, hipStream_t stream) {
    hipLaunchKernelGGL(layernorm256_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, w, b, out, rows, out_packed);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}
int launch_attention(const AttnArgs& a, hipStream_t stream) {
    EVR_REQUIRE(a.heads * AT_D == 256 && a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0, "attention: 8 heads of 32 channels, 16-B aligned rows");
    static const bool valu = getenv("EVR_ATTN_VALU") != nullptr && atoi(getenv("EVR_ATTN_VALU")) != 0;
    if (valu) hipLaunchKernelGGL(attention_kernel, dim3((unsigned)((a.Lq + 255) / 256), a.heads, a.n), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attention_mfma_kernel, dim3((unsigned)((a.Lq + 127) / 128), a.heads, a.n), dim3(256), 0, stream, a);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}
int launch_add_pos(const float* x, const float* pos, float* out, int n, int L, int x_packed, hipStream_t stream) {
    const int64_t total4 = (int64_t)n * L * 64;
    hipLaunchKernelGGL(add_pos_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, x, pos, out, total4, L, x_packed);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}
int launch_mean6(const float* const* in, float* out, int64_t rows, int out_packed, hipStream_t stream) {
    const int64_t total4 = rows * 64;
    hipLaunchKernelGGL(mean6_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, in[0], in[1], in[2], in[3], in[4], in[5], out, total4, out_packed);
    EVR_LAUNCH_CHECK();
    return EVR_OK;
}

}  // namespace evr

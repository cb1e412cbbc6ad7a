// fa3_fwd_f32_kernel.h -- EXACT fp32 forward for fp32 modules (the reference's default dtype, flash_attention_3.py:19-27; BASELINE
// config C1 is one): every product and sum in fp32 on the vector ALUs, nothing is rounded to 16 bits.  It replaces the same seam as
// the MFMA kernels (flash_attention_3.py:120-262) for callers that hand over fp32 operands and want the reference's fp32 numbers
// (<= 1e-5 of the oracle, not the ~1e-2 of bf16 operands).  Throughput is that of fp32 FMAs with a plain tiling (a few TFLOP/s),
// two orders of magnitude below the bf16 path: it is the accuracy mode, the module keeps "bf16" as an explicit option.
//
// Geometry: a workgroup of 256 threads owns 64 query rows of one (batch, head) and walks the keys in tiles of 64.  LDS holds Q^T, K^T
// ([d][row], padded: the S = Q K^T micro-tiles read float4s along the rows) and V ([key][d]) and the 64 x 64 tile of P.
// Thread (ty = tid / 16, tx = tid % 16) computes S rows 4 ty .. +3 x keys 4 tx .. +3, and O rows 4 ty .. +3 x columns tx + 16 i.
// Online softmax exactly as flash_attention_3.py:239-250 (un-normalised until the end), row reductions across the 16 lanes of a row
// group by DPP shuffles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfa {

struct F32Params {
    const float* q;
    const float* k;
    const float* v;
    float* o;
    float* lse;
    const int32_t* seqlens_k;
    const uint8_t* mask;           // optional u8 mask, 0 = masked, byte strides below (a [B,Sk] key mask is (stride, 0, 0, 1))
    int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss;      // element strides
    int64_t m_sb, m_sh, m_sq, m_sk;
    int32_t B, H, Sq, Sk, kv_group, causal;
    float scale;
    // attention dropout of the reference's dense branch (flash_attention_3.py:174-175: dropout(softmax(scores)) @ v): keep-mask bytes
    // [B][H][Sq][Sk] contiguous (non-zero = keep), kept weights scaled by drop_scale = 1 / (1 - p); the row sum stays un-dropped
    const uint8_t* drop_mask;
    float drop_scale;
};

template <int D>
__global__ __launch_bounds__(256) void fa3_fwd_f32_kernel(const F32Params p) {
    constexpr int BM = 64, BN = 64, LD = BM + 4;          // +4: rows of the transposed images start on different banks
    constexpr int NC = D / 16;                            // O columns per thread
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Qt = smem_f;                                   // [D][LD]
    float* Kt = Qt + D * LD;                              // [D][LD]
    float* Vs = Kt + D * LD;                              // [BN][D]
    float* Ps = Vs + BN * D;                              // [BM][BN + 4]
    constexpr int LP = BN + 4;

    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int nqb = (p.Sq + BM - 1) / BM;
    const int bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
    const int b = bh / p.H, h = bh - b * p.H, hkv = h / p.kv_group;
    const int q0 = qb * BM;
    const float* qp = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const float* kp = p.k + (int64_t)b * p.k_sb + (int64_t)hkv * p.k_sh;
    const float* vp = p.v + (int64_t)b * p.v_sb + (int64_t)hkv * p.v_sh;
    int kv_len = p.Sk;
    if (p.seqlens_k) kv_len = min(kv_len, max(p.seqlens_k[b], 0));
    const int kv_end = p.causal ? min(kv_len, q0 + BM) : kv_len;

    // Q block, transposed and pre-scaled by nothing (the scale multiplies the fp32 score, as the reference does with q: :138)
    for (int i = tid; i < BM * (D / 4); i += 256) {
        const int r = i / (D / 4), c4 = i - r * (D / 4);
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + r < p.Sq) x = *(const float4*)(qp + (int64_t)(q0 + r) * p.q_ss + 4 * c4);
        Qt[(4 * c4 + 0) * LD + r] = x.x; Qt[(4 * c4 + 1) * LD + r] = x.y; Qt[(4 * c4 + 2) * LD + r] = x.z; Qt[(4 * c4 + 3) * LD + r] = x.w;
    }
    float o[4][NC];
    float m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m[i] = -INFINITY; l[i] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) o[i][c] = 0.f;
    }
    for (int k0 = 0; k0 < kv_end; k0 += BN) {
        __syncthreads();                                   // the previous tile's K^T / V / P are no longer read
        for (int i = tid; i < BN * (D / 4); i += 256) {
            const int r = i / (D / 4), c4 = i - r * (D / 4);
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
            if (k0 + r < p.Sk) {
                x = *(const float4*)(kp + (int64_t)(k0 + r) * p.k_ss + 4 * c4);
                y = *(const float4*)(vp + (int64_t)(k0 + r) * p.v_ss + 4 * c4);
            }
            Kt[(4 * c4 + 0) * LD + r] = x.x; Kt[(4 * c4 + 1) * LD + r] = x.y; Kt[(4 * c4 + 2) * LD + r] = x.z; Kt[(4 * c4 + 3) * LD + r] = x.w;
            *(float4*)(Vs + r * D + 4 * c4) = y;
        }
        __syncthreads();
        // S micro-tile: rows 4 ty .. +3, keys 4 tx .. +3
        float s[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[i][j] = 0.f;
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
            const float4 a = *(const float4*)(Qt + d * LD + 4 * ty);
            const float4 c = *(const float4*)(Kt + d * LD + 4 * tx);
            const float av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[i][j] = __builtin_fmaf(av[i], cv[j], s[i][j]);
        }
        // scale, mask, online softmax per row (16 lanes tx share a row group)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qi = q0 + 4 * ty + i;
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kj = k0 + 4 * tx + j;
                bool vis = kj < kv_len && (!p.causal || kj <= qi) && qi < p.Sq;
                if (vis && p.mask)
                    vis = p.mask[(int64_t)b * p.m_sb + (int64_t)h * p.m_sh + (int64_t)qi * p.m_sq + (int64_t)kj * p.m_sk] != 0;
                s[i][j] = vis ? s[i][j] * p.scale : -INFINITY;
                mx = fmaxf(mx, s[i][j]);
            }
#pragma unroll
            for (int w = 1; w < 16; w <<= 1) mx = fmaxf(mx, __shfl_xor(mx, w, 16));
            const float m_new = fmaxf(m[i], mx);
            const float alpha = m_new == -INFINITY ? 1.f : __expf(m[i] - m_new);       // (a row with no visible key so far keeps l = 0, O = 0)
            float rs = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = m_new == -INFINITY ? 0.f : __expf(s[i][j] - m_new);
                rs += e;
                float ed = e;
                if (p.drop_mask) {
                    const int kj = k0 + 4 * tx + j;
                    const bool keep = qi < p.Sq && kj < p.Sk && p.drop_mask[(((int64_t)b * p.H + h) * p.Sq + qi) * p.Sk + kj] != 0;
                    ed = keep ? e * p.drop_scale : 0.f;
                }
                Ps[(4 * ty + i) * LP + 4 * tx + j] = ed;
            }
#pragma unroll
            for (int w = 1; w < 16; w <<= 1) rs += __shfl_xor(rs, w, 16);
            l[i] = l[i] * alpha + rs;
            m[i] = m_new;
#pragma unroll
            for (int c = 0; c < NC; ++c) o[i][c] *= alpha;
        }
        __syncthreads();
        // O rows 4 ty .. +3, columns tx + 16 c  +=  P (64 keys) x V
#pragma unroll 4
        for (int kk = 0; kk < BN; ++kk) {
            float pv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[i] = Ps[(4 * ty + i) * LP + kk];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float vv = Vs[kk * D + tx + 16 * c];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i][c] = __builtin_fmaf(pv[i], vv, o[i][c]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qi = q0 + 4 * ty + i;
        if (qi >= p.Sq) continue;
        const float inv = l[i] > 0.f ? 1.f / l[i] : 0.f;
        float* op = p.o + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh + (int64_t)qi * p.o_ss;
#pragma unroll
        for (int c = 0; c < NC; ++c) op[tx + 16 * c] = o[i][c] * inv;
        if (p.lse && tx == 0) p.lse[((int64_t)b * p.H + h) * p.Sq + qi] = l[i] > 0.f ? m[i] + __logf(l[i]) : -INFINITY;
    }
}

template <int D> constexpr int f32_lds_bytes() { return (2 * D * 68 + 64 * D + 64 * 68) * 4; }

}  // namespace pfa

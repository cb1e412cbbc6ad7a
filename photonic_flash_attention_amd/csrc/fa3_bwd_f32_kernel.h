// fa3_bwd_f32_kernel.h -- fp32 backward of the attention core on the matrix cores (v_mfma_f32_32x32x2_f32: fp32 operands, bitwise an
// fmaf chain, at the fp32 vector rate), for the cases the 16-bit MFMA backward does not take: fp32 operands (fp32 modules, the reference's
// default dtype) and attention dropout (the reference's dense branch, flash_attention_3.py:174-175; it differentiates through its eager
// core with autograd, this is that gradient).  Companion of fa3_fwd_f32_kernel.h.  Round 2's version was a VALU micro-tiling.
//
//   S = scale Q K^T (+ masks),  P = exp(S - lse),  Pd = P o keep * drop_scale,  O = Pd V
//   delta_i = sum_d dO_id O_id,  dPd = dO V^T,  dS = P o (dPd o keep * drop_scale - delta_i)
//   dQ = scale dS K,   dK = scale dS^T Q,   dV = Pd^T dO
//
// One kernel, two roles (template MODE), no atomics, bitwise reproducible.  A workgroup of 4 waves owns 128 rows of the OWN side
// (MODE 0: queries -> dQ; MODE 1: keys -> dK, dV), a wave 32 of them, and walks the OTHER side 32 rows at a time through a
// double-buffered padded LDS image of its two operands (MODE 0: K, V; MODE 1: Q, dO).  A lane keeps half of its own two rows in
// registers (contraction index paired (d, d + D/2), as in the forward).  Per tile, with "own" on the MFMA lane (column) and the tile's
// rows on the accumulator registers (row crow(e, h) = (e & 3) + 8 (e >> 2) + 4 h):
//   x1 = Y1 . own1^T   (MODE 0: S^T = K Q^T;  MODE 1: S = Q K^T)         x2 = Y2 . own2^T   (MODE 0: dP^T = V dO^T;  MODE 1: dP = dO V^T)
//   P, dS element-wise (lse / delta: the lane's own scalars in MODE 0, per register from LDS in MODE 1)
//   acc1^T += Y1^T . dS  (MODE 0: dQ^T += K^T dS^T;  MODE 1: dK^T += Q^T dS)      MODE 1 only: acc2^T += Y2^T . Pd  (dV^T += dO^T Pd)
// -- the SAME code for both roles: dS / Pd registers are the B operand as they stand, the A operand Y[crow(e, h)][32 db + (lane & 31)] is a
// conflict-free ds_read_b32 of the row-major image.  192 (MODE 0) / 256 (MODE 1) MFMAs of depth 2 per 32 x 32 tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfa {

struct F32BwdParams {
    const float* q; const float* k; const float* v; const float* o; const float* dout; const float* lse;
    float* dq; float* dk; float* dv;
    const int32_t* seqlens_k;
    const uint8_t* mask;             // optional u8 mask of the forward (0 = masked), byte strides
    const uint8_t* drop_mask;        // optional keep mask [B][H][Sq][Sk] of the forward (non-zero = keep)
    int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss, do_sb, do_sh, do_ss;
    int64_t dq_sb, dq_sh, dq_ss, dk_sb, dk_sh, dk_ss, dv_sb, dv_sh, dv_ss;     // element strides
    int64_t m_sb, m_sh, m_sq, m_sk;
    int32_t B, H, Sq, Sk, causal;
    float scale, drop_scale;
};

constexpr int F32B_BM = 128, F32B_BN = 32;
template <int D> constexpr int f32_bwd_lds_bytes() { return (2 * 2 * F32B_BN * (D + 4) + 2 * 2 * F32B_BN) * 4; }      // 2 stages x (2 images + lse / delta rows)

typedef float f32x16_b __attribute__((ext_vector_type(16)));

template <int D, int MODE>
__global__ __launch_bounds__(256, 1) void fa3_bwd_f32_kernel(const F32BwdParams p) {
    constexpr int BM = F32B_BM, BN = F32B_BN, LD = D + 4, HD = D / 2, NDB = D / 32;
    constexpr int TILE = BN * LD;
    constexpr int NLD = BN * D / 4 / 256;
    constexpr int STAGE = 2 * TILE + 2 * BN;               // floats per stage: Y1, Y2, lse[BN], delta[BN]
    extern __shared__ __attribute__((aligned(16))) float smem_b[];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int n_own = MODE == 0 ? p.Sq : p.Sk, n_oth = MODE == 0 ? p.Sk : p.Sq;
    const int nblk = (n_own + BM - 1) / BM;
    const int bh = blockIdx.x / nblk;
    int blk = blockIdx.x - bh * nblk;
    if (p.causal && MODE == 0) blk = nblk - 1 - blk;       // heaviest query blocks first (key blocks: the first ones are the heaviest already)
    const int b = bh / p.H, h = bh - b * p.H;
    const int r0 = blk * BM, rw = r0 + wave * 32, own = rw + r;
    int kv_len = p.Sk;
    if (p.seqlens_k) kv_len = min(kv_len, max(p.seqlens_k[b], 0));
    const float* qp = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const float* kp = p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const float* vp = p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
    const float* op = p.o + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh;
    const float* gp = p.dout + (int64_t)b * p.do_sb + (int64_t)h * p.do_sh;
    const float* lsep = p.lse + ((int64_t)b * p.H + h) * p.Sq;
    const float* own1 = MODE == 0 ? qp : kp;
    const float* own2 = MODE == 0 ? gp : vp;
    const int64_t own1_ss = MODE == 0 ? p.q_ss : p.k_ss, own2_ss = MODE == 0 ? p.do_ss : p.v_ss;
    const float* oth1 = MODE == 0 ? kp : qp;
    const float* oth2 = MODE == 0 ? vp : gp;
    const int64_t oth1_ss = MODE == 0 ? p.k_ss : p.q_ss, oth2_ss = MODE == 0 ? p.v_ss : p.do_ss;
    const float c = p.scale * 1.4426950408889634f;         // exp(x) = exp2(x log2 e)

    // this lane's half (d in [hh D/2, hh D/2 + D/2)) of its two own rows, kept for the whole block
    float f1[HD], f2[HD];
    float my_lse = INFINITY, my_delta = 0.f;               // MODE 0: the lane's query
    {
        const bool ok = own < n_own;
        const float* a1 = own1 + (int64_t)(ok ? own : 0) * own1_ss + hh * HD;
        const float* a2 = own2 + (int64_t)(ok ? own : 0) * own2_ss + hh * HD;
        const float* ao = op + (int64_t)(ok ? own : 0) * p.o_ss + hh * HD;
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            const float4 x = ok ? *(const float4*)(a1 + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 y = ok ? *(const float4*)(a2 + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            f1[4 * i] = x.x; f1[4 * i + 1] = x.y; f1[4 * i + 2] = x.z; f1[4 * i + 3] = x.w;
            f2[4 * i] = y.x; f2[4 * i + 1] = y.y; f2[4 * i + 2] = y.z; f2[4 * i + 3] = y.w;
            if (MODE == 0 && ok) {                         // delta of the own query: this lane's half of sum_d dO O
                d = __builtin_fmaf(y.x, ao[4 * i], d); d = __builtin_fmaf(y.y, ao[4 * i + 1], d);
                d = __builtin_fmaf(y.z, ao[4 * i + 2], d); d = __builtin_fmaf(y.w, ao[4 * i + 3], d);
            }
        }
        if (MODE == 0) {
            my_delta = d + __shfl_xor(d, 32);
            my_lse = ok ? lsep[own] : INFINITY;
        }
    }
    f32x16_b acc1[NDB], acc2[MODE == 1 ? NDB : 1];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc1[db][e] = 0.f;
            if (MODE == 1) acc2[db][e] = 0.f;
        }

    // range of the other side that can see / be seen by this block
    int t_begin = 0, t_end = n_oth;
    if (MODE == 0) {
        t_end = p.causal ? min(kv_len, r0 + BM) : kv_len;
    } else if (p.causal) {
        t_begin = (r0 / BN) * BN;                          // queries before the block's first key see none of its keys
    }
    const int nt = t_end > t_begin ? (t_end - t_begin + BN - 1) / BN : 0;

    float4 y1r[NLD], y2r[NLD];
    auto fetch_stash = [&](int t, int stage) {
        const int t0 = t_begin + t * BN;
        float* Y1 = smem_b + stage * STAGE;
        float* Y2 = Y1 + TILE;
        float* rowc = Y2 + TILE;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j, row = i / (D / 4), c4 = i - row * (D / 4);
            const bool ok = t0 + row < n_oth;
            y1r[j] = ok ? *(const float4*)(oth1 + (int64_t)(t0 + row) * oth1_ss + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            y2r[j] = ok ? *(const float4*)(oth2 + (int64_t)(t0 + row) * oth2_ss + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (MODE == 1) {                                   // lse and delta of the tile's 32 queries: 8 threads per query
            const int row = tid >> 3, part = tid & 7;
            float d = 0.f;
            if (t0 + row < p.Sq)
                for (int cc = part * 4; cc < D; cc += 32) {
                    const float4 g = *(const float4*)(gp + (int64_t)(t0 + row) * p.do_ss + cc);
                    const float* oo = op + (int64_t)(t0 + row) * p.o_ss + cc;
                    d = __builtin_fmaf(g.x, oo[0], d); d = __builtin_fmaf(g.y, oo[1], d);
                    d = __builtin_fmaf(g.z, oo[2], d); d = __builtin_fmaf(g.w, oo[3], d);
                }
            d += __shfl_xor(d, 1, 8);
            d += __shfl_xor(d, 2, 8);
            d += __shfl_xor(d, 4, 8);
            if (part == 0) {
                rowc[row] = t0 + row < p.Sq ? lsep[t0 + row] : INFINITY;
                rowc[BN + row] = d;
            }
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j, row = i / (D / 4), c4 = i - row * (D / 4);
            *(float4*)(Y1 + row * LD + 4 * c4) = y1r[j];
            *(float4*)(Y2 + row * LD + 4 * c4) = y2r[j];
        }
    };
    if (nt > 0) fetch_stash(0, 0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int t0 = t_begin + t * BN;
        if (t + 1 < nt) fetch_stash(t + 1, (t + 1) & 1);   // the other stage was last read one iteration ago (barrier below)
        const float* Y1 = smem_b + (t & 1) * STAGE;
        const float* Y2 = Y1 + TILE;
        const float* rowc = Y2 + TILE;
        // wave-uniform: does this wave's strip meet the tile at all under the causal mask?
        const bool live = !p.causal || (MODE == 0 ? t0 <= rw + 31 : t0 + BN - 1 >= rw);
        if (live) {
            f32x16_b x1, x2;
#pragma unroll
            for (int e = 0; e < 16; ++e) x1[e] = x2[e] = 0.f;
            const float* y1row = Y1 + r * LD + hh * HD;
            const float* y2row = Y2 + r * LD + hh * HD;
#pragma unroll
            for (int i = 0; i < HD / 4; ++i) {
                const float4 a = *(const float4*)(y1row + 4 * i);
                x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, f1[4 * i], x1, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, f1[4 * i + 1], x1, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, f1[4 * i + 2], x1, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, f1[4 * i + 3], x1, 0, 0, 0);
                if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < HD / 4; ++i) {
                const float4 a = *(const float4*)(y2row + 4 * i);
                x2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, f2[4 * i], x2, 0, 0, 0);
                x2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, f2[4 * i + 1], x2, 0, 0, 0);
                x2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, f2[4 * i + 2], x2, 0, 0, 0);
                x2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, f2[4 * i + 3], x2, 0, 0, 0);
                if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
            }
            // ---- P and dS: element e is (own row `own`, other row t0 + crow(e, hh))
            const bool easy = !p.mask && !p.drop_mask && !p.causal && t0 + BN <= (MODE == 0 ? kv_len : p.Sq) &&
                              (MODE == 0 ? rw + 31 < p.Sq : rw + 31 < kv_len);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int oth = t0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const int qi = MODE == 0 ? own : oth, kj = MODE == 0 ? oth : own;
                const float lse = MODE == 0 ? my_lse : rowc[(e & 3) + 8 * (e >> 2) + 4 * hh];
                const float dl = MODE == 0 ? my_delta : rowc[BN + (e & 3) + 8 * (e >> 2) + 4 * hh];
                bool vis = true;
                float keep = 1.f;
                if (!easy) {
                    vis = qi < p.Sq && kj < kv_len && (!p.causal || kj <= qi);
                    if (vis && p.mask) vis = p.mask[(int64_t)b * p.m_sb + (int64_t)h * p.m_sh + (int64_t)qi * p.m_sq + (int64_t)kj * p.m_sk] != 0;
                    if (p.drop_mask && vis) keep = p.drop_mask[(((int64_t)b * p.H + h) * p.Sq + qi) * p.Sk + kj] ? p.drop_scale : 0.f;
                }
                // lse = -inf: a row the forward found fully masked; +inf: a row that does not exist
                const float pr = (vis && lse != -INFINITY && lse != INFINITY) ? __builtin_amdgcn_exp2f(__builtin_fmaf(x1[e], c, -lse * 1.4426950408889634f)) : 0.f;
                x1[e] = pr * (x2[e] * keep - dl) * p.scale;                 // dS (scaled)
                x2[e] = pr * keep;                                          // Pd
            }
            // ---- acc1^T += Y1^T dS;  MODE 1: acc2^T += Y2^T Pd
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * hh;
                    acc1[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(Y1[row * LD + 32 * db + r], x1[e], acc1[db], 0, 0, 0);
                    if (MODE == 1) acc2[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(Y2[row * LD + 32 * db + r], x2[e], acc2[db], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: lane (own row, hh) stores columns 32 db + 8 g + 4 hh + 0..3
    if (own < n_own) {
        float* d1 = MODE == 0 ? p.dq + (int64_t)b * p.dq_sb + (int64_t)h * p.dq_sh + (int64_t)own * p.dq_ss
                              : p.dk + (int64_t)b * p.dk_sb + (int64_t)h * p.dk_sh + (int64_t)own * p.dk_ss;
        float* d2 = p.dv + (int64_t)b * p.dv_sb + (int64_t)h * p.dv_sh + (int64_t)own * p.dv_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = 32 * db + 8 * g + 4 * hh;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    d1[col + j] = acc1[db][4 * g + j];
                    if (MODE == 1) d2[col + j] = acc2[db][4 * g + j];
                }
            }
    }
}

}  // namespace pfa

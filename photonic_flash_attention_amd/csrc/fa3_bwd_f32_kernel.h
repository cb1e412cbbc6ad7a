// fa3_bwd_f32_kernel.h -- fp32 backward of the attention core, for the cases the MFMA backward does not take: fp32 operands and
// attention dropout (the reference's dense branch, flash_attention_3.py:174-175; it differentiates through its eager core with
// autograd, this is that gradient).  Companion of fa3_fwd_f32_kernel.h: every product and sum in fp32, plain register micro-tiles.
//
//   S = scale Q K^T (+ masks),  P = exp(S - lse),  Pd = P o keep * drop_scale,  O = Pd V
//   delta_i = sum_d dO_id O_id,  dPd = dO V^T,  dS = P o (dPd o keep * drop_scale - delta_i)
//   dQ = scale dS K,   dK = scale dS^T Q,   dV = Pd^T dO
//
// One kernel, two roles (template MODE): MODE 0 owns 64 query rows and walks the keys 32 at a time (-> dQ); MODE 1 owns 64 keys and
// walks the queries 32 at a time (-> dK, dV).  No atomics, bitwise reproducible.  The "own" operands (Q, dO / K, V) and the tile
// of the other side sit transposed in LDS ([d][row]); thread (tid / 8, tid % 8) computes a 2 x 4 micro-tile of S and dP, the tiles of
// dS (and Pd) go through LDS, thread (tid / 16, tid % 16) accumulates 4 rows x D / 16 columns of the gradients.
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

template <int D, int MODE>
__global__ __launch_bounds__(256) void fa3_bwd_f32_kernel(const F32BwdParams p) {
    constexpr int BM = 64, BN = 32, LO = BM + 4, LT = BN + 4, NC = D / 16;
    extern __shared__ __attribute__((aligned(16))) float smem_b[];
    float* X1 = smem_b;                  // own operand 1, transposed [D][LO]: Q (MODE 0) / K (MODE 1)
    float* X2 = X1 + D * LO;             // own operand 2: dO / V
    float* Y1 = X2 + D * LO;             // other side's tile, transposed [D][LT]: K / Q
    float* Y2 = Y1 + D * LT;             //                                        V / dO
    float* T1 = Y2 + D * LT;             // dS tile [BM][LT]
    float* T2 = T1 + BM * LT;            // Pd tile [BM][LT] (MODE 1)
    float* rowc = T2 + BM * LT;          // per other-side row: lse[BN], delta[BN] (MODE 1); own rows: delta[BM] (MODE 0)

    const int tid = threadIdx.x;
    const int n_own = MODE == 0 ? p.Sq : p.Sk, n_oth = MODE == 0 ? p.Sk : p.Sq;
    const int nblk = (n_own + BM - 1) / BM;
    const int bh = blockIdx.x / nblk, blk = blockIdx.x - bh * nblk;
    const int b = bh / p.H, h = bh - b * p.H;
    const int r0 = blk * BM;
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

    for (int i = tid; i < BM * (D / 4); i += 256) {
        const int r = i / (D / 4), c4 = i - r * (D / 4);
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
        if (r0 + r < n_own) {
            x = *(const float4*)(own1 + (int64_t)(r0 + r) * own1_ss + 4 * c4);
            y = *(const float4*)(own2 + (int64_t)(r0 + r) * own2_ss + 4 * c4);
        }
        X1[(4 * c4 + 0) * LO + r] = x.x; X1[(4 * c4 + 1) * LO + r] = x.y; X1[(4 * c4 + 2) * LO + r] = x.z; X1[(4 * c4 + 3) * LO + r] = x.w;
        X2[(4 * c4 + 0) * LO + r] = y.x; X2[(4 * c4 + 1) * LO + r] = y.y; X2[(4 * c4 + 2) * LO + r] = y.z; X2[(4 * c4 + 3) * LO + r] = y.w;
    }
    if (MODE == 0) {                     // delta of the own query rows: 4 threads per row
        const int r = tid >> 2, part = tid & 3;
        float d = 0.f;
        if (r0 + r < p.Sq)
            for (int c = part; c < D; c += 4) d = __builtin_fmaf(gp[(int64_t)(r0 + r) * p.do_ss + c], op[(int64_t)(r0 + r) * p.o_ss + c], d);
        d += __shfl_xor(d, 1, 4);
        d += __shfl_xor(d, 2, 4);
        if (part == 0) rowc[r] = d;
    }
    const int ty = tid >> 3, tx = tid & 7;             // S / dP micro-tile: own rows 2 ty, 2 ty + 1; other rows 4 tx .. +3
    const int ty2 = tid >> 4, tx2 = tid & 15;          // gradient tile: own rows 4 ty2 .. +3, columns tx2 + 16 c
    float acc1[4][NC], acc2[4][NC];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc1[i][c] = acc2[i][c] = 0.f;

    // range of the other side that can see / be seen by this block
    int t_begin = 0, t_end = n_oth;
    if (MODE == 0) {
        t_end = p.causal ? min(kv_len, r0 + BM) : kv_len;
    } else if (p.causal) {
        t_begin = (r0 / BN) * BN;                      // queries before the block's first key see none of its keys
    }
    for (int t0 = t_begin; t0 < t_end; t0 += BN) {
        __syncthreads();
        for (int i = tid; i < BN * (D / 4); i += 256) {
            const int r = i / (D / 4), c4 = i - r * (D / 4);
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
            if (t0 + r < n_oth) {
                x = *(const float4*)(oth1 + (int64_t)(t0 + r) * oth1_ss + 4 * c4);
                y = *(const float4*)(oth2 + (int64_t)(t0 + r) * oth2_ss + 4 * c4);
            }
            Y1[(4 * c4 + 0) * LT + r] = x.x; Y1[(4 * c4 + 1) * LT + r] = x.y; Y1[(4 * c4 + 2) * LT + r] = x.z; Y1[(4 * c4 + 3) * LT + r] = x.w;
            Y2[(4 * c4 + 0) * LT + r] = y.x; Y2[(4 * c4 + 1) * LT + r] = y.y; Y2[(4 * c4 + 2) * LT + r] = y.z; Y2[(4 * c4 + 3) * LT + r] = y.w;
        }
        if (MODE == 1) {                 // lse and delta of the 32 queries of this tile: 8 threads per query
            const int r = tid >> 3, part = tid & 7;
            float d = 0.f;
            if (t0 + r < p.Sq)
                for (int c = part; c < D; c += 8) d = __builtin_fmaf(gp[(int64_t)(t0 + r) * p.do_ss + c], op[(int64_t)(t0 + r) * p.o_ss + c], d);
            d += __shfl_xor(d, 1, 8);
            d += __shfl_xor(d, 2, 8);
            d += __shfl_xor(d, 4, 8);
            if (part == 0) {
                rowc[r] = t0 + r < p.Sq ? lsep[t0 + r] : INFINITY;
                rowc[BN + r] = d;
            }
        }
        __syncthreads();
        float s[2][4], dp[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[i][j] = dp[i][j] = 0.f;
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
            const float a0 = X1[d * LO + 2 * ty], a1 = X1[d * LO + 2 * ty + 1];
            const float g0 = X2[d * LO + 2 * ty], g1 = X2[d * LO + 2 * ty + 1];
            const float4 c = *(const float4*)(Y1 + d * LT + 4 * tx);
            const float4 e = *(const float4*)(Y2 + d * LT + 4 * tx);
            const float cv[4] = {c.x, c.y, c.z, c.w}, ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[0][j] = __builtin_fmaf(a0, cv[j], s[0][j]);
                s[1][j] = __builtin_fmaf(a1, cv[j], s[1][j]);
                dp[0][j] = __builtin_fmaf(g0, ev[j], dp[0][j]);
                dp[1][j] = __builtin_fmaf(g1, ev[j], dp[1][j]);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int own = r0 + 2 * ty + i, oth = t0 + 4 * tx + j;
                const int qi = MODE == 0 ? own : oth, kj = MODE == 0 ? oth : own;
                bool vis = qi < p.Sq && kj < kv_len && (!p.causal || kj <= qi);
                if (vis && p.mask) vis = p.mask[(int64_t)b * p.m_sb + (int64_t)h * p.m_sh + (int64_t)qi * p.m_sq + (int64_t)kj * p.m_sk] != 0;
                const float lse = vis ? (MODE == 0 ? lsep[qi] : rowc[4 * tx + j]) : INFINITY;
                const float dl = MODE == 0 ? rowc[2 * ty + i] : rowc[BN + 4 * tx + j];
                const float pr = (vis && lse != -INFINITY) ? __expf(s[i][j] * p.scale - lse) : 0.f;      // lse = -inf: a row the forward found fully masked
                float keep = 1.f;
                if (p.drop_mask && vis) keep = p.drop_mask[(((int64_t)b * p.H + h) * p.Sq + qi) * p.Sk + kj] ? p.drop_scale : 0.f;
                const float ds = pr * (dp[i][j] * keep - dl);
                T1[(2 * ty + i) * LT + 4 * tx + j] = ds * p.scale;
                if (MODE == 1) T2[(2 * ty + i) * LT + 4 * tx + j] = pr * keep;
            }
        __syncthreads();
        // gradients: own rows 4 ty2 .. +3, columns tx2 + 16 c:  acc1 += dS x (other operand 1),  acc2 += Pd x (other operand 2)
#pragma unroll 4
        for (int j = 0; j < BN; ++j) {
            float a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = T1[(4 * ty2 + i) * LT + j];
                if (MODE == 1) w[i] = T2[(4 * ty2 + i) * LT + j];
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float y1 = Y1[(tx2 + 16 * c) * LT + j];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc1[i][c] = __builtin_fmaf(a[i], y1, acc1[i][c]);
                if (MODE == 1) {
                    const float y2 = Y2[(tx2 + 16 * c) * LT + j];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc2[i][c] = __builtin_fmaf(w[i], y2, acc2[i][c]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 4 * ty2 + i;
        if (r >= n_own) continue;
        if (MODE == 0) {
            float* d = p.dq + (int64_t)b * p.dq_sb + (int64_t)h * p.dq_sh + (int64_t)r * p.dq_ss;
#pragma unroll
            for (int c = 0; c < NC; ++c) d[tx2 + 16 * c] = acc1[i][c];
        } else {
            float* dk = p.dk + (int64_t)b * p.dk_sb + (int64_t)h * p.dk_sh + (int64_t)r * p.dk_ss;
            float* dv = p.dv + (int64_t)b * p.dv_sb + (int64_t)h * p.dv_sh + (int64_t)r * p.dv_ss;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                dk[tx2 + 16 * c] = acc1[i][c];
                dv[tx2 + 16 * c] = acc2[i][c];
            }
        }
    }
}

template <int D> constexpr int f32_bwd_lds_bytes() { return (2 * D * 68 + 2 * D * 36 + 2 * 64 * 36 + 128) * 4; }

}  // namespace pfa

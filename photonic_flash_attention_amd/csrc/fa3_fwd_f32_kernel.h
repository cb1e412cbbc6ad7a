// fa3_fwd_f32_kernel.h -- EXACT fp32 forward for fp32 modules (the reference's default dtype, flash_attention_3.py:19-27; BASELINE
// config C1 is one) on the matrix cores: gfx950's v_mfma_f32_32x32x2_f32 takes fp32 operands and is bitwise an fmaf chain, at the
// fp32 vector rate (157 TFLOP/s peak; MI355X_MICROARCH.md, Matrix cores) -- nothing is rounded to 16 bits anywhere.  It replaces the
// same seam as the 16-bit kernels (flash_attention_3.py:120-262) for callers that hand over fp32 operands and want the reference's
// fp32 numbers (<= 2e-5 of the oracle, not the ~1e-2 of bf16 operands).  Round 2's version of this file was a plain VALU tiling at a few
// TFLOP/s; the matrix form is the same algorithm 20-30 x faster.
//
// Geometry: a workgroup of 4 waves owns 128 query rows of one (batch, head), a wave 32 of them; keys are walked in tiles of 32 through
// a double-buffered LDS image (K and V rows as they lie in memory, rows padded by 16 B so that a lane-per-row read is conflict-free).
// Both products are "swapped" like the 16-bit kernels':
//   S^T (32 keys x 32 queries) = K Q^T: per MFMA one K value (A) and one Q value (B) per lane; the contraction index is paired as
//        (d, d + D/2) so that a lane reads D/2 CONTIGUOUS floats of its key row (ds_read_b128) and keeps D/2 floats of its query row in
//        registers for the whole item.  Each lane then holds 16 scores of ONE query (lane & 31): keys crow(e, h) = (e & 3) + 8 (e >> 2) + 4 h.
//   O^T (D x 32 queries) += V^T P^T: the exponentials ARE the B operand, register e at k-step e (keys crow(e, 0) | crow(e, 1) on the two
//        lane halves); the A operand V[crow(e, h)][32 db + (lane & 31)] is a conflict-free ds_read_b32 of the row-major V image.
// Online softmax exactly as flash_attention_3.py:239-250 (un-normalised until the end), a row's two lanes joined by one DPP swap; O is
// rescaled only when some row's maximum moved.  Masks (causal, seqlens_k, any u8 mask), grouped K/V heads, LSE and the dense branch's
// dropout keep-mask as before.
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

constexpr int F32_BM = 128, F32_BN = 32;
template <int D> constexpr int f32_lds_bytes() { return 2 * 2 * F32_BN * (D + 4) * 4; }      // 2 stages x (K + V) x 32 rows x (D + 4) floats

typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int D>
__global__ __launch_bounds__(256, 2) void fa3_fwd_f32_kernel(const F32Params p) {
    constexpr int BM = F32_BM, BN = F32_BN, LD = D + 4, HD = D / 2, NDB = D / 32;
    constexpr int TILE = BN * LD;                          // floats of one K (or V) image
    constexpr int NLD = BN * D / 4 / 256;                  // float4 loads per thread and image
    extern __shared__ __attribute__((aligned(16))) float smem_f[];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int nqb = (p.Sq + BM - 1) / BM;
    const int bh = blockIdx.x / nqb;
    int qb = blockIdx.x - bh * nqb;
    if (p.causal) qb = nqb - 1 - qb;                       // heaviest blocks first
    const int b = bh / p.H, h = bh - b * p.H, hkv = h / p.kv_group;
    const int q0 = qb * BM, qw = q0 + wave * 32, qi = qw + r;
    const float* qp = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const float* kp = p.k + (int64_t)b * p.k_sb + (int64_t)hkv * p.k_sh;
    const float* vp = p.v + (int64_t)b * p.v_sb + (int64_t)hkv * p.v_sh;
    int kv_len = p.Sk;
    if (p.seqlens_k) kv_len = min(kv_len, max(p.seqlens_k[b], 0));
    const int kv_end = p.causal ? min(kv_len, q0 + BM) : kv_len;
    const int nt = (kv_end + BN - 1) / BN;
    const float c = p.scale * 1.4426950408889634f;         // exp(x) = exp2(x log2 e)

    // this lane's half of its query row (d in [hh D/2, hh D/2 + D/2)), kept for the whole item
    float qf[HD];
    {
        const bool ok = qi < p.Sq;
        const float* qr = qp + (int64_t)(ok ? qi : 0) * p.q_ss + hh * HD;
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            const float4 x = ok ? *(const float4*)(qr + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            qf[4 * i] = x.x; qf[4 * i + 1] = x.y; qf[4 * i + 2] = x.z; qf[4 * i + 3] = x.w;
        }
    }
    f32x16_t o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
    float m = -INFINITY, l = 0.f;                          // m: the row's (both lanes'); l: this lane's share of the row sum

    // tile loads: thread t takes float4 (row, c4) = (i / (D/4), i % (D/4)), i = t + 256 j
    float4 kreg[NLD], vreg[NLD];
    auto fetch = [&](int t) {
        const int k0 = t * BN;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j, row = i / (D / 4), c4 = i - row * (D / 4);
            const bool ok = k0 + row < p.Sk;
            kreg[j] = ok ? *(const float4*)(kp + (int64_t)(k0 + row) * p.k_ss + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            vreg[j] = ok ? *(const float4*)(vp + (int64_t)(k0 + row) * p.v_ss + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int stage) {
        float* Ks = smem_f + stage * 2 * TILE;
        float* Vs = Ks + TILE;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j, row = i / (D / 4), c4 = i - row * (D / 4);
            *(float4*)(Ks + row * LD + 4 * c4) = kreg[j];
            *(float4*)(Vs + row * LD + 4 * c4) = vreg[j];
        }
    };
    if (nt > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int k0 = t * BN;
        if (t + 1 < nt) {                                  // next tile: global -> registers -> the other stage, before this tile's math (the
            fetch(t + 1);                                  // co-resident workgroup's waves cover the wait; nothing stays live across the MFMAs)
            stash((t + 1) & 1);
        }
        const float* Ks = smem_f + (t & 1) * 2 * TILE;
        const float* Vs = Ks + TILE;
        const bool live = !p.causal || k0 <= qw + 31;      // (wave-uniform: tiles right of the wave's last row are skipped)
        if (live) {
            // ---- S^T = K Q^T: D/2 MFMAs of depth 2
            f32x16_t s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            const float* kr = Ks + r * LD + hh * HD;
#pragma unroll
            for (int i = 0; i < HD / 4; ++i) {
                const float4 x = *(const float4*)(kr + 4 * i);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, qf[4 * i], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, qf[4 * i + 1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, qf[4 * i + 2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, qf[4 * i + 3], s, 0, 0, 0);
                if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);      // (keeps hipcc from hoisting all D/8 row reads: registers)
            }
            // ---- masks: score e of this lane is (query qi, key k0 + crow(e, hh))
            const bool edge = k0 + BN > kv_len || (p.causal && k0 + BN - 1 > qw) || p.mask != nullptr || qw + 31 >= p.Sq;
            if (edge) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kj = k0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    bool vis = kj < kv_len && (!p.causal || kj <= qi) && qi < p.Sq;
                    if (vis && p.mask)
                        vis = p.mask[(int64_t)b * p.m_sb + (int64_t)h * p.m_sh + (int64_t)qi * p.m_sq + (int64_t)kj * p.m_sk] != 0;
                    s[e] = vis ? s[e] : -INFINITY;
                }
            }
            // ---- online softmax (flash_attention_3.py:239-250), in the exp2 domain: x = c s - c m
            float mx = s[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m, mx);
            const bool dead = m_new == -INFINITY;            // no visible key so far: l = 0, O = 0 stay
            const float mc = dead ? 0.f : m_new * c;
            const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m * c - mc);      // (m = -inf: exp2(-inf) = 0, and l = 0, O = 0 anyway)
            float rs = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = dead ? 0.f : __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c, -mc));
                rs += pe;
                s[e] = pe;
            }
            if (p.drop_mask) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kj = k0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    const bool keep = qi < p.Sq && kj < p.Sk && p.drop_mask[(((int64_t)b * p.H + h) * p.Sq + qi) * p.Sk + kj] != 0;
                    s[e] = keep ? s[e] * p.drop_scale : 0.f;
                }
            }
            l = l * alpha + rs;
            if (__builtin_amdgcn_ballot_w64(m_new != m) != 0) {      // some row's maximum moved: rescale O (all of a lane's O belongs to its query)
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
            }
            m = m_new;
            // ---- O^T += V^T P^T: per d block 16 MFMAs of depth 2, k-step e = keys crow(e, 0) | crow(e, 1)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float a = Vs[((e & 3) + 8 * (e >> 2) + 4 * hh) * LD + 32 * db + r];
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[e], o[db], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: the row's two lanes join their sums; lane (query, hh) stores d = 32 db + 8 g + 4 hh + 0..3
    l += __shfl_xor(l, 32);
    if (qi < p.Sq) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        float* op = p.o + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh + (int64_t)qi * p.o_ss;
        const bool vec = ((reinterpret_cast<uintptr_t>(p.o) & 15u) == 0) && ((p.o_sb | p.o_sh | p.o_ss) & 3) == 0;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float* dst = op + 32 * db + 8 * g + 4 * hh;
                const float4 x = make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
                if (vec) {
                    *(float4*)dst = x;
                } else {
                    dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
                }
            }
        if (p.lse && hh == 0) p.lse[((int64_t)b * p.H + h) * p.Sq + qi] = l > 0.f ? m * p.scale + __logf(l) : -INFINITY;
    }
}

}  // namespace pfa

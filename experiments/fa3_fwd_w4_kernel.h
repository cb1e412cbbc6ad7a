// fa3_fwd_w4_kernel.h -- Flash-Attention forward for MI355X: 4 waves x 64 query rows, ONE wave per SIMD.
//
// Same math, HBM/LDS layouts, MFMA operand maps and DMA staging as fa3_fwd_kernel.h (read that first).  The
// 8-wave kernel puts two waves on every SIMD; both run the same program between the same barriers, so they sit
// in QK^T / softmax / PV together (matrix pipe contended, then idle: profiles/r01_v13_stamp_phases.txt).
// Here a workgroup is 4 waves = one per SIMD with the whole 512-entry register file, and each wave owns TWO
// 32-row query blocks a, b whose phases are staggered inside the wave's own instruction stream:
//
//      matrix pipe :  S_a = K Q_a^T | S_b = K Q_b^T      | O_a += V^T P_a^T   | O_b += V^T P_b^T
//      vector ALU  :                | softmax(a) -> P_a  | softmax(b) -> P_b  |
//
// i.e. every softmax runs beside MFMAs of the OTHER query block (independent data, same basic block), and the
// matrix-only stretches have the pipe to themselves.  O accumulators (2 x 64 fp32) live in the accumulator half
// of the register file.  One barrier per 64-key tile, K/V by LDS-DMA (8 pieces per wave per tile).
#pragma once
#include "fa3_fwd_kernel.h"

namespace pfa {

template <typename T, int D, bool CAUSAL, bool SPLITP, bool KMASK, int VAR, typename OT>
__global__ __launch_bounds__(256, 1) void fa3_fwd_w4_kernel(const FwdParams p) {
    using E = Elem<T>;
    using v8 = typename E::v8;
    using v4 = typename E::v4;
    typedef __attribute__((address_space(3))) v8 lds_v8;
    constexpr int NW = 4;
    constexpr int BLOCK_M = 256;              // 4 waves x 2 query blocks x 32 rows
    constexpr int KS = D / 16;
    constexpr int DB = D / 32;
    constexpr int TILE_BYTES = BLOCK_N * D * 2;
    constexpr int BUF_BYTES = 2 * TILE_BYTES;
    constexpr int HALF_TILE = TILE_BYTES / 2;
    constexpr int PIECES = TILE_BYTES / 1024;
    constexpr int PPW = PIECES / NW;          // 4 (D=128) or 2 (D=64) pieces per wave per image

    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* const smem_l = (lds_char*)smem;
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem_l;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    const int BH = p.B * p.H;
    const int n = blockIdx.x;
    const int qrank = n / BH;
    const int bh = n - qrank * BH;
    const int qblk = CAUSAL ? (p.nqblk - 1 - qrank) : qrank;
    const int b = bh / p.H;
    const int hh = bh - b * p.H;

    const int q0 = qblk * BLOCK_M;
    const int wave_q0 = q0 + wave * 64;       // query block a = rows wave_q0..+31, b = +32..+63

    int kv_len = p.Sk;
    if (p.seqlens_k) kv_len = min(kv_len, max(p.seqlens_k[b], 0));
    const int kv_end = CAUSAL ? min(kv_len, q0 + BLOCK_M) : kv_len;
    const int nt = (kv_end + BLOCK_N - 1) / BLOCK_N;

    const T* __restrict__ qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)hh * p.q_sh;
    const T* __restrict__ kp = (const T*)p.k + (int64_t)b * p.k_sb + (int64_t)hh * p.k_sh;
    const T* __restrict__ vp = (const T*)p.v + (int64_t)b * p.v_sb + (int64_t)hh * p.v_sh;

    // per query block state
    struct QB {
        v8 qf[KS];
        f32x16 o[DB];
        float m, l;
        int my_q, q_first, kv_end;          // this lane's row, the block's first row, keys the block needs
        const uint8_t* mp;
    };
    QB A, Bq;
    auto init_qb = [&](QB& X, int first) {
        X.q_first = first;
        X.my_q = first + r;
        X.kv_end = CAUSAL ? min(kv_len, first + 32) : kv_len;
        const int qrow = min(X.my_q, p.Sq - 1);
        const T* src = qp + (int64_t)qrow * p.q_ss + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) X.qf[ks] = *(const v8*)(src + 16 * ks);
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) X.o[i][e] = 0.f;
        X.m = -1e30f;
        X.l = 0.f;
        X.mp = KMASK ? p.mask + (int64_t)b * p.m_sb + (int64_t)hh * p.m_sh + (int64_t)qrow * p.m_sq : nullptr;
    };
    init_qb(A, wave_q0);
    init_qb(Bq, wave_q0 + 32);

    // ---- LDS-DMA through buffer descriptors ---------------------------------------------------------------------------
    uint32_t kvoff[PPW], vvoff[PPW];
    {
        const int R0 = 4 * wave + (lane >> 4);
        const int sw = ((R0 & 3) << 2) | ((R0 >> 2) & 3);
        const int cc = (lane & 15) ^ sw;
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            int key, col;
            if constexpr (D == 128) {
                key = R0 + 4 * NW * t;
                col = cc * 8;
            } else {
                key = 2 * (R0 + 4 * NW * t) + (cc >> 3);
                col = (cc & 7) * 8;
            }
            kvoff[t] = (uint32_t)(key * (int)p.k_ss + col) * 2u;
            vvoff[t] = (uint32_t)(key * (int)p.v_ss + col) * 2u;
        }
    }
    const int64_t k_slab = ((int64_t)(p.Sk - 1) * p.k_ss + D) * 2;
    const int64_t v_slab = ((int64_t)(p.Sk - 1) * p.v_ss + D) * 2;
    auto dma_tile = [&](auto bufc, int j) {
        constexpr int BUF = decltype(bufc)::value;
        const int64_t kstep = (int64_t)j * BLOCK_N * p.k_ss * 2, vstep = (int64_t)j * BLOCK_N * p.v_ss * 2;
        const srd_t ksrd = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)kp + kstep), 0,
                                                             (int)max((int64_t)0, k_slab - kstep), 0x00020000);
        const srd_t vsrd = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)vp + vstep), 0,
                                                             (int)max((int64_t)0, v_slab - vstep), 0x00020000);
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            const uint32_t kd = smem_base + BUF * BUF_BYTES + (wave + NW * t) * 1024;
            lds_dma16_buf(ksrd, kvoff[t], kd);
            lds_dma16_buf(vsrd, vvoff[t], kd + TILE_BYTES);
        }
    };

    // ---- per-lane LDS read addresses ----------------------------------------------------------------------------------
    uint32_t koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = smem_base + tile_off<D>(r, 2 * ks + h);
    const int g1 = (lane >> 4) & 1;
    const int tq = (lane & 15) >> 2;
    const int tp = lane & 3;
    constexpr int NS2 = (D == 128) ? 1 : 2;
    uint32_t voff[NS2][DB][2];
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2)
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
                voff[s2][db][hi] = smem_base + TILE_BYTES +
                                   tile_off<D>(16 * s2 + 4 * h + tq + 8 * hi, db * 4 + 2 * g1 + (tp >> 1)) + 8 * (tp & 1);

    const float c = p.scale_log2;
    const float thr = (VAR & VAR_DEFER_MAX) ? 8.0f / c : 0.0f;

    // ---- phases (all take the query block by reference) ---------------------------------------------------------------
    auto qk = [&](auto bufc, const QB& X, f32x16 (&s)[2]) {
        constexpr int BUF = decltype(bufc)::value;
        const lds_char* kimg = (const lds_char*)(uintptr_t)(BUF * BUF_BYTES);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) {
            const v8 a = *(const lds_v8*)(kimg + koff[i % KS] + (i / KS) * HALF_TILE);
            s[i / KS] = E::mfma(a, X.qf[i % KS], s[i / KS]);
        }
    };
    auto mask_tile = [&](const QB& X, f32x16 (&s)[2], int key_base) {
        const bool need = (key_base + BLOCK_N > kv_len) || (CAUSAL && key_base + BLOCK_N - 1 > X.q_first) || KMASK;
        if (need) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = key_base + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * h;
                    bool ok = key < kv_len;
                    if (CAUSAL) ok = ok && (key <= X.my_q);
                    if (KMASK) ok = ok && (X.mp[(int64_t)min(key, p.Sk - 1) * p.m_sk] != 0);
                    s[kb][e] = ok ? s[kb][e] : -INFINITY;
                }
        }
    };
    auto max_rescale = [&](QB& X, const f32x16 (&s)[2]) {
        float mx = max3(s[0][0], s[1][0], s[0][1]);
        mx = max3(mx, s[1][1], s[0][2]);
#pragma unroll
        for (int e = 2; e < 16; e += 2) {
            mx = max3(mx, s[1][e], s[0][e + 1]);
            if (e + 2 < 16) mx = max3(mx, s[1][e + 1], s[0][e + 2]);
            else mx = fmaxf(mx, s[1][e + 1]);
        }
        mx = row_pair_max(mx);
        if (__builtin_amdgcn_ballot_w64(mx > X.m + thr) != 0) {
            const float m_new = fmaxf(X.m, mx);
            const float alpha = fast_exp2((X.m - m_new) * c);
            X.m = m_new;
            X.l *= alpha;
            // O lives in the accumulator half of the register file (MFMA C/D); the rare rescale reads each
            // register out, multiplies and writes it back in asm so that hipcc never needs O in VGPRs
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float x = X.o[i][e], t;
                    asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 1\n\tv_accvgpr_write_b32 %0, %1"
                                 : "+a"(x), "=&v"(t)
                                 : "v"(alpha));
                    X.o[i][e] = x;
                }
        }
    };
    auto exp_pack = [&](QB& X, f32x16 (&s)[2], v8 (&ph)[4], v8 (&pl)[4]) {
        const float mc = X.m * c;
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[0][e] = fast_exp2(__builtin_fmaf(s[0][e], c, -mc));
            s[1][e] = fast_exp2(__builtin_fmaf(s[1][e], c, -mc));
            ps0 += s[0][e];
            ps1 += s[1][e];
        }
        X.l += ps0 + ps1;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = s[kb][8 * s2 + e];
                    const T hi = (T)pv;
                    ph[2 * kb + s2][e] = hi;
                    if (SPLITP) pl[2 * kb + s2][e] = (T)(pv - (float)hi);
                }
    };
    auto pv = [&](auto bufc, QB& X, const v8 (&ph)[4], const v8 (&pl)[4]) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                constexpr int S2I = (D == 128) ? 0 : 1;
                const int koffs = BUF * BUF_BYTES + kb * HALF_TILE + ((D == 128) ? s2 * 16 * 256 : 0);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const v4 lo = E::tr_read((const lds_char*)(uintptr_t)(voff[s2 * S2I][db][0] + koffs));
                    const v4 hi4 = E::tr_read((const lds_char*)(uintptr_t)(voff[s2 * S2I][db][1] + koffs));
                    v8 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = lo[e];
                        a[4 + e] = hi4[e];
                    }
                    X.o[db] = E::mfma(a, ph[2 * kb + s2], X.o[db]);
                    if (SPLITP) X.o[db] = E::mfma(a, pl[2 * kb + s2], X.o[db]);
                }
            }
    };

    auto tile = [&](auto bufc, int j) {
        constexpr int BUF = decltype(bufc)::value;
        const int key_base = j * BLOCK_N;
        if (j + 1 < nt) dma_tile(IC<BUF ^ 1>{}, j + 1);
        const bool act_a = key_base < A.kv_end;      // wave-uniform; act_a implies act_b (b's rows are later)
        const bool act_b = key_base < Bq.kv_end;
        if (act_a) {
            f32x16 sa[2], sb[2];
            v8 pha[4], pla[4], phb[4], plb[4];
            qk(bufc, A, sa);                          // matrix only
            mask_tile(A, sa, key_base);
            max_rescale(A, sa);
            qk(bufc, Bq, sb);                         // matrix ...
            exp_pack(A, sa, pha, pla);                // ... beside vector
            asm volatile("" :: "v"(pha[0]), "v"(pha[1]), "v"(pha[2]), "v"(pha[3]), "v"(A.l));   // keep exp(a) here
            mask_tile(Bq, sb, key_base);
            max_rescale(Bq, sb);
            pv(bufc, A, pha, pla);                    // matrix ...
            exp_pack(Bq, sb, phb, plb);               // ... beside vector
            asm volatile("" :: "v"(phb[0]), "v"(phb[1]), "v"(phb[2]), "v"(phb[3]), "v"(Bq.l));
            pv(bufc, Bq, phb, plb);                   // matrix only
        } else if (act_b) {
            f32x16 sb[2];
            v8 phb[4], plb[4];
            qk(bufc, Bq, sb);
            mask_tile(Bq, sb, key_base);
            max_rescale(Bq, sb);
            exp_pack(Bq, sb, phb, plb);
            pv(bufc, Bq, phb, plb);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    };

    if (nt > 0) dma_tile(IC<0>{}, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        asm volatile("" : "+v"(A.qf[ks]));
        asm volatile("" : "+v"(Bq.qf[ks]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int j = 0; j < nt; j += 2) {
        tile(IC<0>{}, j);
        if (j + 1 < nt) tile(IC<1>{}, j + 1);
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------------
    auto store_qb = [&](QB& X) {
        const float l_tot = row_pair_sum(X.l);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (X.my_q < p.Sq) {
            OT* orow = (OT*)p.o + (int64_t)b * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)X.my_q * p.o_ss;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = db * 32 + 8 * g + 4 * h;
                    if constexpr (sizeof(OT) == 4) {
                        f32x4 w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = X.o[db][4 * g + e] * inv;
                        *(f32x4*)(orow + d) = w;
                    } else {
                        v4 w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = (T)(X.o[db][4 * g + e] * inv);
                        *(v4*)(orow + d) = w;
                    }
                }
            if (p.lse && h == 0) {
                const float lse =
                    l_tot > 0.f ? (X.m * c + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f : -INFINITY;
                p.lse[((int64_t)b * p.H + hh) * p.Sq + X.my_q] = lse;
            }
        }
    };
    store_qb(A);
    store_qb(Bq);
}

}  // namespace pfa

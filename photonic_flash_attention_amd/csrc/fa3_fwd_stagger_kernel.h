// fa3_fwd_stagger_kernel.h -- Flash-Attention forward for MI355X, staggered wave halves.
//
// Same math, LDS images and MFMA operand maps as fa3_fwd_kernel.h (read that header first).  What changes
// is WHEN each wave does which phase.  In the plain kernel all eight waves leave the per-tile barrier
// together, so the two waves that share a SIMD (w and w+4) run QK^T together (matrix pipe contended, vector
// ALU idle), then the softmax together (vector ALU contended, matrix pipe idle), then PV together; s_memtime
// stamps (profiles/r01_v13_stamp_phases.txt) put ~45 % of the tile time into that lockstep.
//
// Here waves 4-7 run ONE PHASE BEHIND waves 0-3: between two barriers
//     waves 0-3 (A):   S(j) = K(j) Q^T   ->  softmax(j)       ->  O += P(j) V(j)
//     waves 4-7 (B):   O += P(j-1) V(j-1) ->  S(j) = K(j) Q^T  ->  softmax(j)         (P(j) kept in registers)
// so on every SIMD one wave's vector-only softmax sits beside the other's matrix-only QK^T.  B reads V(j-1)
// while A reads V(j) and V(j+1) lands: the V ring has 3 slots (K ring 2): 80 KiB of LDS at D = 128.
// One barrier per tile, same count in both halves.  The DMA issue (each wave moves its own 1-KiB pieces) is
// placed at different points of the two halves so the eight waves do not queue on the CU's address path.
#pragma once
#include "fa3_fwd_kernel.h"

namespace pfa {

template <typename T, int D, bool CAUSAL, bool SPLITP, bool KMASK, int VAR, typename OT>
__global__ __launch_bounds__(512, 2) void fa3_fwd_stagger_kernel(const FwdParams p) {
    using E = Elem<T>;
    using v8 = typename E::v8;
    using v4 = typename E::v4;
    typedef __attribute__((address_space(3))) v8 lds_v8;
    constexpr int NW = 8;
    constexpr int BLOCK_M = NW * WAVE_M;
    constexpr int KS = D / 16;
    constexpr int DB = D / 32;
    constexpr int TILE_BYTES = BLOCK_N * D * 2;
    constexpr int HALF_TILE = TILE_BYTES / 2;
    constexpr int KRING = 0;                  // K tiles: [2][TILE_BYTES]
    constexpr int VRING = 2 * TILE_BYTES;     // V tiles: [3][TILE_BYTES]
    constexpr int PIECES = TILE_BYTES / 1024;
    constexpr int PPW = PIECES / NW;

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
    const int wave_q0 = q0 + wave * WAVE_M;
    const int my_q = wave_q0 + r;

    int kv_len = p.Sk;
    if (p.seqlens_k) kv_len = min(kv_len, max(p.seqlens_k[b], 0));
    const int kv_end = CAUSAL ? min(kv_len, q0 + BLOCK_M) : kv_len;
    const int wave_kv_end = CAUSAL ? min(kv_len, wave_q0 + WAVE_M) : kv_len;
    const int nt = (kv_end + BLOCK_N - 1) / BLOCK_N;

    const T* __restrict__ qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)hh * p.q_sh;
    const T* __restrict__ kp = (const T*)p.k + (int64_t)b * p.k_sb + (int64_t)(hh / p.kv_group) * p.k_sh;
    const T* __restrict__ vp = (const T*)p.v + (int64_t)b * p.v_sb + (int64_t)(hh / p.kv_group) * p.v_sh;
    const uint8_t* __restrict__ kmp =
        KMASK ? p.mask + (int64_t)b * p.m_sb + (int64_t)hh * p.m_sh + (int64_t)min(my_q, p.Sq - 1) * p.m_sq : nullptr;

    v8 qf[KS];
    {
        const int qrow = min(my_q, p.Sq - 1);
        const T* src = qp + (int64_t)qrow * p.q_ss + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const v8*)(src + 16 * ks);
    }

    // ---- LDS-DMA through buffer descriptors (see fa3_fwd_kernel.h) ------------------------------------------------
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
    auto dma_k = [&](int kslot, int j) {
        const int64_t step = (int64_t)j * BLOCK_N * p.k_ss * 2;
        const srd_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)kp + step), 0,
                                                            (int)max((int64_t)0, k_slab - step), 0x00020000);
#pragma unroll
        for (int t = 0; t < PPW; ++t)
            lds_dma16_buf(srd, kvoff[t], smem_base + KRING + kslot * TILE_BYTES + (wave + NW * t) * 1024);
    };
    auto dma_v = [&](int vslot, int j) {
        const int64_t step = (int64_t)j * BLOCK_N * p.v_ss * 2;
        const srd_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)vp + step), 0,
                                                            (int)max((int64_t)0, v_slab - step), 0x00020000);
#pragma unroll
        for (int t = 0; t < PPW; ++t)
            lds_dma16_buf(srd, vvoff[t], smem_base + VRING + vslot * TILE_BYTES + (wave + NW * t) * 1024);
    };

    // ---- per-lane LDS read addresses (absolute) --------------------------------------------------------------------
    uint32_t koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = smem_base + KRING + tile_off<D>(r, 2 * ks + h);
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
                voff[s2][db][hi] = smem_base + VRING +
                                   tile_off<D>(16 * s2 + 4 * h + tq + 8 * hi, db * 4 + 2 * g1 + (tp >> 1)) + 8 * (tp & 1);

    f32x16 o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    float m_run = -1e30f;
    float l_run = 0.f;
    const float c = p.scale_log2;
    const float thr = (VAR & VAR_DEFER_MAX) ? 8.0f / c : 0.0f;

    // ---- phases ----------------------------------------------------------------------------------------------------------
    // S^T = K Q^T for the 64-key tile in K slot KSLOT (compile time): 2 key blocks x KS MFMAs, reads PF ahead
    auto qk = [&](auto kslotc, f32x16 (&s)[2]) {
        constexpr int KOFF = decltype(kslotc)::value * TILE_BYTES;
        const lds_char* kimg = (const lds_char*)(uintptr_t)KOFF;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
        constexpr int NQK = 2 * KS;
        constexpr int PF = 4;
        v8 afr[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) afr[i] = *(const lds_v8*)(kimg + koff[i % KS] + (i / KS) * HALF_TILE);
#pragma unroll
        for (int i = 0; i < NQK; ++i) {
            s[i / KS] = E::mfma(afr[i % PF], qf[i % KS], s[i / KS]);
            if (i + PF < NQK) afr[i % PF] = *(const lds_v8*)(kimg + koff[(i + PF) % KS] + ((i + PF) / KS) * HALF_TILE);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
        for (int i = 0; i < NQK - PF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);
    };
    // mask + online softmax of one tile; leaves P as packed bf16/f16 fragments (4 k-steps)
    auto softmax = [&](f32x16 (&s)[2], int key_base, v8 (&ph)[4], v8 (&pl)[4]) {
        const bool need_mask = (key_base + BLOCK_N > kv_len) || (CAUSAL && key_base + BLOCK_N - 1 > wave_q0) || KMASK;
        if (need_mask) {
            asm volatile("" ::: "memory");   // real branch, not 64 predicated VALU ops per tile
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = key_base + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * h;
                    bool ok = key < kv_len;
                    if (CAUSAL) ok = ok && (key <= my_q);
                    if (KMASK) ok = ok && (kmp[(int64_t)min(key, p.Sk - 1) * p.m_sk] != 0);
                    s[kb][e] = ok ? s[kb][e] : -INFINITY;
                }
        }
        float mx = max3(s[0][0], s[1][0], s[0][1]);
        mx = max3(mx, s[1][1], s[0][2]);
#pragma unroll
        for (int e = 2; e < 16; e += 2) {
            mx = max3(mx, s[1][e], s[0][e + 1]);
            if (e + 2 < 16) mx = max3(mx, s[1][e + 1], s[0][e + 2]);
            else mx = fmaxf(mx, s[1][e + 1]);
        }
        mx = row_pair_max(mx);
        if (__builtin_amdgcn_ballot_w64(mx > m_run + thr) != 0) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = fast_exp2((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
        }
        const float mc = m_run * c;
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[0][e] = fast_exp2(__builtin_fmaf(s[0][e], c, -mc));
            s[1][e] = fast_exp2(__builtin_fmaf(s[1][e], c, -mc));
            ps0 += s[0][e];
            ps1 += s[1][e];
        }
        l_run += ps0 + ps1;
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
    // O^T += V^T P^T for the tile in V slot `vslot` (run time: ring of 3)
    auto pv = [&](int vslot, const v8 (&ph)[4], const v8 (&pl)[4]) {
        const uint32_t vbase = vslot * TILE_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                constexpr int S2I = (D == 128) ? 0 : 1;
                const int koffs = kb * HALF_TILE + ((D == 128) ? s2 * 16 * 256 : 0);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const v4 lo = E::tr_read((const lds_char*)(uintptr_t)(voff[s2 * S2I][db][0] + vbase + koffs));
                    const v4 hi4 = E::tr_read((const lds_char*)(uintptr_t)(voff[s2 * S2I][db][1] + vbase + koffs));
                    v8 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = lo[e];
                        a[4 + e] = hi4[e];
                    }
                    o[db] = E::mfma(a, ph[2 * kb + s2], o[db]);
                    if (SPLITP) o[db] = E::mfma(a, pl[2 * kb + s2], o[db]);
                }
            }
    };
    auto publish = [&]() {   // this wave's DMA pieces landed + its LDS reads done, then everybody's
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    };

    // ---- prologue ----------------------------------------------------------------------------------------------------------
    if (nt > 0) {
        dma_k(0, 0);
        dma_v(0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    publish();

    v8 ph[4], pl[4];
    const bool lead = wave < NW / 2;    // waves 0-3 lead, waves 4-7 run one phase behind (wave-uniform)

    if (lead) {
        auto iter = [&](auto kslotc, int j, int vslot, int vnext) {
            constexpr int KSLOT = decltype(kslotc)::value;
            if (j + 1 < nt) {
                dma_k(KSLOT ^ 1, j + 1);
                dma_v(vnext, j + 1);
            }
            if (j * BLOCK_N < wave_kv_end) {
                f32x16 s[2];
                qk(kslotc, s);
                softmax(s, j * BLOCK_N, ph, pl);
                pv(vslot, ph, pl);
            }
            publish();
        };
        int vs = 0;
        for (int j = 0; j < nt; j += 2) {
            int vn = vs == 2 ? 0 : vs + 1;
            iter(IC<0>{}, j, vs, vn);
            vs = vn;
            if (j + 1 < nt) {
                vn = vs == 2 ? 0 : vs + 1;
                iter(IC<1>{}, j + 1, vs, vn);
                vs = vn;
            }
        }
    } else {
        auto iter = [&](auto kslotc, int j, int vprev, int vnext) {
            constexpr int KSLOT = decltype(kslotc)::value;
            if (j >= 1 && (j - 1) * BLOCK_N < wave_kv_end) pv(vprev, ph, pl);   // PV of the previous tile
            if (j + 1 < nt) {   // DMA issue after the first matrix phase: not at the same time as waves 0-3
                dma_k(KSLOT ^ 1, j + 1);
                dma_v(vnext, j + 1);
            }
            if (j * BLOCK_N < wave_kv_end) {
                f32x16 s[2];
                qk(kslotc, s);
                softmax(s, j * BLOCK_N, ph, pl);
            }
            publish();
        };
        int vs = 0, vp_ = 2;   // slot of V(j), slot of V(j-1)
        for (int j = 0; j < nt; j += 2) {
            int vn = vs == 2 ? 0 : vs + 1;
            iter(IC<0>{}, j, vp_, vn);
            vp_ = vs;
            vs = vn;
            if (j + 1 < nt) {
                vn = vs == 2 ? 0 : vs + 1;
                iter(IC<1>{}, j + 1, vp_, vn);
                vp_ = vs;
                vs = vn;
            }
        }
        if (nt >= 1 && (nt - 1) * BLOCK_N < wave_kv_end) pv(vp_, ph, pl);         // drain: PV of the last tile
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------------
    const float l_tot = row_pair_sum(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (my_q < p.Sq) {
        OT* orow = (OT*)p.o + (int64_t)b * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)my_q * p.o_ss;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * h;
                if constexpr (sizeof(OT) == 4) {
                    f32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = o[db][4 * g + e] * inv;
                    *(f32x4*)(orow + d) = w;
                } else {
                    v4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = (T)(o[db][4 * g + e] * inv);
                    *(v4*)(orow + d) = w;
                }
            }
        if (p.lse && h == 0) {
            const float lse = l_tot > 0.f ? (m_run * c + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f : -INFINITY;
            p.lse[((int64_t)b * p.H + hh) * p.Sq + my_q] = lse;
        }
    }
}

}  // namespace pfa

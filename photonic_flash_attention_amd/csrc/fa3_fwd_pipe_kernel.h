// fa3_fwd_pipe_kernel.h -- software-pipelined Flash-Attention forward for MI355X (gfx950).
//
// Same math, data layout, LDS images and MFMA operand maps as fa3_fwd_kernel.h (read that header first);
// what changes is the SCHEDULE.  The plain kernel runs QK^T -> softmax -> PV per 64-key tile, so both waves
// of a SIMD sit in their VALU-only softmax stretch together while the matrix pipe idles.  Here every wave
// works on 32-key HALF tiles and is one half ahead with the scores:
//
//   half-step t:   region 1:  S(t+1) = K(t+1) Q^T  [8 MFMA]   beside   P(t) = exp2(c S(t) - c m), row sums, bf16 pack
//                  (rare) mask S(t+1)
//                  region 2:  O^T += V(t)^T P(t)^T  [8 MFMA]  beside   row max of S(t+1)
//                  (rare) running max grew past the headroom -> rescale O, l
//
// so each scheduling region has matrix work AND vector work of the same wave to interleave, and the max /
// rescale decision for a half is taken one half-step BEFORE its exponentials are needed.
//
// K runs one tile ahead of V: per 64-key tile j the workgroup has two barriers,
//   S_j: V(j) landed, V ring slot of V(j-1) free  -> issue DMA of V(j+1)
//   M_j: K(j+1) landed, K ring slot of K(j) free  -> issue DMA of K(j+2)
// each DMA gets a full tile of flight time, and the waits are COUNTED (s_waitcnt vmcnt(PPW) leaves the
// younger tile's pieces in flight across the barrier; the DMA is inline asm, invisible to hipcc's waitcnt pass).
#pragma once
#include "fa3_fwd_kernel.h"

namespace pfa {

template <typename T, int D, bool CAUSAL, bool SPLITP, bool KMASK, int VAR, typename OT>
__global__ __launch_bounds__(512, 2) void fa3_fwd_pipe_kernel(const FwdParams p) {
    using E = Elem<T>;
    using v8 = typename E::v8;
    using v4 = typename E::v4;
    typedef __attribute__((address_space(3))) v8 lds_v8;
    constexpr int NW = 8;
    constexpr int BLOCK_M = NW * WAVE_M;
    constexpr int KS = D / 16;                // k-steps of one QK^T half (32 keys x D)
    constexpr int DB = D / 32;                // 32-wide d blocks of the PV product
    constexpr int TILE_BYTES = BLOCK_N * D * 2;
    constexpr int HALF_TILE = TILE_BYTES / 2; // 32 keys
    constexpr int KRING = 0;                  // K tiles: [2][TILE_BYTES]
    constexpr int VRING = 2 * TILE_BYTES;     // V tiles: [2][TILE_BYTES]
    constexpr int PIECES = TILE_BYTES / 1024; // 1-KiB DMA pieces per tile image
    constexpr int PPW = PIECES / NW;          // pieces per wave per image: 2 (D=128) or 1 (D=64)
    static_assert(PPW == 1 || PPW == 2, "unexpected DMA piece count");

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
    const int qblk = CAUSAL ? (p.nqblk - 1 - qrank) : qrank;   // heaviest causal blocks first
    const int b = bh / p.H;
    const int hh = bh - b * p.H;

    const int q0 = qblk * BLOCK_M;
    const int wave_q0 = q0 + wave * WAVE_M;
    const int my_q = wave_q0 + r;

    int kv_len = p.Sk;
    if (p.seqlens_k) kv_len = min(kv_len, max(p.seqlens_k[b], 0));
    const int kv_end = CAUSAL ? min(kv_len, q0 + BLOCK_M) : kv_len;
    const int wave_kv_end = CAUSAL ? min(kv_len, wave_q0 + WAVE_M) : kv_len;   // keys this wave needs
    const int nt = (kv_end + BLOCK_N - 1) / BLOCK_N;

    const T* __restrict__ qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)hh * p.q_sh;
    const T* __restrict__ kp = (const T*)p.k + (int64_t)b * p.k_sb + (int64_t)(hh / p.kv_group) * p.k_sh;
    const T* __restrict__ vp = (const T*)p.v + (int64_t)b * p.v_sb + (int64_t)(hh / p.kv_group) * p.v_sh;
    const uint8_t* __restrict__ kmp =
        KMASK ? p.mask + (int64_t)b * p.m_sb + (int64_t)hh * p.m_sh + (int64_t)min(my_q, p.Sq - 1) * p.m_sq : nullptr;

    // Q fragments (B operand of S^T = K Q^T)
    v8 qf[KS];
    {
        const int qrow = min(my_q, p.Sq - 1);
        const T* src = qp + (int64_t)qrow * p.q_ss + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const v8*)(src + 16 * ks);
    }

    // ---- LDS-DMA source mapping (swizzle on the source chunk, linear LDS destination) ----------------------
    int dma_key[PPW];
    int dma_col;
    {
        const int R0 = 4 * wave + (lane >> 4);
        const int sw = ((R0 & 3) << 2) | ((R0 >> 2) & 3);
        const int cc = (lane & 15) ^ sw;
        if constexpr (D == 128) {
            dma_col = cc * 8;
#pragma unroll
            for (int t = 0; t < PPW; ++t) dma_key[t] = R0 + 4 * NW * t;
        } else {
            dma_col = (cc & 7) * 8;
#pragma unroll
            for (int t = 0; t < PPW; ++t) dma_key[t] = 2 * (R0 + 4 * NW * t) + (cc >> 3);
        }
    }
    auto dma_k = [&](auto bufc, int j) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            const int key = min(j * BLOCK_N + dma_key[t], p.Sk - 1);
            lds_dma16(kp + (int64_t)key * p.k_ss + dma_col, smem_base + KRING + BUF * TILE_BYTES + (wave + NW * t) * 1024);
        }
    };
    auto dma_v = [&](auto bufc, int j) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            const int key = min(j * BLOCK_N + dma_key[t], p.Sk - 1);
            lds_dma16(vp + (int64_t)key * p.v_ss + dma_col, smem_base + VRING + BUF * TILE_BYTES + (wave + NW * t) * 1024);
        }
    };

    // ---- per-lane LDS read offsets (see fa3_fwd_kernel.h) ------------------------------------------------------
    uint32_t koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = smem_base + tile_off<D>(r, 2 * ks + h);   // absolute LDS address
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
                voff[s2][db][hi] = smem_base + tile_off<D>(16 * s2 + 4 * h + tq + 8 * hi, db * 4 + 2 * g1 + (tp >> 1)) + 8 * (tp & 1);

    f32x16 o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    float m_run = -1e30f;
    float l_run = 0.f;
    const float c = p.scale_log2;
    const float thr = (VAR & VAR_DEFER_MAX) ? 8.0f / c : 0.0f;

    // ---- pieces of a half-step -------------------------------------------------------------------------------------
    // S(half) = K(half) Q^T : KS MFMAs, A fragments PF deep ahead
    constexpr int PF = (KS >= 4) ? 3 : KS;
    auto qk_half = [&](auto kbufc, auto kbc, f32x16& s) {
        constexpr int OFF = KRING + decltype(kbufc)::value * TILE_BYTES + decltype(kbc)::value * HALF_TILE;
        const lds_char* kimg = (const lds_char*)(uintptr_t)OFF;   // koff[] carries the LDS base
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        v8 afr[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) afr[i] = *(const lds_v8*)(kimg + koff[i]);
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            s = E::mfma(afr[i % PF], qf[i], s);
            if (i + PF < KS) afr[i % PF] = *(const lds_v8*)(kimg + koff[i + PF]);
        }
    };
    // P = exp2(c S - c m) in place, row-sum share, bf16 pack (m_run already covers this half)
    auto exp_half = [&](f32x16& s, v8 (&ph)[2], v8 (&pl)[2]) {
        const float mc = m_run * c;
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            s[e] = fast_exp2(__builtin_fmaf(s[e], c, -mc));
            s[e + 1] = fast_exp2(__builtin_fmaf(s[e + 1], c, -mc));
            ps0 += s[e];
            ps1 += s[e + 1];
        }
        l_run += ps0 + ps1;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pv = s[8 * s2 + e];
                const T hi = (T)pv;
                ph[s2][e] = hi;
                if (SPLITP) pl[s2][e] = (T)(pv - (float)hi);
            }
    };
    // O^T += V(half)^T P^T : 2 k-steps x DB MFMAs (x2 with split P)
    auto pv_half = [&](auto vbufc, auto kbc, const v8 (&ph)[2], const v8 (&pl)[2]) {
        constexpr int OFF = VRING + decltype(vbufc)::value * TILE_BYTES + decltype(kbc)::value * HALF_TILE;
        const lds_char* vimg = (const lds_char*)(uintptr_t)OFF;   // voff[] carries the LDS base
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            constexpr int S2I = (D == 128) ? 0 : 1;
            const int koffs = (D == 128) ? s2 * 16 * 256 : 0;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const v4 lo = E::tr_read(vimg + voff[s2 * S2I][db][0] + koffs);
                const v4 hi4 = E::tr_read(vimg + voff[s2 * S2I][db][1] + koffs);
                v8 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = lo[e];
                    a[4 + e] = hi4[e];
                }
                o[db] = E::mfma(a, ph[s2], o[db]);
                if (SPLITP) o[db] = E::mfma(a, pl[s2], o[db]);
            }
        }
    };
    // mask one half (rare, wave-uniform branch)
    auto mask_half = [&](f32x16& s, int key_base) {
        const bool need = (key_base + 32 > kv_len) || (CAUSAL && key_base + 31 > wave_q0) || KMASK;
        if (need) {
            asm volatile("" ::: "memory");   // keep it a real branch (hipcc would if-convert it onto every half)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = key_base + (e & 3) + 8 * (e >> 2) + 4 * h;
                bool ok = key < kv_len;
                if (CAUSAL) ok = ok && (key <= my_q);
                if (KMASK) ok = ok && (kmp[(int64_t)min(key, p.Sk - 1) * p.m_sk] != 0);
                s[e] = ok ? s[e] : -INFINITY;
            }
        }
    };
    // row max of one half (both lanes of the row)
    auto max_half = [&](const f32x16& s) -> float {
        float mx = max3(s[0], s[1], s[2]);
#pragma unroll
        for (int e = 3; e + 1 < 16; e += 2) mx = max3(mx, s[e], s[e + 1]);
        mx = fmaxf(mx, s[15]);
        return row_pair_max(mx);
    };
    // adopt a new reference max when some row outgrew the headroom (rare after the first halves)
    auto rescale = [&](float mx) {
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
    };

    // half-step t: cur = half t (scores ready, max decided), nxt = half t+1
    //   KBUFN/KBN: ring slot / key block of half t+1;  VBUFC/KBC: ring slot / key block of half t
    auto half_step = [&](auto kbufn, auto kbn, auto vbufc, auto kbc, f32x16& s_cur, f32x16& s_nxt, int t) {
        const bool act_cur = 32 * t < wave_kv_end;
        const bool act_nxt = 32 * (t + 1) < wave_kv_end;
        v8 ph[2], pl[2];
        if (act_nxt) {
            qk_half(kbufn, kbn, s_nxt);                       // region 1: matrix
            exp_half(s_cur, ph, pl);                          // region 1: vector
            // pin P here: hipcc otherwise SINKS the exponentials below the mask branch, next to the PV MFMAs
            asm volatile("" :: "v"(ph[0]), "v"(ph[1]), "v"(l_run));
            if (SPLITP) asm volatile("" :: "v"(pl[0]), "v"(pl[1]));
            if (VAR & VAR_SCHED) {
                __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
                for (int i = 0; i < KS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i + PF < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 40 / KS, 0);   // fma / add / cvt
                    __builtin_amdgcn_sched_group_barrier(0x400, 16 / KS, 0);   // v_exp
                }
            }
            mask_half(s_nxt, 32 * (t + 1));
            pv_half(vbufc, kbc, ph, pl);                      // region 2: matrix
            const float mx = max_half(s_nxt);                 // region 2: vector
            rescale(mx);
        } else if (act_cur) {
            exp_half(s_cur, ph, pl);
            pv_half(vbufc, kbc, ph, pl);
        }
    };

    auto wait_dma = [&](bool younger_in_flight) {
        // retire this wave's DMA pieces of the tile being published; a younger tile's PPW pieces may stay in flight
        if (younger_in_flight) {
            if constexpr (PPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS reads of the slot being recycled are done
        __builtin_amdgcn_s_barrier();
    };

    f32x16 sA, sB;   // scores of even / odd halves

    // one 64-key tile j whose K/V live in ring slot BUF = j & 1
    auto tile_iter = [&](auto bufc, int j) {
        constexpr int BUF = decltype(bufc)::value;
        // S_j: V(j) landed everywhere; V slot BUF^1 (V(j-1)) is free
        wait_dma(j >= 1 && j + 1 < nt);                        // K(j+1) may still be in flight
        if (j + 1 < nt) dma_v(IC<BUF ^ 1>{}, j + 1);
        half_step(IC<BUF>{}, IC<1>{}, IC<BUF>{}, IC<0>{}, sA, sB, 2 * j);          // S(j,1) | P,PV (j,0)
        // M_j: K(j+1) landed everywhere; K slot BUF (K(j)) is free
        if (j + 1 < nt) {
            wait_dma(true);                                    // V(j+1) may still be in flight
            if (j + 2 < nt) dma_k(IC<BUF>{}, j + 2);
        }
        half_step(IC<BUF ^ 1>{}, IC<0>{}, IC<BUF>{}, IC<1>{}, sB, sA, 2 * j + 1);   // S(j+1,0) | P,PV (j,1)
    };

    // ---- prologue: K(0), V(0), K(1) in flight; scores, mask, max and scale of half 0 ---------------------------
    if (nt > 0) {
        dma_k(IC<0>{}, 0);
        dma_v(IC<0>{}, 0);
        if (nt > 1) dma_k(IC<1>{}, 1);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));   // Q landed before the loop (see fa3_fwd_kernel.h)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nt > 0 && 0 < wave_kv_end) {
        qk_half(IC<0>{}, IC<0>{}, sA);
        mask_half(sA, 0);
        rescale(max_half(sA));
    }
    for (int j = 0; j < nt; j += 2) {
        tile_iter(IC<0>{}, j);
        if (j + 1 < nt) tile_iter(IC<1>{}, j + 1);
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------
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

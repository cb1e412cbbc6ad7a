// fa3_fwd_w4_kernel.h -- Flash-Attention forward, D = 128: 4 waves x 64 query rows, ONE wave per SIMD (512 registers).
//
// Same math, HBM/LDS images, MFMA operand maps and LDS-DMA staging as fa3_fwd_kernel.h (read that first).  What changes
// is the ratio of LDS traffic to matrix work and who hides what:
//
//  * a wave owns TWO 32-row query blocks (a, b) and feeds BOTH from every K / V^T fragment it reads: half the LDS bytes
//    (and ds_read instructions) per MFMA of the 8-wave x 32-row kernel, which is power- and issue-limited by exactly that
//    (profiles/r01_ablation_lds_vs_mfma.txt);
//  * there is no partner wave on the SIMD, so the softmax is software-pipelined inside the wave's own stream and
//    hand-placed between the MFMAs (sched_barrier after every step; at most ~5 vector issues per MFMA):
//
//        iteration j :  phase A   S(j+1) = K(j+1) Q^T   (32 MFMA)  ||  finish softmax(j) -> P(j)            || K reads, V DMA
//                       phase B   O     += V(j)^T P(j)  (32 MFMA)  ||  start softmax(j+1) (max, exps of kb 0) || V^T reads, K DMA
//
//    S is double-buffered in registers (2 x 64 VGPRs); O (128) and Q (64) live in the accumulator half of the register
//    file and are touched only by inline-asm MFMAs ("a" constraints), so hipcc never copies them through VGPRs.
//    K runs one tile ahead of V in the LDS rings (2 slots each, 64 KiB); one barrier per 64-key tile.
//    The O rescale decided in phase B (rare: defer-max) is applied after that phase's MFMAs.
//  * tiles that need element masks (causal diagonal, key tail) and the last tile take the same steps without the
//    interleave (`plain`), so the arithmetic is identical on both paths.
#pragma once
#include "fa3_fwd_kernel.h"

namespace pfa {

template <typename T> struct W4Asm;
template <> struct W4Asm<__bf16> {
    template <typename V8> static __device__ __forceinline__ void s0(f32x16& s, V8 k, V8 q) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q));
    }
    template <typename V8> static __device__ __forceinline__ void s(f32x16& s, V8 k, V8 q) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q));
    }
    template <typename V8> static __device__ __forceinline__ void o(f32x16& o, V8 v, u32x4 p) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(p));
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        uint32_t r;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
};
template <> struct W4Asm<_Float16> {
    template <typename V8> static __device__ __forceinline__ void s0(f32x16& s, V8 k, V8 q) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q));
    }
    template <typename V8> static __device__ __forceinline__ void s(f32x16& s, V8 k, V8 q) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q));
    }
    template <typename V8> static __device__ __forceinline__ void o(f32x16& o, V8 v, u32x4 p) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(p));
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        const h2 t = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(uint32_t, t);
    }
};

template <int N, int I = 0, typename F>
__device__ __forceinline__ void w4_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        w4_for<N, I + 1>(f);
    }
}

template <typename T, bool CAUSAL, typename OT>
__global__ __launch_bounds__(256, 1) void fa3_fwd_w4_kernel(const FwdParams p) {
    using E = Elem<T>;
    using M = W4Asm<T>;
    using v8 = typename E::v8;
    using v4 = typename E::v4;
    typedef __attribute__((address_space(3))) v8 lds_v8;
    constexpr int D = 128, NW = 4, BLOCK_M = 256, KS = D / 16, DB = D / 32;
    constexpr int TILE_BYTES = BLOCK_N * D * 2, HALF_TILE = TILE_BYTES / 2;
    constexpr int K_BASE = 0, V_BASE = 2 * TILE_BYTES;
    constexpr int PPW = (TILE_BYTES / 1024) / NW;          // 4 DMA pieces per wave per image
    constexpr int PFK = 2, PFV = 2;                        // operand-fragment rings (steps ahead)

    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int BH = p.B * p.H, n = blockIdx.x, qrank = n / BH, bh = n - qrank * BH;
    const int qblk = CAUSAL ? (p.nqblk - 1 - qrank) : qrank;
    const int b = bh / p.H, hh = bh - b * p.H;
    const int q0 = qblk * BLOCK_M, wave_q0 = q0 + wave * 64;

    int kv_len = p.Sk;
    if (p.seqlens_k) kv_len = min(kv_len, max(p.seqlens_k[b], 0));
    const int kv_end = CAUSAL ? min(kv_len, q0 + BLOCK_M) : kv_len;
    const int nt = (kv_end + BLOCK_N - 1) / BLOCK_N;                                   // tiles the workgroup streams
    const int wave_kv_end = CAUSAL ? min(kv_len, wave_q0 + 64) : kv_len;
    const int wnt = (wave_kv_end + BLOCK_N - 1) / BLOCK_N;                             // tiles this wave computes

    const T* qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)hh * p.q_sh;
    const char* kp = (const char*)((const T*)p.k + (int64_t)b * p.k_sb + (int64_t)hh * p.k_sh);
    const char* vp = (const char*)((const T*)p.v + (int64_t)b * p.v_sb + (int64_t)hh * p.v_sh);

    // ---- query-block state -----------------------------------------------------------------------------------------
    struct QB {
        v8 qf[KS];                          // accumulator file ("a")
        f32x16 o[DB];                       // accumulator file ("+a")
        float m, l, mc, m_thr, alpha;       // alpha: pending O rescale (1 = none)
        float ps0;                          // row-sum share of key block 0 of the tile whose softmax has started
        int my_q;
    };
    QB A, Bq;
    const float c = p.scale_log2;
    const float thr = 8.0f / c;
    auto init_qb = [&](QB& X, int first) {
        X.my_q = first + r;
        const T* src = qp + (int64_t)min(X.my_q, p.Sq - 1) * p.q_ss + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) X.qf[ks] = *(const v8*)(src + 16 * ks);
#pragma unroll
        for (int i = 0; i < DB; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) X.o[i][e] = 0.f;
            asm volatile("" : "+a"(X.o[i]));
        }
        X.m = -1e30f;
        X.l = 0.f;
        X.mc = -1e30f * c;
        X.m_thr = -1e30f;
        X.alpha = 1.0f;
        X.ps0 = 0.f;
    };
    init_qb(A, wave_q0);
    init_qb(Bq, wave_q0 + 32);

    // ---- LDS-DMA: K(j) -> K slot j&1, V(j) -> V slot j&1; one buffer descriptor per piece (SALU) ------------------------
    uint32_t koffd, voffd;          // per-lane source byte offsets of piece 0 (pieces step by 16 rows: uniform, in the base)
    {
        const int R0 = 4 * wave + (lane >> 4);
        const int sw = ((R0 & 3) << 2) | ((R0 >> 2) & 3);
        const int cc = (lane & 15) ^ sw;
        koffd = (uint32_t)(R0 * (int)p.k_ss + cc * 8) * 2u;
        voffd = (uint32_t)(R0 * (int)p.v_ss + cc * 8) * 2u;
    }
    const uint32_t k_slab = (uint32_t)(((int64_t)(p.Sk - 1) * p.k_ss + D) * 2), v_slab = (uint32_t)(((int64_t)(p.Sk - 1) * p.v_ss + D) * 2);
    auto srd_at = [](const char* base, uint32_t off, uint32_t slab) {
        const uint64_t a = (uint64_t)(uintptr_t)(base + off);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0,
                                                 (int)__builtin_amdgcn_readfirstlane(slab > off ? slab - off : 0u), 0x00020000);
    };
    auto dma_k = [&](int j, int t) {      // piece t of K(j); rows past Sk (and whole tiles past the last) arrive as zeros
        const uint32_t off = ((uint32_t)j * (uint32_t)(BLOCK_N * 2) + (uint32_t)(32 * t)) * (uint32_t)p.k_ss;
        lds_dma16_buf(srd_at(kp, j < nt ? off : k_slab, k_slab), koffd, smem_base + K_BASE + (j & 1) * TILE_BYTES + (wave + NW * t) * 1024);
    };
    auto dma_v = [&](int j, int t) {
        const uint32_t off = ((uint32_t)j * (uint32_t)(BLOCK_N * 2) + (uint32_t)(32 * t)) * (uint32_t)p.v_ss;
        lds_dma16_buf(srd_at(vp, j < nt ? off : v_slab, v_slab), voffd, smem_base + V_BASE + (j & 1) * TILE_BYTES + (wave + NW * t) * 1024);
    };

    // ---- per-lane LDS read addresses (opaque: see VAR_DIET in fa3_fwd_kernel.h) ---------------------------------------------
    uint32_t koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        koff[ks] = smem_base + K_BASE + tile_off<D>(r, 2 * ks + h);
        asm volatile("" : "+v"(koff[ks]));
    }
    const int g1 = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    uint32_t voff[DB][2];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            voff[db][hi] = smem_base + V_BASE + tile_off<D>(4 * h + tq + 8 * hi, db * 4 + 2 * g1 + (tp >> 1)) + 8 * (tp & 1);
            asm volatile("" : "+v"(voff[db][hi]));
        }

    // ---- the steps of one tile ---------------------------------------------------------------------------------------------
    // A tile's work is cut into 16 QK^T steps, 16 softmax-finish pieces, 16 PV steps and 16 softmax-start pieces; the
    // interleaved iteration pairs them one to one, the plain iteration runs them back to back.
    v8 kfr[PFK];
    auto k_read = [&](int slot, int i) { return *(const lds_v8*)(uintptr_t)(koff[i % KS] + slot * TILE_BYTES + (i / KS) * HALF_TILE); };
    auto qk_begin = [&](int slot) {
#pragma unroll
        for (int i = 0; i < PFK; ++i) kfr[i] = k_read(slot, i);
    };
    // step i: key block kb = i / 8, k-step ks = i % 8; one K fragment, two MFMAs
    auto qk_step = [&](auto ic, int slot, f32x16 (&sa)[2], f32x16 (&sb)[2]) {
        constexpr int i = decltype(ic)::value, kb = i / KS, ks = i % KS;
        if constexpr (ks == 0) {
            M::s0(sa[kb], kfr[i % PFK], A.qf[ks]);
            M::s0(sb[kb], kfr[i % PFK], Bq.qf[ks]);
        } else {
            M::s(sa[kb], kfr[i % PFK], A.qf[ks]);
            M::s(sb[kb], kfr[i % PFK], Bq.qf[ks]);
        }
        if constexpr (i + PFK < 2 * KS) kfr[i % PFK] = k_read(slot, i + PFK);
    };
    // softmax-finish piece i: block (i < 8 ? a : b), element pair q = i % 8 of key block 1 gets its exponentials; the
    // pair's key-block-0 P dword and the previous pair's key-block-1 sum + P dword are produced alongside
    float ps1 = 0.f;
    auto sm2_one = [&](QB& X, int q, f32x16 (&s)[2], uint32_t (&pd)[16]) {
        const int e0 = 2 * q, e1 = e0 + 1;
        s[1][e0] = fast_exp2(__builtin_fmaf(s[1][e0], c, -X.mc));
        s[1][e1] = fast_exp2(__builtin_fmaf(s[1][e1], c, -X.mc));
        pd[(e0 >> 3) * 4 + ((e0 & 7) >> 1)] = M::pack2(s[0][e0], s[0][e1]);
        if (q > 0) {
            ps1 += s[1][e0 - 2];
            asm volatile("" : "+v"(ps1));
            ps1 += s[1][e0 - 1];
            pd[(2 + ((e0 - 2) >> 3)) * 4 + (((e0 - 2) & 7) >> 1)] = M::pack2(s[1][e0 - 2], s[1][e0 - 1]);
        }
    };
    auto sm2_end = [&](QB& X, f32x16 (&s)[2], uint32_t (&pd)[16]) {
        ps1 += s[1][14];
        asm volatile("" : "+v"(ps1));
        ps1 += s[1][15];
        pd[15] = M::pack2(s[1][14], s[1][15]);
        X.l += X.ps0 + ps1;
        ps1 = 0.f;
    };
    auto sm2_piece = [&](auto ic, f32x16 (&ca)[2], f32x16 (&cb)[2], uint32_t (&pa)[16], uint32_t (&pb)[16]) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < 8) {
            sm2_one(A, i, ca, pa);
        } else {
            if constexpr (i == 8) sm2_end(A, ca, pa);
            sm2_one(Bq, i - 8, cb, pb);
        }
    };
    // PV step idx: k-step (kb, s2) = idx / 4, d block db = idx % 4; one V^T fragment (two transposed reads), two MFMAs
    v4 vlo[PFV], vhi[PFV];
    auto v_read = [&](int slot, int idx, v4& lo, v4& hi4) {
        const uint32_t ko = slot * TILE_BYTES + (idx / 8) * HALF_TILE + ((idx / 4) & 1) * 16 * 256;
        lo = E::tr_read((const lds_char*)(uintptr_t)(voff[idx % 4][0] + ko));
        hi4 = E::tr_read((const lds_char*)(uintptr_t)(voff[idx % 4][1] + ko));
    };
    auto pv_begin = [&](int slot) {
#pragma unroll
        for (int i = 0; i < PFV; ++i) v_read(slot, i, vlo[i], vhi[i]);
    };
    auto pv_step = [&](auto ic, int slot, const uint32_t (&pa)[16], const uint32_t (&pb)[16]) {
        constexpr int idx = decltype(ic)::value, f = idx / 4, db = idx % 4;
        v8 a;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = vlo[idx % PFV][e];
            a[4 + e] = vhi[idx % PFV][e];
        }
        const u32x4 fa = {pa[4 * f], pa[4 * f + 1], pa[4 * f + 2], pa[4 * f + 3]};
        const u32x4 fb = {pb[4 * f], pb[4 * f + 1], pb[4 * f + 2], pb[4 * f + 3]};
        M::o(A.o[db], a, fa);
        M::o(Bq.o[db], a, fb);
        if constexpr (idx + PFV < 16) v_read(slot, idx + PFV, vlo[idx % PFV], vhi[idx % PFV]);
    };
    // softmax-start piece idx: block (idx < 8 ? a : b), local step t = idx % 8: t 0-1 row max and the (branch-free,
    // per-row) defer-max update, t 2-7 the exponentials and sums of key block 0
    float mx_carry = 0.f;
    auto sm1_one = [&](QB& X, int t, f32x16 (&s)[2]) {
        if (t == 0) {
            mx_carry = max16_first(s[0]);
        } else if (t == 1) {
            float mx = max16_next(mx_carry, s[1]);
            mx = row_pair_max_asm(mx);
            const bool grow = mx > X.m_thr;                  // this row's max outgrew the headroom
            const float m_new = grow ? fmaxf(X.m, mx) : X.m;
            const float al = grow ? fast_exp2((X.m - m_new) * c) : 1.0f;
            X.m = m_new;
            X.m_thr = grow ? m_new + thr : X.m_thr;
            X.mc = m_new * c;
            X.l *= al;
            X.alpha = al;                                    // O *= alpha after the PV MFMAs in flight beside this
            X.ps0 = 0.f;
        } else {
            const int lo = (t - 2) * 3, hi = (t == 7) ? 16 : lo + 3;
#pragma unroll
            for (int e = lo; e < hi; ++e) {
                s[0][e] = fast_exp2(__builtin_fmaf(s[0][e], c, -X.mc));
                X.ps0 += s[0][e];
                asm volatile("" : "+v"(X.ps0));
            }
        }
    };
    auto sm1_piece = [&](auto ic, f32x16 (&na)[2], f32x16 (&nb)[2]) {
        constexpr int idx = decltype(ic)::value;
        if constexpr (idx < 8) sm1_one(A, idx, na);
        else sm1_one(Bq, idx - 8, nb);
    };
    auto apply_mask = [&](QB& X, f32x16 (&s)[2], int key_base) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = key_base + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * h;
                bool ok = key < kv_len;
                if (CAUSAL) ok = ok && (key <= X.my_q);
                s[kb][e] = ok ? s[kb][e] : -INFINITY;
            }
    };
    auto needs_mask = [&](int key_base) { return (key_base + BLOCK_N > kv_len) || (CAUSAL && key_base + BLOCK_N - 1 > wave_q0); };
    // the deferred O rescale (rare): O never leaves the accumulator file
    auto rescale = [&](QB& X) {
        if (__builtin_amdgcn_ballot_w64(X.alpha != 1.0f) != 0) {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // last MFMA -> v_accvgpr_read
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float x = X.o[i][e], t;
                    asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 1\n\tv_accvgpr_write_b32 %0, %1"
                                 : "+a"(x), "=&v"(t)
                                 : "v"(X.alpha));
                    X.o[i][e] = x;
                }
            asm volatile("s_nop 7" ::: "memory");
        }
        X.alpha = 1.0f;
    };
    auto publish = [&]() {          // this wave's DMA pieces have landed; everyone is done with the slots about to be refilled
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    };

    // ---- iterations ---------------------------------------------------------------------------------------------------------
    // interleaved: tile j's finish + PV beside tile j+1's QK^T + start (no masks on tile j+1)
    auto steady = [&](auto pc, int j, f32x16 (&ca)[2], f32x16 (&cb)[2], f32x16 (&na)[2], f32x16 (&nb)[2]) {
        constexpr int P = decltype(pc)::value;
        uint32_t pa[16], pb[16];
        qk_begin(P ^ 1);
        w4_for<16>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            qk_step(ic, P ^ 1, na, nb);
            if constexpr (i % 4 == 1) dma_v(j + 1, i / 4);
            sm2_piece(ic, ca, cb, pa, pb);
            __builtin_amdgcn_sched_barrier(0);
        });
        sm2_end(Bq, cb, pb);
        pv_begin(P);
        __builtin_amdgcn_sched_barrier(0);
        w4_for<16>([&](auto ic) {
            constexpr int idx = decltype(ic)::value;
            pv_step(ic, P, pa, pb);
            if constexpr (idx % 3 == 1 && idx < 12) dma_k(j + 2, idx / 3);
            sm1_piece(ic, na, nb);
            __builtin_amdgcn_sched_barrier(0);
        });
        rescale(A);
        rescale(Bq);
    };
    // plain: the same steps back to back; tile j+1 (if this wave has one) may need masks
    auto plain = [&](auto pc, int j, bool has_next, f32x16 (&ca)[2], f32x16 (&cb)[2], f32x16 (&na)[2], f32x16 (&nb)[2]) {
        constexpr int P = decltype(pc)::value;
        uint32_t pa[16], pb[16];
        if (j + 1 < nt) {
#pragma unroll
            for (int t = 0; t < PPW; ++t) dma_v(j + 1, t);
        }
        if (j + 2 < nt) {
#pragma unroll
            for (int t = 0; t < PPW; ++t) dma_k(j + 2, t);
        }
        if (has_next) {
            qk_begin(P ^ 1);
            w4_for<16>([&](auto ic) { qk_step(ic, P ^ 1, na, nb); });
        }
        w4_for<16>([&](auto ic) { sm2_piece(ic, ca, cb, pa, pb); });
        sm2_end(Bq, cb, pb);
        pv_begin(P);
        w4_for<16>([&](auto ic) { pv_step(ic, P, pa, pb); });
        if (has_next) {
            if (needs_mask((j + 1) * BLOCK_N)) {
                asm volatile("" ::: "memory");
                apply_mask(A, na, (j + 1) * BLOCK_N);
                apply_mask(Bq, nb, (j + 1) * BLOCK_N);
            }
            w4_for<16>([&](auto ic) { sm1_piece(ic, na, nb); });
            rescale(A);
            rescale(Bq);
        }
    };
    auto iter = [&](auto pc, int j, f32x16 (&ca)[2], f32x16 (&cb)[2], f32x16 (&na)[2], f32x16 (&nb)[2]) {
        if (j < wnt) {
            const bool has_next = j + 1 < wnt;
            if (has_next && !needs_mask((j + 1) * BLOCK_N)) steady(pc, j, ca, cb, na, nb);
            else plain(pc, j, has_next, ca, cb, na, nb);
        } else {                     // this wave is past its last tile: keep feeding the rings, keep the barriers
            if (j + 1 < nt) {
#pragma unroll
                for (int t = 0; t < PPW; ++t) dma_v(j + 1, t);
            }
            if (j + 2 < nt) {
#pragma unroll
                for (int t = 0; t < PPW; ++t) dma_k(j + 2, t);
            }
        }
        publish();
    };

    // ---- prologue -----------------------------------------------------------------------------------------------------------
    f32x16 S0a[2], S0b[2], S1a[2], S1b[2];      // S(j) of even / odd tiles
    if (nt > 0) {
#pragma unroll
        for (int t = 0; t < PPW; ++t) dma_k(0, t);
#pragma unroll
        for (int t = 0; t < PPW; ++t) dma_v(0, t);
        if (nt > 1) {
#pragma unroll
            for (int t = 0; t < PPW; ++t) dma_k(1, t);
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {           // Q into the accumulator file, once
        asm volatile("" : "+a"(A.qf[ks]));
        asm volatile("" : "+a"(Bq.qf[ks]));
    }
    publish();
    if (wnt > 0) {
        qk_begin(0);
        w4_for<16>([&](auto ic) { qk_step(ic, 0, S0a, S0b); });
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");              // last MFMA -> first VALU read of S
        if (needs_mask(0)) {
            asm volatile("" ::: "memory");
            apply_mask(A, S0a, 0);
            apply_mask(Bq, S0b, 0);
        }
        w4_for<16>([&](auto ic) { sm1_piece(ic, S0a, S0b); });
        A.alpha = 1.0f;                          // O is still zero
        Bq.alpha = 1.0f;
    }
    __builtin_amdgcn_s_barrier();               // K slot 0 is free for K(2)

    for (int j = 0; j < nt; j += 2) {
        iter(IC<0>{}, j, S0a, S0b, S1a, S1b);
        if (j + 1 < nt) iter(IC<1>{}, j + 1, S1a, S1b, S0a, S0b);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                  // last MFMA -> epilogue reads of O

    // ---- epilogue: normalise, 16-byte stores (cdna guide T21) ----------------------------------------------------------------------
    auto store_qb = [&](QB& X) {
        const float l_tot = row_pair_sum(X.l);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if constexpr (sizeof(OT) == 2) {
            char* orow = (char*)((OT*)p.o + (int64_t)b * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)min(X.my_q, p.Sq - 1) * p.o_ss) + 16 * h;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    v4 wa, wb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        wa[e] = (T)(X.o[db][4 * g + e] * inv);
                        wb[e] = (T)(X.o[db][4 * g + 4 + e] * inv);
                    }
                    const u32x2 ua = __builtin_bit_cast(u32x2, wa), ub = __builtin_bit_cast(u32x2, wb);
                    auto r0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
                    const u32x4 w = {r0[0], r1[0], r0[1], r1[1]};
                    if (X.my_q < p.Sq) *(u32x4*)(orow + 2 * (db * 32 + 8 * g)) = w;
                }
        } else if (X.my_q < p.Sq) {
            OT* orow = (OT*)p.o + (int64_t)b * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)X.my_q * p.o_ss;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = X.o[db][4 * g + e] * inv;
                    *(f32x4*)(orow + db * 32 + 8 * g + 4 * h) = w;
                }
        }
        if (X.my_q < p.Sq && p.lse && h == 0) {
            const float lse = l_tot > 0.f ? (X.m * c + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f : -INFINITY;
            p.lse[((int64_t)b * p.H + hh) * p.Sq + X.my_q] = lse;
        }
    };
    store_qb(A);
    store_qb(Bq);
}

}  // namespace pfa

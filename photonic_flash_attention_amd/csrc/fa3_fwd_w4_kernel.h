// fa3_fwd_w4_kernel.h -- Flash-Attention forward, D = 128: 4 waves x 64 query rows, ONE wave per SIMD (512 registers).
//
// Same math, HBM/LDS images, MFMA operand maps and LDS-DMA staging as fa3_fwd_kernel.h (read that first).  What changes
// is the ratio of LDS traffic to matrix work and who hides what:
//
//  * a wave owns TWO 32-row query blocks (a, b) and feeds BOTH from every K / V^T fragment it reads: half the LDS bytes
//    (and ds_read instructions) per MFMA of the 8-wave x 32-row kernel, which is power- and issue-limited by exactly that
//    (profiles/r01_ablation_lds_vs_mfma.txt);
//  * there is no partner wave on the SIMD, so the softmax is software-pipelined inside the wave's own stream and
//    hand-placed between the MFMAs: a tile is 64 half-steps of ONE MFMA + at most ~5 vector issues each, with a
//    sched_barrier behind the MFMA and behind its filler (two MFMAs back to back stall the in-order wave on the second):
//
//        iteration j :  phase A   S(j+1) = K(j+1) Q^T   (32 MFMA)  ||  finish softmax(j) -> P(j)            || K reads, all 8 DMA pieces
//                       phase B   O     += V(j)^T P(j)  (32 MFMA)  ||  start softmax(j+1) (max, exps of kb 0) || V^T reads
//
//    S is double-buffered in registers (2 x 64 VGPRs); O (128) and Q (64) live in the accumulator half of the register
//    file and are touched only by inline-asm MFMAs ("a" constraints), so hipcc never copies them through VGPRs.
//    K runs one tile ahead of V in the LDS rings (2 slots each, 64 KiB); one barrier per 64-key tile.
//    The O rescale decided in phase B (rare: defer-max) is applied after that phase's MFMAs.
//  * one loop body for every tile: masks (causal diagonal, key tail) are a wave-uniform branch between the two phases,
//    and a wave's last iteration treats the non-existent next tile as fully masked (its softmax start then leaves
//    m, l and O untouched) -- a second, non-interleaved body costs hipcc 300-600 bytes of scratch per lane.
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
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {      // one v_cvt_pk_bf16_f32 (plain C++: an asm
        typedef __attribute__((ext_vector_type(2))) __bf16 b2;                 // statement costs an s_nop behind it)
        const b2 t = {(__bf16)a, (__bf16)b};
        const uint32_t r = __builtin_bit_cast(uint32_t, t);
        asm volatile("" ::"v"(r));                                             // ... converted HERE, not sunk to its use
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
        const uint32_t r = __builtin_bit_cast(uint32_t, t);
        asm volatile("" ::"v"(r));
        return r;
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
#ifdef PFA_W4_STAMP
    const unsigned long long t_entry = stamp();
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();     // constant 100 MHz: absolute, comparable across CUs
#endif
    constexpr int TILE_BYTES = BLOCK_N * D * 2, HALF_TILE = TILE_BYTES / 2;
    constexpr int K_BASE = 0, V_BASE = 2 * TILE_BYTES;
    constexpr int Q_BASE = 4 * TILE_BYTES;                 // 64 KiB: the Q block's landing zone, [wave][strip][32 rows][256 B]
    constexpr int PPW = (TILE_BYTES / 1024) / NW;          // 4 DMA pieces per wave per image
    constexpr int PFK = 4, PFV = 3;                        // operand-fragment rings (steps ahead)

    // Fetch every kernel argument the fill needs in ONE batch of scalar loads: left alone, hipcc loads them where they are first
    // used -- ten dependent s_load / s_waitcnt round trips (~1500 cycles) before the first DMA piece could be issued.
    asm volatile("" ::"s"(p.q), "s"(p.k), "s"(p.v), "s"(p.seqlens_k), "s"(p.q_sb), "s"(p.q_sh), "s"(p.q_ss), "s"(p.k_sb), "s"(p.k_sh),
                 "s"(p.k_ss), "s"(p.v_sb), "s"(p.v_sh), "s"(p.v_ss), "s"(p.B), "s"(p.H), "s"(p.Sq), "s"(p.Sk), "s"(p.nqblk), "s"(p.kv_group),
                 "s"(p.xcd_group), "s"(p.magic_h), "s"(p.magic_g), "s"(p.scale_log2));
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Launch geometry (pfa_capi.hip): x = XCD + 8 * (head within the XCD's current group of G heads), y = Q block rank,
    // z = group.  Workgroups are dispatched x-fastest and workgroup n lands on XCD n % 8, so an XCD walks its heads G at a
    // time -- all Q blocks of a group, heaviest first, before the next group: the ~32 resident workgroups stream G heads'
    // K/V and a tile fetched into that XCD's L2 is re-read there by the head's other Q blocks (FETCH_SIZE at C3: 215 MB ->
    // 161 MB with G = 4).  xcd_group = 0: x = head, y = Q block rank.  No integer divisions on the way to the first load.
    const int qrank = blockIdx.y;
    const int bh = p.xcd_group > 0 ? (int)(blockIdx.x & 7) + 8 * ((int)blockIdx.z * p.xcd_group + (int)(blockIdx.x >> 3)) : (int)blockIdx.x;
    const int qblk = CAUSAL ? (p.nqblk - 1 - qrank) : qrank;
    // bh / H and hh / kv_group as multiply-high by floor(2^32 / d) + 1 (exact while n * d < 2^32: the host checks B * H * H;
    // d = 1 has no such constant)
    const int b = p.H == 1 ? bh : (int)__umulhi((uint32_t)bh, p.magic_h), hh = bh - b * p.H;
    const int hkv = p.kv_group == 1 ? hh : (int)__umulhi((uint32_t)hh, p.magic_g);
    const int q0 = qblk * BLOCK_M, wave_q0 = q0 + wave * 64;

    int kv_len = p.Sk;
    if (p.seqlens_k) kv_len = min(kv_len, max(p.seqlens_k[b], 0));
    const int kv_end = CAUSAL ? min(kv_len, q0 + BLOCK_M) : kv_len;
    const int nt = (kv_end + BLOCK_N - 1) / BLOCK_N;                                   // tiles the workgroup streams
    const int wave_kv_end = CAUSAL ? min(kv_len, wave_q0 + 64) : kv_len;
    const int wnt = (wave_kv_end + BLOCK_N - 1) / BLOCK_N;                             // tiles this wave computes

    const T* qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)hh * p.q_sh;
    const char* kp = (const char*)((const T*)p.k + (int64_t)b * p.k_sb + (int64_t)hkv * p.k_sh);
    const char* vp = (const char*)((const T*)p.v + (int64_t)b * p.v_sb + (int64_t)hkv * p.v_sh);

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
        X.my_q = first + r;                 // O is zeroed beside the Q DMA (dma_q), in the shadow of its issue stalls
        X.m = -1e30f;
        X.l = 0.f;
        X.mc = -1e30f * c;
        X.m_thr = -1e30f;
        X.alpha = 1.0f;
        X.ps0 = 0.f;
    };
    // Which 32-row strips a wave owns is free (a lane is a row).  Without a causal mask: strip w and its mirror 7 - w
    // (measured +1.6 % at C4 against two adjacent strips).  With one, adjacent strips: mirrored strips would balance
    // the diagonal but need a one-block loop body, which measured 4.6 % slower overall.
    init_qb(A, CAUSAL ? wave_q0 : q0 + 32 * wave);
    init_qb(Bq, CAUSAL ? wave_q0 + 32 : q0 + 32 * (7 - wave));

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
    // One descriptor per image for the whole kernel (base = the head's K / V slab, records = its bytes): the tile and
    // piece offsets ride in the per-lane offset (one s_add + one v_add per piece), so the range check still zero-fills
    // rows past Sk and nothing is rebuilt on the SALU per piece.  M0 is declared clobbered instead of saved/restored.
    auto whole_srd = [](const char* base, uint32_t bytes) {
        const uint64_t a = (uint64_t)(uintptr_t)base;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0,
                                                 (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    const srd_t ksrd = whole_srd(kp, k_slab), vsrd = whole_srd(vp, v_slab);
    auto dma_piece = [](srd_t srd, uint32_t voff, uint32_t lds_dst) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(srd), "s"(lds_dst)
                     : "memory", "m0");
    };
    const uint32_t k_tile = (uint32_t)(BLOCK_N * 2) * (uint32_t)p.k_ss, v_tile = (uint32_t)(BLOCK_N * 2) * (uint32_t)p.v_ss;
    // per-lane offsets of the four pieces of an image (piece t = 16 rows further), fixed for the kernel: a piece then costs
    // one scalar multiply-free add (tile offset, wave-uniform) folded into ONE v_add, m0, and the load itself.  No clamp:
    // offsets past the slab fail the descriptor's range check (nothing is written), and they cannot wrap (slab < 2 GiB).
    uint32_t koffd_t[PPW], voffd_t[PPW];
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
        koffd_t[t] = koffd + (uint32_t)t * 32u * (uint32_t)p.k_ss;
        voffd_t[t] = voffd + (uint32_t)t * 32u * (uint32_t)p.v_ss;
        asm volatile("" : "+v"(koffd_t[t]), "+v"(voffd_t[t]));
    }
    auto dma_k = [&](int j, int t) {      // piece t of K(j); rows past Sk arrive as zeros; tiles past the workgroup's last are skipped by the range check or fetched and never read
        dma_piece(ksrd, koffd_t[t] + (uint32_t)j * k_tile, smem_base + K_BASE + (j & 1) * TILE_BYTES + (wave + NW * t) * 1024);
    };
    auto dma_v = [&](int j, int t) {
        dma_piece(vsrd, voffd_t[t] + (uint32_t)j * v_tile, smem_base + V_BASE + (j & 1) * TILE_BYTES + (wave + NW * t) * 1024);
    };
    // Q takes the same road: per-lane 16-byte loads with a lane per ROW touch 32 cache lines per instruction and kept the
    // four waves' address units busy for ~5000 cycles of every workgroup's fill; 1-KiB DMA pieces cover four WHOLE rows each.
    // A wave fetches its own two strips into its own 16 KiB (chunk index XOR row, as in the store epilogue) and reads the
    // fragments back after its own vmcnt wait -- no barrier.  Rows past Sq re-read row Sq - 1 (never stored).
    const srd_t qsrd = whole_srd((const char*)qp, (uint32_t)(((int64_t)(p.Sq - 1) * p.q_ss + D) * 2));
    auto dma_q = [&](QB& X, int blk) {
        const int first = X.my_q - r;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = 4 * t + (lane >> 4);
            const uint32_t voff = (uint32_t)(min(first + row, p.Sq - 1) * (int)p.q_ss) * 2u + (uint32_t)(((lane & 15) ^ (row & 15)) << 4);
            dma_piece(qsrd, voff, smem_base + Q_BASE + wave * 16384 + blk * 8192 + t * 1024);
            if (t & 1) {                    // a quarter of O's zeros after every second piece
#pragma unroll
                for (int e = 0; e < 16; ++e) X.o[t >> 1][e] = 0.f;
                asm volatile("" : "+a"(X.o[t >> 1]));
            }
        }
    };
    auto read_q = [&](QB& X, int blk) {
        const uint32_t base = smem_base + Q_BASE + wave * 16384 + blk * 8192 + r * 256;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) X.qf[ks] = *(const lds_v8*)(uintptr_t)(base + (((2 * ks + h) ^ (r & 15)) << 4));
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

    // ---- the half-steps of one tile -----------------------------------------------------------------------------------------
    // One half-step = ONE MFMA plus the vector work issued in its 32-cycle shadow (an in-order wave that issues two MFMAs
    // back to back stalls on the second until the pipe is free and wastes the first one's shadow).  A tile is 32 QK^T
    // half-steps (phase A, beside the softmax FINISH of the previous tile) and 32 PV half-steps (phase B, beside the
    // softmax START of the next tile).  Half-step hs: fragment step hs / 2, query block hs % 2.
    v8 kfr[PFK];
    auto k_read = [&](int slot, int i) { return *(const lds_v8*)(uintptr_t)(koff[i % KS] + slot * TILE_BYTES + (i / KS) * HALF_TILE); };
    auto qk_begin = [&](int slot) {
#pragma unroll
        for (int i = 0; i < PFK; ++i) kfr[i] = k_read(slot, i);
    };
    auto qk_half = [&](auto hc, int slot, f32x16 (&sa)[2], f32x16 (&sb)[2]) {
        constexpr int hs = decltype(hc)::value, i = hs / 2, kb = i / KS, ks = i % KS;
        if constexpr (hs % 2 == 0) {
            if constexpr (ks == 0) M::s0(sa[kb], kfr[i % PFK], A.qf[ks]);
            else M::s(sa[kb], kfr[i % PFK], A.qf[ks]);
        } else {
            if constexpr (ks == 0) M::s0(sb[kb], kfr[i % PFK], Bq.qf[ks]);
            else M::s(sb[kb], kfr[i % PFK], Bq.qf[ks]);
            if constexpr (i + PFK < 2 * KS) kfr[i % PFK] = k_read(slot, i + PFK);
        }
    };
    // softmax finish, half-step hs: block (hs < 16 ? a : b), element e = hs % 16 of key block 1 gets its exponential;
    // even e: the key-block-0 P dword of the pair (e, e+1); odd e: sum + P dword of the previous key-block-1 pair
    float ps1 = 0.f;
    auto sm2_end = [&](QB& X, f32x16 (&s)[2], uint32_t (&pd)[16]) {
#pragma unroll
        for (int e = 12; e < 16; ++e) {
            ps1 += s[1][e];
        }
        pd[15] = M::pack2(s[1][14], s[1][15]);
        X.l += X.ps0 + ps1;
        ps1 = 0.f;
    };
    // even e (half-steps with no LDS read to issue): exp(e), the key-block-0 P dword of (e, e+1), the sums of the
    // key-block-1 pair finished two half-steps ago; odd e: exp(e) and the P dword of the previous key-block-1 pair
    auto sm2_one = [&](QB& X, int e, f32x16 (&s)[2], uint32_t (&pd)[16]) {
        s[1][e] = fast_exp2(__builtin_fmaf(s[1][e], c, -X.mc));
        if (e == 0) X.ps0 += s[0][15];            // left over from the softmax start (its sums lag by one half-step)
        if ((e & 1) == 0) {
            pd[(e >> 3) * 4 + ((e & 7) >> 1)] = M::pack2(s[0][e], s[0][e + 1]);
            if (e >= 4) {
                ps1 += s[1][e - 4];
                ps1 += s[1][e - 3];
                asm volatile("" ::"v"(ps1));      // summed HERE (input-only anchor: no hazard nop behind it)
            }
        } else if (e >= 3) {
            pd[(2 + ((e - 3) >> 3)) * 4 + (((e - 3) & 7) >> 1)] = M::pack2(s[1][e - 3], s[1][e - 2]);
        }
    };
    auto sm2_half = [&](auto hc, f32x16 (&ca)[2], f32x16 (&cb)[2], uint32_t (&pa)[16], uint32_t (&pb)[16]) {
        constexpr int hs = decltype(hc)::value;
        if constexpr (hs < 16) {
            sm2_one(A, hs, ca, pa);
        } else {
            if constexpr (hs == 16) sm2_end(A, ca, pa);
            sm2_one(Bq, hs - 16, cb, pb);
        }
    };
    // PV half-step: fragment step idx = hs / 2 -> k-step (kb, s2) = idx / 4, d block db = idx % 4
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
    auto pv_half = [&](auto hc, int slot, const uint32_t (&pa)[16], const uint32_t (&pb)[16]) {
        constexpr int hs = decltype(hc)::value, idx = hs / 2, f = idx / 4, db = idx % 4;
        v8 a;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = vlo[idx % PFV][e];
            a[4 + e] = vhi[idx % PFV][e];
        }
        if constexpr (hs % 2 == 0) {
            const u32x4 fa = {pa[4 * f], pa[4 * f + 1], pa[4 * f + 2], pa[4 * f + 3]};
            M::o(A.o[db], a, fa);
        } else {
            const u32x4 fb = {pb[4 * f], pb[4 * f + 1], pb[4 * f + 2], pb[4 * f + 3]};
            M::o(Bq.o[db], a, fb);
            if constexpr (idx + PFV < 16) v_read(slot, idx + PFV, vlo[idx % PFV], vhi[idx % PFV]);
        }
    };
    // softmax start, half-step hs: block (hs < 16 ? a : b), local u = hs % 16: u 0-3 row max (8 values each, then the
    // lane pair), u 4 the branch-free per-row defer-max update, u 5-15 the exponentials and sums of key block 0
    float mx_carry = 0.f;
    uint64_t any_grow = 0;
    auto sm1_one = [&](QB& X, int u, f32x16 (&s)[2]) {
        if (u < 4) {
            const f32x16& t = s[u >> 1];
            const int o8 = (u & 1) * 8;
            float mx = (u == 0) ? -INFINITY : mx_carry;
            asm("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %0, %0, %7, %8"
                : "+v"(mx)
                : "v"(t[o8]), "v"(t[o8 + 1]), "v"(t[o8 + 2]), "v"(t[o8 + 3]), "v"(t[o8 + 4]), "v"(t[o8 + 5]), "v"(t[o8 + 6]), "v"(t[o8 + 7]));
            mx_carry = (u == 3) ? row_pair_max_asm(mx) : mx;
        } else if (u == 4) {
            // branch-free, per row: a max inside the headroom leaves m alone, and then alpha = exp2(0) = 1 exactly
            const bool grow = mx_carry > X.m_thr;
            any_grow |= __builtin_amdgcn_ballot_w64(grow);   // scalar: decides the (rare) O rescale after this phase
            const float m_new = grow ? mx_carry : X.m;
            const float al = fast_exp2((X.m - m_new) * c);
            X.m = m_new;
            X.m_thr = m_new + thr;
            X.mc = m_new * c;
            X.l *= al;
            X.alpha = al;                                    // O *= alpha after the PV MFMAs in flight beside this
            X.ps0 = 0.f;
        } else {
            // 16 elements over 11 half-steps: 1,2,1,2,...,1 -- one on the odd half-steps (they also issue the two V^T
            // reads and their wait), two on the even ones.  (Running the v_fma one element ahead of its v_exp and the
            // sum one behind was measured 3 % slower.)
            const int v = u - 5, lo = (v / 2) * 3 + (v & 1), n = (v & 1) ? 2 : 1, nprev = (v == 0) ? 0 : ((v & 1) ? 1 : 2);
#pragma unroll
            for (int e = lo; e < lo + n; ++e) s[0][e] = fast_exp2(__builtin_fmaf(s[0][e], c, -X.mc));
            // sums lag one half-step behind the exponentials (a VALU right behind the v_exp it depends on costs an
            // s_nop); element 15 is added by the first finish half-step of the next iteration
#pragma unroll
            for (int e = lo - nprev; e < lo; ++e) X.ps0 += s[0][e];
            asm volatile("" ::"v"(X.ps0));
        }
    };
    auto sm1_half = [&](auto hc, f32x16 (&na)[2], f32x16 (&nb)[2]) {
        constexpr int hs = decltype(hc)::value;
        if constexpr (hs < 16) sm1_one(A, hs, na);
        else sm1_one(Bq, hs - 16, nb);
    };
    // element masks without control flow: key(kb, e) < lim  <=>  32 kb + (e & 3) + 8 (e >> 2) < lim - key_base - 4 h
    auto apply_mask = [&](QB& X, f32x16 (&s)[2], int key_base, bool none = false) {
        int lim = none ? 0 : kv_len;
        if (CAUSAL) lim = min(lim, X.my_q + 1);
        const int t = lim - key_base - 4 * h;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float x = s[kb][e];
                x = (32 * kb + (e & 3) + 8 * (e >> 2) < t) ? x : -INFINITY;
                s[kb][e] = x;
            }
    };
    auto needs_mask = [&](int key_base) { return (key_base + BLOCK_N > kv_len) || (CAUSAL && key_base + BLOCK_N - 1 > wave_q0); };
    // the deferred O rescale (rare): O never leaves the accumulator file
    auto rescale = [&](QB& X) {
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
        X.alpha = 1.0f;
    };
    auto publish = [&]() {          // this wave's DMA pieces have landed; everyone is done with the slots about to be refilled
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    };

    // ---- iterations ---------------------------------------------------------------------------------------------------------
    // interleaved: tile j's finish + PV beside tile j+1's QK^T + start (no masks on tile j+1)
    unsigned long long st_acc[24] = {}, st_last = 0;
    auto steady = [&](auto pc, int j, f32x16 (&ca)[2], f32x16 (&cb)[2], f32x16 (&na)[2], f32x16 (&nb)[2]) {
        constexpr int P = decltype(pc)::value;
        uint32_t pa[16], pb[16];
#ifdef PFA_W4_STAMP
        const unsigned long long t0 = stamp();
#endif
        qk_begin(P ^ 1);
        w4_for<32>([&](auto hc) {
            constexpr int hs = decltype(hc)::value;
            qk_half(hc, P ^ 1, na, nb);
            __builtin_amdgcn_sched_barrier(0);      // the MFMA first, its shadow's work after it (hipcc otherwise alternates)
            // 8 DMA pieces, one every fourth half-step: the last lands a full phase before the barrier's vmcnt(0)
            if constexpr (hs % 4 == 1 && hs < 16) dma_v(j + 1, hs / 4);
            else if constexpr (hs % 4 == 1) dma_k(j + 2, hs / 4 - 4);
            sm2_half(hc, ca, cb, pa, pb);
            __builtin_amdgcn_sched_barrier(0);
        });
        sm2_end(Bq, cb, pb);
#ifdef PFA_W4_STAMP
        const unsigned long long t1 = stamp();
        st_acc[0] += t1 - t0;
#endif
        // diagonal / key-tail tile, or no tile j+1 at all for this wave (then S(j+1) is junk and is masked out whole:
        // its softmax start leaves m, l and O untouched): wave-uniform, rare
        const bool last = j + 1 >= wnt;
        if (last || needs_mask((j + 1) * BLOCK_N)) {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // last QK^T MFMA -> VALU reads of S(j+1)
            apply_mask(A, na, (j + 1) * BLOCK_N, last);
            apply_mask(Bq, nb, (j + 1) * BLOCK_N, last);
        }
        pv_begin(P);
        __builtin_amdgcn_sched_barrier(0);
        w4_for<32>([&](auto hc) {
            pv_half(hc, P, pa, pb);
            __builtin_amdgcn_sched_barrier(0);
            sm1_half(hc, na, nb);
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef PFA_W4_STAMP
        const unsigned long long t2 = stamp();
        st_acc[1] += t2 - t1;
        st_acc[4] += 1;
#endif
        if (any_grow != 0) {                      // rare (defer-max): some row's max outgrew its headroom
            rescale(A);
            rescale(Bq);
            any_grow = 0;
        }
#ifdef PFA_W4_STAMP
        st_last = stamp();
        st_acc[3] += st_last - t2;
#endif
    };
    auto iter = [&](auto pc, int j, f32x16 (&ca)[2], f32x16 (&cb)[2], f32x16 (&na)[2], f32x16 (&nb)[2]) {
        if (j < wnt) {
            steady(pc, j, ca, cb, na, nb);
        } else {                     // this wave is past its last tile: keep feeding the rings, keep the barriers
#pragma unroll
            for (int t = 0; t < PPW; ++t) dma_v(j + 1, t);
#pragma unroll
            for (int t = 0; t < PPW; ++t) dma_k(j + 2, t);
        }
        publish();
#ifdef PFA_W4_STAMP
        if (j < wnt) st_acc[2] += stamp() - st_last;
#endif
    };

    // ---- prologue -----------------------------------------------------------------------------------------------------------
    f32x16 S0a[2], S0b[2], S1a[2], S1b[2];      // S(j) of even / odd tiles
    // The fill of a workgroup is bandwidth bound (Q 64 KiB + three 16-KiB tiles at ~11 B/cycle/CU): wait only for what the
    // first QK^T needs (Q, K0) and let V0 and K1 land under QK^T(0) and the first softmax start.
#ifdef PFA_W4_STAMP
    st_acc[16] = stamp() - t_entry;                                       // setup done
#endif
    // Issue order = arrival order: strip A's Q rows, K0, strip B's Q rows.  About 44 KiB are in flight per CU at a time (the
    // pieces' issue stalls behind that), so strip A's QK^T(0) and softmax start run while strip B's rows are still arriving.
    dma_q(A, 0);
#pragma unroll
    for (int t = 0; t < PPW; ++t) dma_k(0, t);
    dma_q(Bq, 1);
#ifdef PFA_W4_STAMP
    st_acc[8] = stamp() - t_entry;                                        // setup + issue of the Q and K0 pieces
#endif
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                      // strip A's rows and this wave's K0 pieces
#ifdef PFA_W4_STAMP
    st_acc[9] = stamp() - t_entry;                                        // ... + Q(A) / K0 arrival
#endif
    read_q(A, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(A.qf[ks]));    // Q into the accumulator file, once
    __builtin_amdgcn_s_barrier();               // everyone's K0 pieces are in LDS
#ifdef PFA_W4_STAMP
    st_acc[10] = stamp() - t_entry;                                       // ... + barrier
#endif
    auto qk_one = [&](auto ic, QB& X, f32x16 (&s)[2]) {      // QK^T(0) of one strip: 16 MFMAs, K fragments from slot 0
        constexpr int i = decltype(ic)::value, kb = i / KS, ks = i % KS;
        if constexpr (ks == 0) M::s0(s[kb], kfr[i % PFK], X.qf[ks]);
        else M::s(s[kb], kfr[i % PFK], X.qf[ks]);
        if constexpr (i + PFK < 2 * KS) kfr[i % PFK] = k_read(0, i + PFK);
    };
    if (wnt > 0) {
        qk_begin(0);
        w4_for<16>([&](auto ic) {               // V0 and K1 are issued in the MFMA shadows
            constexpr int i = decltype(ic)::value;
            qk_one(ic, A, S0a);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i % 2 == 1) {
                if constexpr (i < 8) dma_v(0, i / 2);
                else dma_k(1, i / 2 - 4);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");              // last MFMA -> first VALU read of S
        if (needs_mask(0)) {
            asm volatile("" ::: "memory");
            apply_mask(A, S0a, 0);
        }
        w4_for<16>([&](auto hc) { sm1_one(A, decltype(hc)::value, S0a); });
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                      // strip B's rows (V0 and K1 were issued behind them)
    read_q(Bq, 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(Bq.qf[ks]));
    if (wnt > 0) {
        qk_begin(0);
        w4_for<16>([&](auto ic) { qk_one(ic, Bq, S0b); });
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        if (needs_mask(0)) {
            asm volatile("" ::: "memory");
            apply_mask(Bq, S0b, 0);
        }
        w4_for<16>([&](auto hc) { sm1_one(Bq, decltype(hc)::value, S0b); });
        A.alpha = 1.0f;                          // O is still zero
        Bq.alpha = 1.0f;
        any_grow = 0;
    }
#ifdef PFA_W4_STAMP
    st_acc[11] = stamp() - t_entry;                                       // ... + QK^T(0) + first softmax start
#endif
    publish();                                  // V0 and K1 landed; K slot 0 is free for K(2)

#ifdef PFA_W4_STAMP
    const unsigned long long t_loop0 = stamp();
    st_acc[5] = t_loop0 - t_entry;
#endif
    for (int j = 0; j < nt; j += 2) {
        iter(IC<0>{}, j, S0a, S0b, S1a, S1b);
        if (j + 1 < nt) iter(IC<1>{}, j + 1, S1a, S1b, S0a, S0b);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                  // last MFMA -> epilogue reads of O
#ifdef PFA_W4_STAMP
    const unsigned long long t_loop1 = stamp();
    st_acc[6] = t_loop1 - t_loop0;
#endif

    // ---- epilogue: normalise, 16-byte stores (cdna guide T21) ----------------------------------------------------------------------
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    auto store_qb = [&](QB& X, int blk) {
        const float l_tot = row_pair_sum(X.l);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 inv2 = {inv, inv};
        if constexpr (sizeof(OT) == 2) {
            // 16-bit store through LDS (free after the loop's last barrier; every wave uses its own 16 KiB): a lane pair
            // first forms 16-byte chunks of its row (cdna guide T21), the block is written as a [32 rows][256 B] image
            // with the chunk index XOR-swizzled by the row, and read back so that one store instruction covers four WHOLE
            // rows -- per-lane stores at the row stride touch 64 cache lines per instruction, these touch 8.
            const uint32_t lbase = smem_base + wave * 16384 + blk * 8192;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    v4 wa, wb;
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {     // v_pk_mul_f32: two products per instruction
                        const f32x2 pa = f32x2{X.o[db][4 * g + e], X.o[db][4 * g + e + 1]} * inv2;
                        const f32x2 pb = f32x2{X.o[db][4 * g + 4 + e], X.o[db][4 * g + 5 + e]} * inv2;
                        wa[e] = (T)pa[0]; wa[e + 1] = (T)pa[1];
                        wb[e] = (T)pb[0]; wb[e + 1] = (T)pb[1];
                    }
                    const u32x2 ua = __builtin_bit_cast(u32x2, wa), ub = __builtin_bit_cast(u32x2, wb);
                    auto r0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
                    const u32x4 w = {r0[0], r1[0], r0[1], r1[1]};
                    const uint32_t ch = 4 * db + g + h;                   // 16-byte chunk of the row this lane now holds
                    *(lds_u32x4*)(uintptr_t)(lbase + r * 256 + ((ch ^ (r & 15)) << 4)) = w;
                }
            const int first = __builtin_amdgcn_readfirstlane(X.my_q - r);
            store_rows_from_lds<256>(lbase, lane, (char*)((OT*)p.o + (int64_t)b * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)first * p.o_ss),
                                     p.o_ss * 2, p.Sq - first);
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
    store_qb(A, 0);
    store_qb(Bq, 1);
#ifdef PFA_W4_STAMP
    st_acc[12] = stamp() - t_loop1;                                       // normalise + stage + store issue
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_acc[7] = stamp() - t_loop1;
    st_acc[13] = rt_entry;
    st_acc[14] = __builtin_amdgcn_s_memrealtime();
    st_acc[15] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);   // HW_ID, XCC_ID
    if (lane == 0 && p.dbg) {
        unsigned long long* d = p.dbg + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * NW + wave) * 24;
#pragma unroll
        for (int i = 0; i < 24; ++i) d[i] = st_acc[i];
    }
#endif
}

}  // namespace pfa

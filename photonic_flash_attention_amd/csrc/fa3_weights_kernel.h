// fa3_weights_kernel.h -- attention weights P = softmax(scale * Q K^T + mask), materialised [B,H,Sq,Sk].
//
// The reference returns them on request (need_weights=True): the dense branch hands back the softmax matrix
// (core/flash_attention_3.py:171,180), the tiled branch a per-tile, never re-normalised slice (:257-258, row sums
// 1.4-1.6 -- a reference defect we do not reproduce).  Here the weights are always the true softmax: a second,
// memory-bound pass recomputes S = Q K^T with the same swapped 32x32x16 MFMA product as the forward kernel and
// writes exp2(c*s - lse*log2e) using the forward pass's LSE.  Traffic = B*H*Sq*Sk output elements, which is why
// the forward kernel never writes them unless asked.
//
// Workgroup = 128 query rows, 4 waves.  K arrives as the forward's tile image by LDS-DMA (whole rows in 1-KiB pieces; round 2 read the
// fragments straight from global memory, a lane per key row: 32 cache lines per load instruction, and at D = 128 the address unit set
// the pace).  Two schedules (see the fast path): D = 64 -- a wave takes every fourth 64-key column block for all four 32-row strips,
// K tile in a region of its own, no barrier; D = 128 -- a wave keeps one strip, K tiles shared and double buffered.  Fully masked key
// blocks of a causal problem are not computed: the kernel writes their zeros (every element of W is written exactly once).
#pragma once
#include "fa3_fwd_kernel.h"
#include "fa3_bwd_kernels.h"      // TileDma / TileRead

namespace pfa {

struct WeightsParams {
    const void* q;
    const void* k;
    const float* lse;          // [B,H,Sq] natural-log LSE from the forward pass
    void* w;                   // [B,H,Sq,Sk] by strides, last dim contiguous
    const int32_t* seqlens_k;
    const uint8_t* mask;
    int64_t q_sb, q_sh, q_ss;
    int64_t k_sb, k_sh, k_ss;
    int64_t w_sb, w_sh, w_sq;
    int64_t m_sb, m_sh, m_sq, m_sk;
    int32_t B, H, Sq, Sk;
    int32_t nqblk;
    int32_t kv_group;
    float scale_log2;
    const unsigned long long* mbits;   // mask condensed to words (FwdParams::mbits), or null: bytes
    int64_t mb_sb, mb_sh, mb_sq;
};

template <typename T, int D, bool CAUSAL, bool KMASK, typename WT>
__global__ __launch_bounds__(256, 2) void fa3_weights_kernel(const WeightsParams p) {
    using E = Elem<T>;
    using v8 = typename E::v8;
    constexpr int KS = D / 16, NS = 4;               // k-steps; 32-row strips per workgroup
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;
    // workgroup n runs on XCD n % 8 (round 3 read it off HW_REG_XCC_ID): a head's Q blocks follow each other on ONE XCD, so the
    // workgroups an XCD holds at a time share a few heads' K in its 4 MiB of L2
    const int BH = p.B * p.H;
    const int n = blockIdx.x;
    int qblk, bh;
    if ((BH & 7) == 0) {
        const int xcd = n & 7, slot = n >> 3, hl = slot / p.nqblk;
        qblk = slot - hl * p.nqblk;
        bh = hl * 8 + xcd;
    } else {
        bh = n / p.nqblk;
        qblk = n - bh * p.nqblk;
    }
    const int b = bh / p.H;
    const int hh = bh - b * p.H;
    const int q0 = qblk * 128;

    int kv_len = p.Sk;
    if (p.seqlens_k) kv_len = min(kv_len, max(p.seqlens_k[b], 0));
    // keys any row of strip s can see (wave-uniform)
    auto strip_kv_end = [&](int s) { return CAUSAL ? min(kv_len, q0 + 32 * s + 32) : kv_len; };

    const T* __restrict__ qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)hh * p.q_sh;
    const T* __restrict__ kp = (const T*)p.k + (int64_t)b * p.k_sb + (int64_t)(hh / p.kv_group) * p.k_sh;
    WT* const whead = (WT*)p.w + (int64_t)b * p.w_sb + (int64_t)hh * p.w_sh;
    const float c = p.scale_log2;

    // per-row state of one strip: lane (r, h) holds row q0 + 32 s + r
    struct Strip {
        v8 qf[KS];
        float lse2;                // lse * log2(e); -inf for a fully masked row -> weights 0
        int my_q;
        bool dead;
        const uint8_t* mp;
        const unsigned long long* mw;      // the row's mask words (one per 64 keys), or null
    };
    auto load_strip = [&](Strip& S, int s) {
        S.my_q = q0 + 32 * s + r;
        const int qrow = min(S.my_q, p.Sq - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) S.qf[ks] = *(const v8*)(qp + (int64_t)qrow * p.q_ss + 16 * ks + 8 * h);
        const float lse = p.lse[((int64_t)b * p.H + hh) * p.Sq + qrow];
        S.dead = !(lse > -INFINITY);
        S.lse2 = S.dead ? INFINITY : lse * 1.4426950408889634f;      // dead rows: exp2(x - inf) = 0 without a select
        S.mp = KMASK ? p.mask + (int64_t)b * p.m_sb + (int64_t)hh * p.m_sh + (int64_t)qrow * p.m_sq : nullptr;
        S.mw = (KMASK && p.mbits) ? p.mbits + (int64_t)b * p.mb_sb + (int64_t)hh * p.mb_sh + (int64_t)qrow * p.mb_sq : nullptr;
    };
    auto load_k = [&](int key_base, v8 (&kf)[KS]) {      // lane (key r, h): K[key][16 ks + 8 h .. +7]
        const int krow = min(key_base + r, p.Sk - 1);
        const T* ksrc = kp + (int64_t)krow * p.k_ss + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *(const v8*)(ksrc + 16 * ks);
    };
    // one 32-key block of S^T = K Q^T -> weights of this lane's row, keys key_base + (e&3) + 8(e>>2) + 4h
    // whole: every key of the block exists and is visible to every row of the strip (wave-uniform; the caller knows) -- no per-element tests
    auto weights_of = [&](const Strip& S, int key_base, const f32x16& s, float (&w)[16], bool whole) {
        if (whole) {
#pragma unroll
            for (int e = 0; e < 16; ++e) w[e] = fast_exp2(__builtin_fmaf(s[e], c, -S.lse2));
            return;
        }
        uint32_t mbits32 = 0;                    // the lane's mask bits of this 32-key block (words path)
        if constexpr (KMASK) {
            if (S.mw) mbits32 = (uint32_t)(S.mw[key_base >> 6] >> ((key_base & 32) + 4 * h));
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = key_base + (e & 3) + 8 * (e >> 2) + 4 * h;
            bool ok = key < kv_len && !S.dead;
            if (CAUSAL) ok = ok && (key <= S.my_q);
            if (KMASK) {
                if (S.mw) ok = ok && (((mbits32 >> ((e & 3) + 8 * (e >> 2))) & 1u) != 0);
                else ok = ok && (S.mp[(int64_t)min(key, p.Sk - 1) * p.m_sk] != 0);
            }
            w[e] = ok ? fast_exp2(__builtin_fmaf(s[e], c, -S.lse2)) : 0.f;
        }
    };
    auto block_k = [&](const Strip& S, int key_base, const v8 (&kf)[KS], float (&w)[16], bool whole = false) {
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = E::mfma(kf[ks], S.qf[ks], s);
        weights_of(S, key_base, s, w, whole);
    };

    // Fast path, 64 keys at a time.  The 32 x 64 block of a strip goes through LDS (one [32 rows][64 * sizeof(WT)] image per wave,
    // 16-byte units XOR-swizzled by the row) and leaves as WHOLE 128- or 256-byte row segments, 8 or 4 rows per store instruction
    // (store_rows_from_lds) -- per-lane stores write 8/16-byte pieces at the row stride and ran at 1.2-2.4 TB/s.
    constexpr int ES = sizeof(WT), CB = 4 * ES, RB = 64 * ES, NU = RB / 16;
    constexpr int TILE_BYTES = BLOCK_N * D * 2, HALF_TILE = TILE_BYTES / 2;
    constexpr bool COLS = (D == 64);          // which of the two schedules below
    __shared__ __attribute__((aligned(1024))) char kbuf[(COLS ? 4 : 2) * TILE_BYTES];      // COLS: a K tile per wave; else two shared ones
    __shared__ __attribute__((aligned(16))) char wbuf[4 * 32 * RB];
    typedef __attribute__((address_space(3))) char lchar;
    typedef __attribute__((address_space(3))) v8 lds_v8;
    lchar* const lbase = (lchar*)wbuf + wave * (32 * RB);
    const uint32_t kb_base = (uint32_t)(uintptr_t)(lchar*)kbuf;
    const int64_t k_slab = ((int64_t)(p.Sk - 1) * p.k_ss + D) * 2;
    const bool vec_ok = ((p.w_sq * ES) % 16 == 0) && ((reinterpret_cast<uintptr_t>(p.w) + (p.w_sb * b + p.w_sh * hh) * ES) % 16 == 0) &&
                        k_slab < (1ll << 31);                     // (the DMA's offsets are 32-bit)
    const int fast_limit = vec_ok ? (p.Sk / 64) * 64 : 0;         // keys [0, fast_limit) can leave as whole 64-key segments
    // the 16 weights of key block kb of a strip's 64-key block -> the wave's staging image
    auto stage = [&](const float (&w)[16], int kb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ci = 8 * kb + 2 * g + h;                  // 4-element chunk of the 64-key row
            const int u = (ci * CB) >> 4, uo = (ci * CB) & 15;  // its 16-byte unit and the offset inside
            lchar* dst = lbase + r * RB + ((u ^ (r & (NU - 1))) << 4) + uo;
            if constexpr (ES == 4) {
                *(__attribute__((address_space(3))) f32x4*)dst = f32x4{w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]};
            } else {
                typename E::v4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (T)w[4 * g + e];
                *(__attribute__((address_space(3))) typename E::v4*)dst = t;
            }
        }
    };
    // nothing of the 64-key block is visible to strip s (above the diagonal, past the batch's key length): the kernel writes the zeros
    // itself -- W is written exactly once, the caller allocates it uninitialised
    auto zeros = [&](int s, int key_base) {
        constexpr int RPI = 64 / NU, NI = 32 / RPI;
        const int lr = lane / NU, rows_valid = p.Sq - (q0 + 32 * s);
        char* g = (char*)(whead + (int64_t)(q0 + 32 * s + lr) * p.w_sq + key_base) + 16 * (lane & (NU - 1));
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (RPI * i + lr < rows_valid) *(u32x4*)(g + (int64_t)(RPI * i) * p.w_sq * ES) = z;
    };
    auto store_block = [&](int s, int key_base) {
        store_rows_from_lds<RB>((uint32_t)(uintptr_t)lbase, lane, (char*)(whead + (int64_t)(q0 + 32 * s) * p.w_sq + key_base), p.w_sq * ES,
                                p.Sq - (q0 + 32 * s));
    };
    // the block lies below strip s's diagonal and inside the batch's keys, and no element mask is set: plain exponentials
    auto whole_block = [&](int s, int key_base) { return !KMASK && key_base + 64 <= kv_len && (!CAUSAL || key_base + 63 <= q0 + 32 * s); };
    if constexpr (COLS) {
        // D = 64 -- a wave takes every FOURTH 64-key column block for all four row strips: its K tile arrives by LDS-DMA in a region of its
        // own (no barrier anywhere), goes to registers once and serves the four strips; the next tile is requested as soon as the
        // fragments are out of the LDS, under the whole round's work.  The four waves write four ADJACENT blocks of the same 128 rows at
        // about the same time: 512 contiguous bytes a row.
        Strip S[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) load_strip(S[s], s);
        const int wg_kv_end = min(strip_kv_end(NS - 1), fast_limit);
        const uint32_t kpriv = kb_base + wave * TILE_BYTES;
        TileDma<D, 4> dma[4];                                      // (the image's swizzle depends on the piece's place: four lane maps)
#pragma unroll
        for (int vw = 0; vw < 4; ++vw) dma[vw].init(vw, lane, p.k_ss, p.k_ss);
        TileRead<T, D> rk;
        rk.init(lane, kpriv);
        auto fetch = [&](int key_base) {
#pragma unroll
            for (int vw = 0; vw < 4; ++vw) dma[vw].issue1(vw, key_base >> 6, (const char*)kp, p.k_ss, k_slab, kpriv);
        };
        if (64 * wave < wg_kv_end) fetch(64 * wave);
        for (int key_base = 64 * wave; key_base < fast_limit; key_base += 64 * 4) {
            v8 kf[2][KS];
            if (key_base < wg_kv_end) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile (and this wave's stores of the round before)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *(const lds_v8*)(uintptr_t)(rk.row_off[ks] + kb * HALF_TILE);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kf[kb][ks]));      // all fragments are in registers ...
                if (key_base + 256 < wg_kv_end) fetch(key_base + 256);                       // ... the region is free: the next tile
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (q0 + 32 * s >= p.Sq) continue;                                      // wave-uniform
                if (key_base >= strip_kv_end(s)) {
                    zeros(s, key_base);
                    continue;
                }
                const bool whole = whole_block(s, key_base);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float w[16];
                    block_k(S[s], key_base + 32 * kb, kf[kb], w, whole);
                    stage(w, kb);
                }
                store_block(s, key_base);
            }
        }
    } else {
        // D = 128 (four strips' Q fragments and a tile's K fragments do not fit 256 registers) -- a wave keeps ONE strip; K tile j + 1 is on
        // its way (DMA, two shared buffers, one barrier a step) while the waves take their 32 x 64 blocks of tile j, fragments through a
        // register ring.  The stores are issued BEHIND the step's barrier, so the counted wait in front of it meets only stores that are
        // a whole step old.  128-byte row segments only: 3.7 TB/s is what that pattern stores at with nothing else going on.
        Strip S;
        load_strip(S, wave);                                       // (rows past Sq: clamped loads, nothing of them is stored)
        TileDma<D, 4> dma;
        dma.init(wave, lane, p.k_ss, p.k_ss);
        TileRead<T, D> rk;
        rk.init(lane, kb_base);
        const bool live = q0 + 32 * wave < p.Sq;                   // wave-uniform
        const int my_kv_end = strip_kv_end(wave);
        const int nfast = fast_limit / 64;
        const int nt = min(nfast, (strip_kv_end(NS - 1) + 63) / 64);   // tiles some strip of the workgroup sees
        if (nt > 0) dma.issue1(wave, 0, (const char*)kp, p.k_ss, k_slab, kb_base);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // tile 0 is there
        for (int j = 0; j < nt; ++j) {
            const uint32_t boff = (uint32_t)(j & 1) * TILE_BYTES;
            if (j + 1 < nt) dma.issue1(wave, j + 1, (const char*)kp, p.k_ss, k_slab, kb_base + (boff ^ TILE_BYTES));
            const int key_base = 64 * j;
            const bool seen = live && key_base < my_kv_end;        // wave-uniform
            if (seen) {
                // S^T of both key blocks: 2 KS K fragments through a PF-deep register ring, order pinned with sched_group_barrier
                // (hipcc otherwise emits read -> wait -> MFMA on one 4-register buffer: an LDS round trip per MFMA, 1800 cycles a step)
                f32x16 sc[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sc[kb][e] = 0.f;
                constexpr int NOP = 2 * KS, PF = 4;
                auto frag = [&](int i) { return *(const lds_v8*)(uintptr_t)(rk.row_off[i % KS] + boff + (i / KS) * HALF_TILE); };
                v8 afr[PF];
#pragma unroll
                for (int i = 0; i < PF; ++i) afr[i] = frag(i);
#pragma unroll
                for (int i = 0; i < NOP; ++i) {
                    sc[i / KS] = E::mfma(afr[i % PF], S.qf[i % KS], sc[i / KS]);
                    if (i + PF < NOP) afr[i % PF] = frag(i + PF);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
                for (int i = 0; i < NOP - PF; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);
                const bool whole = whole_block(wave, key_base);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float w[16];
                    weights_of(S, key_base + 32 * kb, sc[kb], w, whole);
                    stage(w, kb);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of tile j + 1 (and its stores of step j - 1, a step old)
            __builtin_amdgcn_s_waitcnt(0xC07F);                    // its reads of tile j
            __builtin_amdgcn_s_barrier();
            if (seen) store_block(wave, key_base);
            else if (live) zeros(wave, key_base);
        }
        if (live)
            for (int j = nt; j < nfast; ++j) zeros(wave, 64 * j);
    }
    // remaining key blocks of strip `wave` (the last, partial 64 keys; everything when rows are unaligned): per-lane stores
    const int wave_q0 = q0 + 32 * wave;
    if (wave_q0 >= p.Sq) return;
    Strip S;
    load_strip(S, wave);
    const int kv_end = strip_kv_end(wave);
    WT* __restrict__ wrow = whead + (int64_t)min(S.my_q, p.Sq - 1) * p.w_sq;
    // (runs to Sk, not to the strip's last visible key: masked elements come out of block_k as zeros and are written too)
    (void)kv_end;
    for (int key_base = fast_limit; key_base < p.Sk; key_base += 32) {
        v8 kf[KS];
        load_k(key_base, kf);
        float w[16];
        block_k(S, key_base, kf, w);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int key0 = key_base + 8 * g + 4 * h;
            if (S.my_q < p.Sq) {
                if (key0 + 3 < p.Sk && ((reinterpret_cast<uintptr_t>(wrow + key0) & (4 * sizeof(WT) - 1)) == 0)) {
                    if constexpr (sizeof(WT) == 4) {
                        *(f32x4*)(wrow + key0) = f32x4{w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]};
                    } else {
                        typename E::v4 t;
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = (T)w[4 * g + e];
                        *(typename E::v4*)(wrow + key0) = t;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (key0 + e < p.Sk) wrow[key0 + e] = (WT)w[4 * g + e];
                }
            }
        }
    }
}

}  // namespace pfa

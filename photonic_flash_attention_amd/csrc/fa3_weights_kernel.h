// fa3_weights_kernel.h -- attention weights P = softmax(scale * Q K^T + mask), materialised [B,H,Sq,Sk].
//
// The reference returns them on request (need_weights=True): the dense branch hands back the softmax matrix
// (core/flash_attention_3.py:171,180), the tiled branch a per-tile, never re-normalised slice (:257-258, row sums
// 1.4-1.6 -- a reference defect we do not reproduce).  Here the weights are always the true softmax: a second,
// memory-bound pass recomputes S = Q K^T with the same swapped 32x32x16 MFMA product as the forward kernel and
// writes exp2(c*s - lse*log2e) using the forward pass's LSE.  Traffic = B*H*Sq*Sk output elements, which is why
// the forward kernel never writes them unless asked.
//
// Workgroup = 128 query rows, 4 waves; a wave takes every fourth 64-key column block for all four 32-row strips, K fragments straight
// from global/L2 to registers and shared by the four strips (no LDS for K).  Fully masked key blocks of a
// causal problem are not computed: the kernel writes their zeros (every element of W is written exactly once).
#pragma once
#include "fa3_fwd_kernel.h"

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
__global__ __launch_bounds__(256) void fa3_weights_kernel(const WeightsParams p) {
    using E = Elem<T>;
    using v8 = typename E::v8;
    constexpr int KS = D / 16, NS = 4;               // k-steps; 32-row strips per workgroup
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;
    const int BH = p.B * p.H;
    const int n = blockIdx.x;
    const int qblk = n / BH;
    const int bh = n - qblk * BH;
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
        S.lse2 = lse * 1.4426950408889634f;
        S.dead = !(lse > -INFINITY);
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
    auto block_k = [&](const Strip& S, int key_base, const v8 (&kf)[KS], float (&w)[16]) {
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = E::mfma(kf[ks], S.qf[ks], s);
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

    // Fast path, 64 keys at a time.  A wave takes every FOURTH 64-key column block and walks all four row strips with the K
    // fragments it loaded once: with a wave per strip every wave fetched all of K, a lane per key row = 32 cache lines per load
    // instruction, and at D = 128 the address unit, not the store, set the pace (2.2 TB/s).  The 32 x 64 block of a strip goes
    // through LDS (one [32 rows][64 * sizeof(WT)] image per wave, 16-byte units XOR-swizzled by the row) and leaves as WHOLE 128-
    // or 256-byte row segments, 8 or 4 rows per store instruction (store_rows_from_lds) -- per-lane stores write 8/16-byte
    // pieces at the row stride (half a cache line per row and key block) and ran at 1.2-2.4 TB/s.
    constexpr int ES = sizeof(WT), CB = 4 * ES, RB = 64 * ES, NU = RB / 16;
    __shared__ __attribute__((aligned(16))) char wbuf[4 * 32 * RB];
    typedef __attribute__((address_space(3))) char lchar;
    lchar* const lbase = (lchar*)wbuf + wave * (32 * RB);
    const bool vec_ok = ((p.w_sq * ES) % 16 == 0) && ((reinterpret_cast<uintptr_t>(p.w) + (p.w_sb * b + p.w_sh * hh) * ES) % 16 == 0);
    const int fast_limit = vec_ok ? (p.Sk / 64) * 64 : 0;         // keys [0, fast_limit) can leave as whole 64-key segments
    {
        Strip S[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) load_strip(S[s], s);
        const int wg_kv_end = strip_kv_end(NS - 1);
        for (int key_base = 64 * wave; key_base < fast_limit; key_base += 64 * 4) {
            v8 kf[2][KS];
            if (key_base < wg_kv_end) {
                load_k(key_base, kf[0]);
                load_k(key_base + 32, kf[1]);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (q0 + 32 * s >= p.Sq) continue;                                      // wave-uniform
                if (key_base >= strip_kv_end(s)) {
                    // nothing of this block is visible to the strip (above the diagonal, past the batch's key length): the kernel
                    // writes the zeros itself -- W is written exactly once, the caller allocates it uninitialised
                    constexpr int RPI = 64 / NU, NI = 32 / RPI;
                    const int lr = lane / NU, rows_valid = p.Sq - (q0 + 32 * s);
                    char* g = (char*)(whead + (int64_t)(q0 + 32 * s + lr) * p.w_sq + key_base) + 16 * (lane & (NU - 1));
                    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        if (RPI * i + lr < rows_valid) *(u32x4*)(g + (int64_t)(RPI * i) * p.w_sq * ES) = z;
                    continue;
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float w[16];
                    block_k(S[s], key_base + 32 * kb, kf[kb], w);
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
                }
                store_rows_from_lds<RB>((uint32_t)(uintptr_t)lbase, lane, (char*)(whead + (int64_t)(q0 + 32 * s) * p.w_sq + key_base),
                                        p.w_sq * ES, p.Sq - (q0 + 32 * s));
            }
        }
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

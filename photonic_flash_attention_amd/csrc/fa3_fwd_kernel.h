// fa3_fwd_kernel.h -- Flash-Attention forward for MI355X (gfx950 / CDNA4), hand-written HIP.
//
// Replaces the reference's eager two-level tile loop (core/flash_attention_3.py:182-262 and the
// dense branch :152-180) with ONE kernel: per 256-row Q block a workgroup of 8 waves walks the
// K/V sequence in 64-key tiles, S^T = K Q^T and O^T += V^T P^T on the bf16/f16 32x32x16 MFMA,
// online softmax in registers (flash_attention_3.py:239-250, but un-normalised until the end).
//
// Geometry (one workgroup = 512 threads = 8 waves, one per 32 Q rows; 1 workgroup per CU):
//   * "Swapped" products: the MFMA computes S^T (keys on the accumulator rows, the Q row on the lane),
//     so one lane holds 32 scores of ONE query row: row max / row sum are in-lane reductions plus one
//     v_permlane32_swap with the lane that holds the row's other 32 keys.
//   * The S^T accumulator registers, converted pairwise to bf16, ARE the B operand of the PV product
//     (k order permuted: element j of lane half h of k-step s is key 16s + 8(j>>2) + 4h + (j&3)); the
//     V^T A operand is fetched in that same key order with ds_read_b64_tr_b16 (hardware transpose
//     read), so P never touches LDS and V is staged row-major exactly as it lies in HBM.
//   * K/V tiles: HBM -> registers (16-byte coalesced loads issued BEFORE the tile's math) -> LDS
//     (written AFTER it) -> one barrier per tile, LDS double-buffered.
//   * LDS image: 256-byte rows, 16-byte chunk index XOR-swizzled with ((R&3)<<2 | (R>>2)&3), which is
//     conflict-free for the ds_read_b128 row reads of K and for the transposed reads of V alike.
//
// HBM traffic per workgroup: Q block once, K/V of its (b,h) once (L2/MALL absorb re-reads by the
// other Q blocks of the head), O once.  Algorithmic bytes per forward: 2(Sq+Sk)*D*2 B per (b,h)*... see
// DESIGN.md.  The kernel is MFMA-bound at D=128 (1024 flop/B causal) -- roofline = bf16 dense MFMA peak.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((address_space(3))) char lds_char;

constexpr int BLOCK_M = 256;  // Q rows per workgroup
constexpr int WAVE_M = 32;    // Q rows per wave
constexpr int BLOCK_N = 64;   // keys per K/V tile
constexpr int NTHREADS = 512;

struct FwdParams {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;
    const int32_t* seqlens_k;
    const uint8_t* key_mask;
    int64_t q_sb, q_sh, q_ss;
    int64_t k_sb, k_sh, k_ss;
    int64_t v_sb, v_sh, v_ss;
    int64_t o_sb, o_sh, o_ss;
    int64_t km_sb;
    int32_t B, H, Sq, Sk;
    int32_t nqblk;        // ceil(Sq / BLOCK_M)
    float scale_log2;     // softmax_scale * log2(e)
};

template <typename T> struct Elem;
template <> struct Elem<__bf16> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ v4 tr_read(const lds_char* p) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4*)p);
    }
};
template <> struct Elem<_Float16> {
    using v8 = f16x8;
    using v4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ v4 tr_read(const lds_char* p) {
        typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;
        return __builtin_bit_cast(v4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)p));
    }
};

// Byte offset of 16-byte chunk `ch` of key row `key` inside one K or V tile image.
// D=128: one key per 256-B LDS row.  D=64: two keys per LDS row (odd key in the upper 128 B).
template <int D>
__device__ __forceinline__ uint32_t tile_off(uint32_t key, uint32_t ch) {
    uint32_t R, c;
    if constexpr (D == 128) {
        R = key;
        c = ch;
    } else {
        R = key >> 1;
        c = ((key & 1) << 3) | ch;
    }
    const uint32_t sw = ((R & 3) << 2) | ((R >> 2) & 3);
    return R * 256u + ((c ^ sw) << 4);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max(x[lane], x[lane ^ 32]) in every lane: the two lanes that share a query row.
__device__ __forceinline__ float row_pair_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float row_pair_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <typename T, int D, bool CAUSAL, bool SPLITP, typename OT>
__global__ __launch_bounds__(NTHREADS, 2) void fa3_fwd_kernel(const FwdParams p) {
    using E = Elem<T>;
    using v8 = typename E::v8;
    using v4 = typename E::v4;
    constexpr int KS = D / 16;                // k-steps of the QK^T product
    constexpr int DB = D / 32;                // 32-wide d blocks of the PV product
    constexpr int CPR = D / 8;                // 16-byte chunks per key row
    constexpr int TILE_BYTES = BLOCK_N * D * 2;
    constexpr int CHUNKS_PER_THREAD = (BLOCK_N * CPR) / NTHREADS;  // 2 (D=128) or 1 (D=64)
    static_assert(CHUNKS_PER_THREAD >= 1, "tile too small for 512 threads");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [buf][K|V][TILE_BYTES]
    lds_char* const smem_l = (lds_char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    // ---- block -> (q block, batch*head): heaviest (longest causal row) blocks first ----------------
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
    const int kv_end = CAUSAL ? min(kv_len, q0 + BLOCK_M) : kv_len;          // keys the block needs
    const int wave_kv_end = CAUSAL ? min(kv_len, wave_q0 + WAVE_M) : kv_len; // keys this wave needs
    const int nt = (kv_end + BLOCK_N - 1) / BLOCK_N;

    const T* __restrict__ qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)hh * p.q_sh;
    const T* __restrict__ kp = (const T*)p.k + (int64_t)b * p.k_sb + (int64_t)hh * p.k_sh;
    const T* __restrict__ vp = (const T*)p.v + (int64_t)b * p.v_sb + (int64_t)hh * p.v_sh;
    const uint8_t* __restrict__ kmp = p.key_mask ? p.key_mask + (int64_t)b * p.km_sb : nullptr;

    // ---- Q fragments: B operand of S^T = K Q^T, lane (r,h) holds Q[my_q][16 ks + 8 h .. +7] -----------
    v8 qf[KS];
    {
        const int qrow = min(my_q, p.Sq - 1);
        const T* src = qp + (int64_t)qrow * p.q_ss + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const v8*)(src + 16 * ks);
    }

    // ---- K/V staging: thread t moves chunk(s) t, t+512 of the [64][D] tile -------------------------------
    u32x4 kreg[CHUNKS_PER_THREAD], vreg[CHUNKS_PER_THREAD];
    uint32_t st_off[CHUNKS_PER_THREAD];   // LDS byte offset inside a tile image
    int st_key[CHUNKS_PER_THREAD];
    int st_col[CHUNKS_PER_THREAD];
#pragma unroll
    for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
        const int c = tid + i * NTHREADS;
        st_key[i] = c / CPR;
        st_col[i] = (c % CPR) * 8;
        st_off[i] = tile_off<D>(st_key[i], c % CPR);
    }
    auto load_tile = [&](int j) {
#pragma unroll
        for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
            const int key = min(j * BLOCK_N + st_key[i], p.Sk - 1);
            kreg[i] = *(const u32x4*)(kp + (int64_t)key * p.k_ss + st_col[i]);
            vreg[i] = *(const u32x4*)(vp + (int64_t)key * p.v_ss + st_col[i]);
        }
    };
    auto store_tile = [&](int buf) {
        lds_char* kb_ = smem_l + buf * 2 * TILE_BYTES;
        lds_char* vb_ = kb_ + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
            *(__attribute__((address_space(3))) u32x4*)(kb_ + st_off[i]) = kreg[i];
            *(__attribute__((address_space(3))) u32x4*)(vb_ + st_off[i]) = vreg[i];
        }
    };

    // ---- per-lane LDS read offsets (inside a tile image) ---------------------------------------------------
    // K row read for key block kb, k-step ks: key = 32 kb + r, chunk = 2 ks + h
    // V transposed read for (kb, s2, dblk, half): lane 4q+p of a 16-lane group gives row q, columns 4p..4p+3
    const int g1 = (lane >> 4) & 1;
    const int tq = (lane & 15) >> 2;
    const int tp = lane & 3;

    f32x16 o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    float m_run = -1e30f;   // running row max, raw score units
    float l_run = 0.f;      // this lane's share of the row sum
    const float c = p.scale_log2;

    if (nt > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();

    for (int j = 0; j < nt; ++j) {
        const int cur = j & 1;
        if (j + 1 < nt) load_tile(j + 1);   // HBM latency hides under this tile's math

        const int key_base = j * BLOCK_N;
        if (key_base < wave_kv_end) {       // wave-uniform: causal tiles right of this wave's rows are skipped
            const lds_char* kimg = smem_l + cur * 2 * TILE_BYTES;
            const lds_char* vimg = kimg + TILE_BYTES;

            // ---- S^T = K Q^T : two 32-key blocks ---------------------------------------------------------
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const v8 a = *(const __attribute__((address_space(3))) v8*)(kimg + tile_off<D>(32 * kb + r, 2 * ks + h));
                    s[kb] = E::mfma(a, qf[ks], s[kb]);
                }
            }

            // ---- mask (tile-uniform test; only diagonal / tail / key-mask tiles pay) -------------------
            const bool need_mask = (key_base + BLOCK_N > kv_len) || (CAUSAL && key_base + BLOCK_N - 1 > wave_q0) ||
                                   (kmp != nullptr);
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = key_base + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * h;
                        bool ok = key < kv_len;
                        if (CAUSAL) ok = ok && (key <= my_q);
                        if (kmp) ok = ok && (kmp[min(key, p.Sk - 1)] != 0);
                        s[kb][e] = ok ? s[kb][e] : -INFINITY;
                    }
            }

            // ---- online softmax (flash_attention_3.py:239-246), row = (lane, lane^32) ------------------
            float mx = s[0][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[0][e]);
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[1][e]);
            mx = row_pair_max(mx);
            const float m_new = fmaxf(m_run, mx);
            const float mc = m_new * c;
            const float alpha = fast_exp2(m_run * c - mc);
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pe = fast_exp2(__builtin_fmaf(s[kb][e], c, -mc));
                    s[kb][e] = pe;
                    psum += pe;
                }
            l_run = l_run * alpha + psum;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // wave-uniform: skip when no row max moved
#pragma unroll
                for (int i = 0; i < DB; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
            }

            // ---- O^T += V^T P^T ---------------------------------------------------------------------------
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    v8 ph, pl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float pv = s[kb][8 * s2 + e];
                        const T hi = (T)pv;
                        ph[e] = hi;
                        if (SPLITP) pl[e] = (T)(pv - (float)hi);
                    }
                    const int key0 = 32 * kb + 16 * s2 + 4 * h + tq;
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const uint32_t ch = db * 4 + 2 * g1 + (tp >> 1);
                        const v4 lo = E::tr_read(vimg + tile_off<D>(key0, ch) + 8 * (tp & 1));
                        const v4 hi4 = E::tr_read(vimg + tile_off<D>(key0 + 8, ch) + 8 * (tp & 1));
                        v8 a;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a[e] = lo[e];
                            a[4 + e] = hi4[e];
                        }
                        o[db] = E::mfma(a, ph, o[db]);
                        if (SPLITP) o[db] = E::mfma(a, pl, o[db]);
                    }
                }
        }

        if (j + 1 < nt) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: normalise (flash_attention_3.py:250 does it per tile; once is equivalent) ------------
    const float l_tot = row_pair_sum(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;   // fully masked row -> zeros (documented divergence)
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

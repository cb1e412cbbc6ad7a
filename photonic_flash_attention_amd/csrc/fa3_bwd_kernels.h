// fa3_bwd_kernels.h -- Flash-Attention BACKWARD for MI355X (gfx950): dQ, dK, dV from the forward's LSE.
//
// The reference gets its backward from autograd through the eager forward (flash_attention_3.py:152-262;
// tests/unit/test_flash_attention_3.py:137-160 only require that gradients exist).  Here it is three kernels
// that recompute P = exp(scale*S - LSE) tile by tile instead of storing the S x S matrices:
//
//   fa3_bwd_delta   delta[b,h,i] = sum_d dO[i,d] * O[i,d]                       (memory bound; since the end of round 1 the
//                   dQ kernel computes it for its own rows and publishes it for the dK/dV kernel -- this one is not launched)
//   fa3_bwd_dq      per 256-row Q block (8 waves x 32 rows, same geometry / LDS images / DMA as the forward):
//                     S^T = K Q^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T
//   fa3_bwd_dkdv    per 128-key block (4 waves x 32 keys, one wave per SIMD: K^T/V^T operand fragments and the
//                   dK^T/dV^T accumulators of the block stay in registers), streaming Q and dO tiles:
//                     S = Q K^T, dP = dO V^T, dV^T += dO^T P, dS = P o (dP - delta), dK^T += Q^T dS
//
// dQ is recomputed from its own pass (7 MFMA products in total instead of 5) rather than summed across key
// blocks with float atomics: bitwise reproducible, no dQ zero-fill, no atomic floor (cdna guide Appendix B).
// Operand maps are those of fa3_fwd_kernel.h: the accumulator of the first product, converted to bf16/f16, is
// the B operand of the product that contracts over its ROW index (k order permuted), and the other operand of
// that product is fetched in the same order by ds_read_b64_tr_b16 from the row-major, XOR-swizzled tile image.
#pragma once
#include "fa3_fwd_kernel.h"

namespace pfa {

struct BwdParams {
    const void* q;
    const void* k;
    const void* v;
    const void* o;
    const void* dout;
    const float* lse;      // [B,H,Sq] natural log
    float* delta;          // [B,H,Sq] workspace
    void* dq;
    void* dk;
    void* dv;
    const int32_t* seqlens_k;
    const uint8_t* mask;   // optional element mask (KMASK kernels), byte strides
    int64_t m_sb, m_sh, m_sq, m_sk;
    int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss;
    int64_t o_sb, o_sh, o_ss, do_sb, do_sh, do_ss;
    int64_t dq_sb, dq_sh, dq_ss, dk_sb, dk_sh, dk_ss, dv_sb, dv_sh, dv_ss;
    int32_t B, H, Sq, Sk;
    int32_t nblk;          // Q blocks (dq) or key blocks (dkdv)
    // element mask condensed by pfa_fa3_bwd into `mask_workspace` (null: the byte paths below): a word per mask row and 64-key tile + the
    // first / last key tile with a visible key per 256 rows (dQ kernel); the same transposed -- a word per key and 64-row tile + the
    // first / last row tile per 128 keys (dK/dV kernel).  Word strides / pair strides, 0 = broadcast; RANGE_PARTS pairs per granule.
    const unsigned long long* mw_row = nullptr;
    int64_t mw_sb = 0, mw_sh = 0, mw_sq = 0;
    const int* rg_row = nullptr;
    int64_t rg_sb = 0, rg_sh = 0;
    int32_t rg_q = 0;
    const unsigned long long* mw_col = nullptr;
    int64_t cw_sb = 0, cw_sh = 0;
    int32_t ntq = 0;
    const int* rg_col = nullptr;
    int64_t crg_sb = 0, crg_sh = 0;
    int32_t kv_group;      // >= 1: query heads per K/V head (k, v, dk, dv hold H / kv_group heads; query head h uses K/V head h / kv_group)
    // a mask that depends on the key only ([B,Sk]: strides over heads and rows are 0) needs no KMASK kernels: the dK/dV kernel folds it
    // into its per-lane "this key exists" flag, the dQ kernel reads it four keys per load
    int32_t mask_dw;            // element mask rows are 4-byte aligned and Sk % 4 == 0: the dQ kernel reads them four keys per load
    const uint8_t* keymask;     // u8 [B][Sk], 0 = masked; or null
    int64_t km_sb;              // byte stride between batches (keys contiguous; host: Sk, base and stride multiples of 4)
    float scale;           // softmax scale
    float scale_log2;      // scale * log2(e)
};

// ---------------------------------------------------------------------------------------------------------------
// delta = rowsum(dO o O): one wave per 64 rows? -> one thread per (row, 8-element chunk), reduced in-wave.
template <typename T, int D>
__global__ __launch_bounds__(256) void fa3_bwd_delta_kernel(const BwdParams p) {
    using v8 = typename Elem<T>::v8;
    constexpr int CPR = D / 8;                       // 16-byte chunks per row
    constexpr int ROWS_PER_BLOCK = 256 / CPR;
    const int bh = blockIdx.y;
    const int b = bh / p.H, hh = bh - b * p.H;
    const int row = blockIdx.x * ROWS_PER_BLOCK + threadIdx.x / CPR;
    const int ch = threadIdx.x % CPR;
    float acc = 0.f;
    if (row < p.Sq) {
        const T* op = (const T*)p.o + (int64_t)b * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)row * p.o_ss + ch * 8;
        const T* gp = (const T*)p.dout + (int64_t)b * p.do_sb + (int64_t)hh * p.do_sh + (int64_t)row * p.do_ss + ch * 8;
        const v8 a = *(const v8*)op, g = *(const v8*)gp;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)g[e];
    }
#pragma unroll
    for (int off = CPR / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (row < p.Sq && ch == 0) p.delta[((int64_t)b * p.H + hh) * p.Sq + row] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// Shared tile machinery: 64-row x D tile images (two per stage), LDS-DMA by buffer descriptor, swizzled.
// raw-buffer descriptor whose inputs are provably wave-uniform (readfirstlane folds away when they already are)
__device__ __forceinline__ srd_t uniform_srd(const char* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0,
                                             (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int D, int NW>
struct TileDma {
    static constexpr int TILE_BYTES = BLOCK_N * D * 2;
    static constexpr int PIECES = TILE_BYTES / 1024;
    static constexpr int PPW = PIECES / NW;
    static constexpr int ROWS_PER_T = (D == 128) ? 4 * NW : 8 * NW;   // source rows between a wave's pieces
    uint32_t off_a, off_b;               // per-lane byte offsets inside the two source slabs (piece 0)
    __device__ __forceinline__ void init(int wave, int lane, int64_t stride_a, int64_t stride_b) {
        const int R0 = 4 * wave + (lane >> 4);
        const int sw = ((R0 & 3) << 2) | ((R0 >> 2) & 3);
        const int cc = (lane & 15) ^ sw;
        int row, col;
        if constexpr (D == 128) {
            row = R0;
            col = cc * 8;
        } else {
            row = 2 * R0 + (cc >> 3);
            col = (cc & 7) * 8;
        }
        off_a = (uint32_t)(row * (int)stride_a + col) * 2u;
        off_b = (uint32_t)(row * (int)stride_b + col) * 2u;
    }
    // tile j of slab A -> LDS at lds_a, of slab B -> lds_b (rows past the slab read as zeros).  Slabs are < 2 GiB
    // (pfa_fa3_bwd checks it), so offsets and remaining-byte counts are 32-bit SALU work.
    __device__ __forceinline__ void issue(int wave, int j, const char* base_a, int64_t stride_a, int64_t slab_a,
                                          uint32_t lds_a, const char* base_b, int64_t stride_b, int64_t slab_b,
                                          uint32_t lds_b) const {
        // one descriptor per piece (SALU only): the uniform row step lives in the base so that the range check still
        // sees it (an SGPR soffset would bypass the check and ragged tails would read past the slab)
        const uint32_t step_a = (uint32_t)(ROWS_PER_T * 2) * (uint32_t)stride_a, step_b = (uint32_t)(ROWS_PER_T * 2) * (uint32_t)stride_b;
        const uint32_t sa0 = (uint32_t)j * (uint32_t)(BLOCK_N * 2) * (uint32_t)stride_a;
        const uint32_t sb0 = (uint32_t)j * (uint32_t)(BLOCK_N * 2) * (uint32_t)stride_b;
        const uint32_t la = (uint32_t)slab_a, lb = (uint32_t)slab_b;
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            const uint32_t sa = sa0 + t * step_a, sb = sb0 + t * step_b;
            const srd_t ra = uniform_srd(base_a + sa, la > sa ? la - sa : 0u);
            const srd_t rb = uniform_srd(base_b + sb, lb > sb ? lb - sb : 0u);
            lds_dma16_buf(ra, off_a, lds_a + (wave + NW * t) * 1024);
            lds_dma16_buf(rb, off_b, lds_b + (wave + NW * t) * 1024);
        }
    }
    // ... of slab A alone (fa3_weights_kernel.h: K tiles)
    __device__ __forceinline__ void issue1(int wave, int j, const char* base_a, int64_t stride_a, int64_t slab_a, uint32_t lds_a) const {
        const uint32_t step_a = (uint32_t)(ROWS_PER_T * 2) * (uint32_t)stride_a;
        const uint32_t sa0 = (uint32_t)j * (uint32_t)(BLOCK_N * 2) * (uint32_t)stride_a;
        const uint32_t la = (uint32_t)slab_a;
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            const uint32_t sa = sa0 + t * step_a;
            lds_dma16_buf(uniform_srd(base_a + sa, la > sa ? la - sa : 0u), off_a, lds_a + (wave + NW * t) * 1024);
        }
    }
};

// compile-time loop: f(IC<0>{}), f(IC<1>{}), ... f(IC<N-1>{})
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<N, I + 1>(f);
    }
}

// per-lane read offsets inside one tile image.  LEAN (D = 128 only): keep one row base and two transposed-read
// bases and derive the others with one v_xor each at the point of use -- the swizzle is an XOR on the chunk bits, so
// chunk 2ks+h of row r sits at base(r, h) ^ (ks << 5) and d-block db at base ^ (db << 6).  Saves 13 VGPRs in the
// dK/dV kernel, whose budget is 256 for two workgroups per CU.  (volatile asm: must not be hoisted back out.)
template <int C>
__device__ __forceinline__ uint32_t xor_const(uint32_t x) {
    if constexpr (C == 0) return x;
    uint32_t o;
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(o) : "i"(C), "v"(x));
    return o;
}
template <typename T, int D, bool LEAN = false>
struct TileRead {
    static_assert(!LEAN || D == 128, "lean addressing relies on one row per 256-byte LDS row");
    static constexpr int KS = D / 16, DB = D / 32, NS2 = (D == 128) ? 1 : 2;
    uint32_t row_off[LEAN ? 1 : KS];                              // row read (32 rows r, chunk 2ks+h), +HALF_TILE for rows 32..63
    uint32_t tr_off[LEAN ? 1 : NS2][LEAN ? 1 : DB][2];            // transposed read (see fa3_fwd_kernel.h)
    __device__ __forceinline__ void init(int lane, uint32_t base) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < (LEAN ? 1 : KS); ++ks) {
            row_off[ks] = base + tile_off<D>(r, 2 * ks + h);
            asm volatile("" : "+v"(row_off[ks]));      // opaque: hipcc otherwise keeps row and chunk parts apart and re-adds them per tile
        }
        const int g1 = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
#pragma unroll
        for (int s2 = 0; s2 < (LEAN ? 1 : NS2); ++s2)
#pragma unroll
            for (int db = 0; db < (LEAN ? 1 : DB); ++db)
#pragma unroll
                for (int hi = 0; hi < 2; ++hi)
                {
                    tr_off[s2][db][hi] = base + tile_off<D>(16 * s2 + 4 * h + tq + 8 * hi, db * 4 + 2 * g1 + (tp >> 1)) + 8 * (tp & 1);
                    asm volatile("" : "+v"(tr_off[s2][db][hi]));
                }
    }
    template <int KSI>
    __device__ __forceinline__ uint32_t row_at() const {
        if constexpr (LEAN) return xor_const<(KSI << 5)>(row_off[0]);
        else return row_off[KSI];
    }
    template <int S2, int DBI, int HI>
    __device__ __forceinline__ uint32_t tr_at() const {
        if constexpr (LEAN) return xor_const<(DBI << 6)>(tr_off[0][0][HI]);
        else return tr_off[S2][DBI][HI];
    }
};

// eight fp32 -> one 8 x 16-bit MFMA operand as four pair conversions (v_cvt_pk_bf16_f32 for bf16): written element by
// element, hipcc converts singly and re-packs with v_alignbit / v_mov (48 instead of 16 instructions per 32 values)
template <typename T>
__device__ __forceinline__ typename Elem<T>::v8 pack8(const float (&x)[8]) {
    typedef __attribute__((ext_vector_type(2))) T t2;
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const t2 t = {(T)x[2 * i], (T)x[2 * i + 1]};
        r[i] = __builtin_bit_cast(uint32_t, t);
    }
    return __builtin_bit_cast(typename Elem<T>::v8, r);
}

// A wave's 32 x D fp32 tile held row-per-lane-pair (acc[db][4g+e] = element d = 32 db + 8 g + 4 h + e of row r), scaled,
// converted to 16 bits and written to global memory as WHOLE rows through LDS: lane pairs first form 16-byte chunks
// (v_permlane32_swap), the tile is laid out [32 rows][D*2 B] with the chunk index XOR-swizzled by the row, and is read
// back so that one store instruction covers 4 (D = 128) or 8 (D = 64) full rows (fa3_fwd_kernel.h epilogue; the per-lane
// form touches 64 cache lines per instruction).  `lbase`: this wave's private 32*D*2-byte LDS region, free to use.
template <typename T, int D>
__device__ __forceinline__ void store_tile_rows_via_lds(const f32x16 (&acc)[D / 32], float mul, uint32_t lbase, int lane,
                                                        char* grow0, int64_t row_stride_bytes, int rows_valid) {
    using v4 = typename Elem<T>::v4;
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;
    constexpr int RB = D * 2, CPRW = RB / 16;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            v4 wa, wb;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wa[e] = (T)(acc[db][4 * g + e] * mul);
                wb[e] = (T)(acc[db][4 * g + 4 + e] * mul);
            }
            const u32x2 ua = __builtin_bit_cast(u32x2, wa), ub = __builtin_bit_cast(u32x2, wb);
            auto r0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
            const u32x4 w = {r0[0], r1[0], r0[1], r1[1]};
            const uint32_t ch = 4 * db + g + h;
            *(lds_u32x4_t*)(uintptr_t)(lbase + r * RB + ((ch ^ (r & (CPRW - 1))) << 4)) = w;
        }
    store_rows_from_lds<RB>(lbase, lane, grow0, row_stride_bytes, rows_valid);
}

// ---------------------------------------------------------------------------------------------------------------
// dQ: forward geometry.  LDS stage = [K image | V image], double buffered (64 KiB at D = 128).
template <typename T, int D, bool CAUSAL, bool KMASK, typename OT>
__global__ __launch_bounds__(512, 2) void fa3_bwd_dq_kernel(const BwdParams p) {
    using E = Elem<T>;
    using v8 = typename E::v8;
    using v4 = typename E::v4;
    typedef __attribute__((address_space(3))) v8 lds_v8;
    constexpr int NW = 8, BLOCK_M = 256, KS = D / 16, DB = D / 32;
    constexpr int TILE_BYTES = BLOCK_N * D * 2, BUF_BYTES = 2 * TILE_BYTES, HALF_TILE = TILE_BYTES / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int BH = p.B * p.H, n = blockIdx.x, qrank = n / BH, bh = n - qrank * BH;
    const int qblk = CAUSAL ? (p.nblk - 1 - qrank) : qrank;
    const int b = bh / p.H, hh = bh - b * p.H;
    const int q0 = qblk * BLOCK_M, wave_q0 = q0 + wave * WAVE_M, my_q = wave_q0 + r;

    int kv_len = p.Sk;
    if (p.seqlens_k) kv_len = min(kv_len, max(p.seqlens_k[b], 0));
    const int kv_end = CAUSAL ? min(kv_len, q0 + BLOCK_M) : kv_len;
    const int wave_kv_end = CAUSAL ? min(kv_len, wave_q0 + WAVE_M) : kv_len;
    int nt = (kv_end + BLOCK_N - 1) / BLOCK_N;
    int j0 = 0;                              // first key tile of the block (element mask: the tile range of its 256 rows)
    if constexpr (KMASK) {
        if (p.rg_row) {
            const int* rg = p.rg_row + 2 * RANGE_PARTS * ((int64_t)b * p.rg_sb + (int64_t)hh * p.rg_sh + (p.rg_q ? (q0 >> 8) : 0));
            int lo = rg[2 * (lane & (RANGE_PARTS - 1))], hi = rg[2 * (lane & (RANGE_PARTS - 1)) + 1];
#pragma unroll
            for (int off = RANGE_PARTS / 2; off >= 1; off >>= 1) {
                lo = min(lo, __shfl_xor(lo, off));
                hi = max(hi, __shfl_xor(hi, off));
            }
            nt = max(min(nt, __builtin_amdgcn_readfirstlane(hi) + 1), 0);
            j0 = min(__builtin_amdgcn_readfirstlane(lo), nt) & ~1;
        }
    }

    const T* qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)hh * p.q_sh;
    const T* gp = (const T*)p.dout + (int64_t)b * p.do_sb + (int64_t)hh * p.do_sh;
    const int hk = hh / p.kv_group;          // grouped-query heads: the K/V head this query head reads
    const char* kp = (const char*)((const T*)p.k + (int64_t)b * p.k_sb + (int64_t)hk * p.k_sh);
    const char* vp = (const char*)((const T*)p.v + (int64_t)b * p.v_sb + (int64_t)hk * p.v_sh);
    const int64_t k_slab = ((int64_t)(p.Sk - 1) * p.k_ss + D) * 2, v_slab = ((int64_t)(p.Sk - 1) * p.v_ss + D) * 2;

    const int qrow = min(my_q, p.Sq - 1);
    v8 qf[KS], gf[KS];                       // Q and dO fragments of this lane's row (B operands)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *(const v8*)(qp + (int64_t)qrow * p.q_ss + 16 * ks + 8 * h);
        gf[ks] = *(const v8*)(gp + (int64_t)qrow * p.do_ss + 16 * ks + 8 * h);
    }
    const uint8_t* mrow = KMASK ? p.mask + (int64_t)b * p.m_sb + (int64_t)hh * p.m_sh + (int64_t)qrow * p.m_sq : nullptr;
    // the row's mask words, one per key tile; the word of tile j + 1 is requested at the top of step j, ahead of that step's DMA pieces,
    // so the wait hipcc puts in front of its use never holds the K/V prefetch up (the byte / dword loads inside `compute` did)
    const unsigned long long* mwrow = (KMASK && p.mw_row) ? p.mw_row + (int64_t)b * p.mw_sb + (int64_t)hh * p.mw_sh + (int64_t)qrow * p.mw_sq : nullptr;
    unsigned long long cur_bits = ~0ull, next_bits = ~0ull;
    const int64_t stat = ((int64_t)b * p.H + hh) * p.Sq + qrow;
    const float lse = p.lse[stat];
    // delta = rowsum(dO o O) of this lane's row, computed here (this kernel holds the dO row anyway; the lane pair (l, l^32)
    // splits the row) and published for the dK/dV kernel that runs after it: no separate memory-bound delta pass
    float delta = 0.f;
    {
        const T* op = (const T*)p.o + (int64_t)b * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)qrow * p.o_ss + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const v8 of = *(const v8*)(op + 16 * ks);
#pragma unroll
            for (int e = 0; e < 8; ++e) delta += (float)of[e] * (float)gf[ks][e];
        }
        delta = row_pair_sum(delta);
        if (h == 0 && my_q < p.Sq) p.delta[stat] = delta;
    }
    const bool dead = !(lse > -INFINITY) || my_q >= p.Sq;
    const float lse2 = dead ? INFINITY : lse * 1.4426950408889634f;   // dead rows: exp2(x - inf) = 0 without a select
    const float c = p.scale_log2;
    f32x16 negdelta;                         // -delta of this lane's row in every register: the dP accumulator's start
#pragma unroll
    for (int e = 0; e < 16; ++e) negdelta[e] = -delta;
    asm volatile("" : "+v"(negdelta));

    TileDma<D, NW> dma;
    dma.init(wave, lane, p.k_ss, p.v_ss);
    TileRead<T, D> rk;                       // K image: row reads (S^T) and transposed reads (dQ^T)
    rk.init(lane, smem_base);

    f32x16 acc[DB];                          // dQ^T[d][q]
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    auto compute = [&](auto bufc, int key_base) {
        constexpr int BOFF = decltype(bufc)::value * BUF_BYTES;
        f32x16 s[2], dp[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
        // 4 KS operand fragments (K, V alternating) through a PF-deep register ring, order pinned with
        // sched_group_barrier: hipcc otherwise emits read -> wait -> MFMA on one 4-register buffer
        constexpr int NOP = 4 * KS, PF = 4;
        auto frag = [&](int i) {     // step i: kb = i / (2 KS), ks = (i / 2) % KS, K (even) or V (odd)
            return *(const lds_v8*)(uintptr_t)(rk.row_off[(i >> 1) % KS] + BOFF + (i & 1) * TILE_BYTES + (i / (2 * KS)) * HALF_TILE);
        };
        v8 afr[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) afr[i] = frag(i);
#pragma unroll
        for (int i = 0; i < NOP; ++i) {
            const int kb = i / (2 * KS), ks = (i >> 1) % KS;
            if ((i & 1) == 0) s[kb] = E::mfma(afr[i % PF], qf[ks], s[kb]);                                // S^T[key][q]
            else dp[kb] = E::mfma(afr[i % PF], gf[ks], ks == 0 ? negdelta : dp[kb]);                        // dP^T[key][q] - delta[q]
            if (i + PF < NOP) afr[i % PF] = frag(i + PF);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
        for (int i = 0; i < NOP - PF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);
        // masks only where a tile crosses the diagonal or the key tail (wave-uniform, a real branch); mask words: only where a row hides a key
        const bool word_all = mwrow && __builtin_amdgcn_ballot_w64(cur_bits != ~0ull) == 0;
        if ((KMASK && !word_all) || p.keymask || (key_base + BLOCK_N > kv_len) || (CAUSAL && key_base + BLOCK_N - 1 > wave_q0)) {
            asm volatile("" ::: "memory");
            // key-only mask: the lane's 32 keys of the tile are 8 groups of 4 consecutive keys = 8 dword loads (the same for the 32
            // lanes of a half-wave), where the element-mask path reads a byte per score.  Keys past Sk: the key < kv_len test.
            uint32_t kmw[2][4];
            const bool dwords = !mwrow && (p.keymask || (KMASK && p.mask_dw));
            if (dwords) {
                const uint8_t* kmrow = p.keymask ? p.keymask + (int64_t)b * p.km_sb : mrow;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        kmw[kb][g] = *(const uint32_t*)(kmrow + min(key_base + 32 * kb + 8 * g + 4 * h, p.Sk - 4));
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = key_base + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * h;
                    bool ok = key < kv_len;
                    if (CAUSAL) ok = ok && (key <= my_q);
                    if (mwrow) ok = ok && (((cur_bits >> (32 * kb + 4 * h + (e & 3) + 8 * (e >> 2))) & 1ull) != 0ull);
                    else if (dwords) ok = ok && (((kmw[kb][e >> 2] >> (8 * (e & 3))) & 0xFFu) != 0);
                    else if (KMASK) ok = ok && (mrow[(int64_t)min(key, p.Sk - 1) * p.m_sk] != 0);
                    s[kb][e] = ok ? s[kb][e] : -INFINITY;
                }
        }
        // dS^T = P^T o (dP^T - delta), P^T = exp2(c S^T - lse2)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = fast_exp2(__builtin_fmaf(s[kb][e], c, -lse2)) * dp[kb][e];
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float dsx[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) dsx[e] = s[kb][8 * s2 + e];
                const v8 ds = pack8<T>(dsx);
                constexpr int S2I = (D == 128) ? 0 : 1;
                const int koffs = BOFF + kb * HALF_TILE + ((D == 128) ? s2 * 16 * 256 : 0);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const v4 lo = E::tr_read((const lds_char*)(uintptr_t)(rk.tr_off[s2 * S2I][db][0] + koffs));
                    const v4 hi4 = E::tr_read((const lds_char*)(uintptr_t)(rk.tr_off[s2 * S2I][db][1] + koffs));
                    v8 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = lo[e];
                        a[4 + e] = hi4[e];
                    }
                    acc[db] = E::mfma(a, ds, acc[db]);
                }
            }
    };
    auto step = [&](auto bufc, int j) {
        constexpr int BUF = decltype(bufc)::value;
        bool seen = true;
        if constexpr (KMASK) {
            if (mwrow) {
                cur_bits = next_bits;
                if (j + 1 < nt) next_bits = mwrow[j + 1];
                seen = __builtin_amdgcn_ballot_w64(cur_bits != 0ull) != 0;      // no row of this wave sees a key of the tile: nothing to add
            }
        }
        if (j + 1 < nt)
            dma.issue(wave, j + 1, kp, p.k_ss, k_slab, smem_base + (BUF ^ 1) * BUF_BYTES, vp, p.v_ss, v_slab,
                      smem_base + (BUF ^ 1) * BUF_BYTES + TILE_BYTES);
        if (seen && j * BLOCK_N < wave_kv_end) compute(bufc, j * BLOCK_N);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    };
    if (nt > j0) dma.issue(wave, j0, kp, p.k_ss, k_slab, smem_base, vp, p.v_ss, v_slab, smem_base + TILE_BYTES);
    if constexpr (KMASK) {
        if (mwrow && nt > j0) next_bits = mwrow[j0];
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        asm volatile("" : "+v"(qf[ks]));
        asm volatile("" : "+v"(gf[ks]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int j = j0; j < nt; j += 2) {
        step(IC<0>{}, j);
        if (j + 1 < nt) step(IC<1>{}, j + 1);
    }
    if constexpr (sizeof(OT) == 2) {     // LDS is free after the loop's last barrier
        char* g0 = (char*)((OT*)p.dq + (int64_t)b * p.dq_sb + (int64_t)hh * p.dq_sh + (int64_t)wave_q0 * p.dq_ss);
        store_tile_rows_via_lds<T, D>(acc, p.scale, smem_base + wave * (32 * D * 2), lane, g0, p.dq_ss * 2, p.Sq - wave_q0);
    } else if (my_q < p.Sq) {
        OT* orow = (OT*)p.dq + (int64_t)b * p.dq_sb + (int64_t)hh * p.dq_sh + (int64_t)my_q * p.dq_ss;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = acc[db][4 * g + e] * p.scale;
                *(f32x4*)(orow + db * 32 + 8 * g + 4 * h) = w;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dK, dV: key-stationary.  Workgroup = 4 waves x 32 keys = 128 keys; LDS stage = [Q image | dO image] (64 rows).
template <typename T, int D, bool CAUSAL, bool KMASK, typename OT>
__global__ __launch_bounds__(256, 2) void fa3_bwd_dkdv_kernel(const BwdParams p) {
    using E = Elem<T>;
    using v8 = typename E::v8;
    using v4 = typename E::v4;
    typedef __attribute__((address_space(3))) v8 lds_v8;
    constexpr int NW = 4, BLOCK_K = 128, KS = D / 16, DB = D / 32;
    constexpr int TILE_BYTES = BLOCK_N * D * 2, BUF_BYTES = 2 * TILE_BYTES, HALF_TILE = TILE_BYTES / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // grouped-query heads: a workgroup owns a key block of ONE K/V head and streams the Q / dO tiles of all its G query heads, one head
    // after the other, into the same dK / dV accumulators (the sum over the group happens in registers, nothing is expanded in memory)
    const int G = p.kv_group, HK = p.H / G;
    const int BH = p.B * HK, n = blockIdx.x, krank = n / BH, bh = n - krank * BH;
    const int kblk = krank;                      // causal: low key blocks see the most query rows -> first
    const int b = bh / HK, hh = bh - b * HK;     // hh: the K/V head
    const int k0 = kblk * BLOCK_K, wave_k0 = k0 + wave * 32, my_key = wave_k0 + r;

    int kv_len = p.Sk;
    if (p.seqlens_k) kv_len = min(kv_len, max(p.seqlens_k[b], 0));
    // (a masked key behaves like one past kv_len: its column feeds only its own dK / dV, which are stored as zeros)
    const bool key_ok = my_key < kv_len && (!p.keymask || p.keymask[(int64_t)b * p.km_sb + min(my_key, p.Sk - 1)] != 0);
    // query tiles this block needs: causal -> rows >= k0
    int t_first = CAUSAL ? (k0 / BLOCK_N) : 0;
    int nt = (p.Sq + BLOCK_N - 1) / BLOCK_N;
    if constexpr (KMASK) {
        if (p.rg_col) {       // element mask: the row tiles in which any key of this block is visible to any query head of the group
            int lo = nt, hi = -1;
            for (int gq = 0; gq < G; ++gq) {
                const int* rg = p.rg_col + 2 * RANGE_PARTS * ((int64_t)b * p.crg_sb + (int64_t)(hh * G + gq) * p.crg_sh + kblk);
                lo = min(lo, rg[2 * (lane & (RANGE_PARTS - 1))]);
                hi = max(hi, rg[2 * (lane & (RANGE_PARTS - 1)) + 1]);
            }
#pragma unroll
            for (int off = RANGE_PARTS / 2; off >= 1; off >>= 1) {
                lo = min(lo, __shfl_xor(lo, off));
                hi = max(hi, __shfl_xor(hi, off));
            }
            nt = max(min(nt, __builtin_amdgcn_readfirstlane(hi) + 1), 0);
            t_first = max(t_first, min(__builtin_amdgcn_readfirstlane(lo), nt));
        }
    }

    const char* qp0 = (const char*)((const T*)p.q + (int64_t)b * p.q_sb + (int64_t)hh * G * p.q_sh);       // query head hh * G + gq: + gq * q_sh
    const char* gp0 = (const char*)((const T*)p.dout + (int64_t)b * p.do_sb + (int64_t)hh * G * p.do_sh);
    const T* kp = (const T*)p.k + (int64_t)b * p.k_sb + (int64_t)hh * p.k_sh;
    const T* vp = (const T*)p.v + (int64_t)b * p.v_sb + (int64_t)hh * p.v_sh;
    const int64_t q_slab = ((int64_t)(p.Sq - 1) * p.q_ss + D) * 2, g_slab = ((int64_t)(p.Sq - 1) * p.do_ss + D) * 2;
    const float* lse_p0 = p.lse + ((int64_t)b * p.H + hh * G) * p.Sq;        // + gq * Sq
    const float* del_p0 = p.delta + ((int64_t)b * p.H + hh * G) * p.Sq;
    // per-row constants of a 64-row tile, staged through LDS beside the tile: st[buf][0][64] = -lse/scale (or -inf
    // for rows that do not exist / are fully masked), st[buf][1][64] = -delta.  Thread t < 64 moves row t.
    constexpr int STAT_BASE = 2 * BUF_BYTES;
    typedef __attribute__((address_space(3))) float lds_float;
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
    float st_l = 0.f, st_d = 0.f;
    auto stat_load = [&](int gq, int j) {
        if (tid < BLOCK_N) {
            const int qi = j * BLOCK_N + tid;
            const bool live = qi < p.Sq;
            const float l = live ? lse_p0[(int64_t)gq * p.Sq + qi] : -INFINITY;
            st_l = (l > -INFINITY) ? -l / p.scale : -INFINITY;
            st_d = live ? -del_p0[(int64_t)gq * p.Sq + qi] : 0.f;
        }
    };
    auto stat_store = [&](int buf) {
        if (tid < BLOCK_N) {
            *(lds_float*)(uintptr_t)(smem_base + STAT_BASE + buf * 512 + tid * 4) = st_l;
            *(lds_float*)(uintptr_t)(smem_base + STAT_BASE + buf * 512 + 256 + tid * 4) = st_d;
        }
    };

    const int krow = min(my_key, p.Sk - 1);
    const uint8_t* mcol0 = KMASK ? p.mask + (int64_t)b * p.m_sb + (int64_t)hh * G * p.m_sh + (int64_t)krow * p.m_sk : nullptr;     // + gq * m_sh
    // the key's transposed mask words, one per 64-row tile (+ gq * cw_sh); the next tile's word is requested a step ahead (see the dQ kernel)
    const unsigned long long* mcw0 = (KMASK && p.mw_col) ? p.mw_col + (int64_t)b * p.cw_sb + (int64_t)hh * G * p.cw_sh + (int64_t)krow * p.ntq : nullptr;
    unsigned long long cur_cb = ~0ull, next_cb = ~0ull;
    v8 kf[KS], vf[KS];                       // K^T / V^T B-operand fragments: lane (key r, h) holds X[key][16ks+8h..+7]
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *(const v8*)(kp + (int64_t)krow * p.k_ss + 16 * ks + 8 * h);
        vf[ks] = *(const v8*)(vp + (int64_t)krow * p.v_ss + 16 * ks + 8 * h);
    }
    const float c = p.scale_log2;

    TileDma<D, NW> dma;
    dma.init(wave, lane, p.q_ss, p.do_ss);
    TileRead<T, D, D == 128> rq;
    rq.init(lane, smem_base);

    f32x16 dk[DB], dv[DB];                   // dK^T[d][key], dV^T[d][key]
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            dk[i][e] = 0.f;
            dv[i][e] = 0.f;
        }

    // one 32-row half of a 64-row Q/dO tile
    auto half = [&](auto bufc, auto hc, int q_base, int gq) {
        constexpr int BOFF = decltype(bufc)::value * BUF_BYTES;
        constexpr int HOFF = decltype(hc)::value * HALF_TILE;
        // accumulators start at the per-row constants: S - lse/scale , dP - delta  (rows = queries = registers)
        f32x16 s, dpv;
        {
            constexpr int SOFF = STAT_BASE + decltype(bufc)::value * 512 + decltype(hc)::value * 128;
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // rows 8g + 4h .. +3 of this half: one 16-byte broadcast read each
                const f32x4 lv = *(const lds_f32x4*)(uintptr_t)(smem_base + SOFF + (8 * g + 4 * h) * 4);
                const f32x4 dv4 = *(const lds_f32x4*)(uintptr_t)(smem_base + SOFF + 256 + (8 * g + 4 * h) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[4 * g + e] = lv[e];
                    dpv[4 * g + e] = dv4[e];
                }
            }
        }
        // element mask: the 16 bytes (column my_key, rows of this half) are requested BEFORE the MFMAs so that their latency runs
        // under them; read where they are used, each half paid a full memory round trip
        uint32_t mbyte[16];
        if constexpr (KMASK) {
            if (!mcw0) {
                const uint8_t* mcol = mcol0 + (int64_t)gq * p.m_sh;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int qi = min(q_base + (e & 3) + 8 * (e >> 2) + 4 * h, p.Sq - 1);
                    mbyte[e] = mcol[(int64_t)qi * p.m_sq];
                }
            }
        }
        // (a pinned prefetch ring as in the dQ kernel costs 20 spilled registers here and doubles the run time: the two
        //  workgroups per CU cover the read latency instead)
        static_for<KS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            const uint32_t ro = rq.template row_at<ks>();
            const v8 qa = *(const lds_v8*)(uintptr_t)(ro + BOFF + HOFF);
            const v8 ga = *(const lds_v8*)(uintptr_t)(ro + BOFF + TILE_BYTES + HOFF);
            s = E::mfma(qa, kf[ks], s);          // S[q][key] - lse/scale
            dpv = E::mfma(ga, vf[ks], dpv);      // dP[q][key] - delta
        });
        // causal: only a half that reaches above the diagonal of this wave's 32 keys needs the mask (wave-uniform branch).
        // Keys past kv_len need none: a lane is a key, its column feeds only its own dK/dV, zeroed at the store.
        if (CAUSAL && q_base < wave_k0 + 31) {
            asm volatile("" ::: "memory");
            const int t = my_key - q_base - 4 * h;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = (t <= (e & 3) + 8 * (e >> 2)) ? s[e] : -INFINITY;
        }
        if constexpr (KMASK) {                  // element mask: column my_key, rows of this half (words: bits 32 hc + ...; bytes: fetched before the MFMAs above)
            if (mcw0) {
                const uint32_t hb = (uint32_t)(cur_cb >> (32 * decltype(hc)::value + 4 * h));
                if (__builtin_amdgcn_ballot_w64((hb & 0x0F0F0F0Fu) != 0x0F0F0F0Fu) != 0) {      // (all sixteen visible for every key: nothing to do)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[e] = ((hb >> ((e & 3) + 8 * (e >> 2))) & 1u) ? s[e] : -INFINITY;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] = (mbyte[e] != 0) ? s[e] : -INFINITY;
            }
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]   (16 query rows per s2)
        static_for<2>([&](auto s2c) {
            constexpr int s2 = decltype(s2c)::value;
            constexpr int S2I = (D == 128) ? 0 : s2;
            constexpr int koffs = BOFF + HOFF + ((D == 128) ? s2 * 16 * 256 : 0);
            float px[8], dx[8];
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int e = 8 * s2 + e8;
                px[e8] = fast_exp2(s[e] * c);               // exp(scale*S - lse); rows staged as -inf and masked entries give 0
                dx[e8] = px[e8] * dpv[e];
            }
            const v8 pb = pack8<T>(px), dsb = pack8<T>(dx);
            static_for<DB>([&](auto dbc) {
                constexpr int db = decltype(dbc)::value;
                const uint32_t t0 = rq.template tr_at<S2I, db, 0>(), t1 = rq.template tr_at<S2I, db, 1>();
                v8 a;
                {
                    const v4 lo = E::tr_read((const lds_char*)(uintptr_t)(t0 + koffs + TILE_BYTES));
                    const v4 hi4 = E::tr_read((const lds_char*)(uintptr_t)(t1 + koffs + TILE_BYTES));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = lo[e];
                        a[4 + e] = hi4[e];
                    }
                }
                dv[db] = E::mfma(a, pb, dv[db]);
                {
                    const v4 lo = E::tr_read((const lds_char*)(uintptr_t)(t0 + koffs));
                    const v4 hi4 = E::tr_read((const lds_char*)(uintptr_t)(t1 + koffs));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = lo[e];
                        a[4 + e] = hi4[e];
                    }
                }
                dk[db] = E::mfma(a, dsb, dk[db]);
            });
        });
    };
    // the stream: query head gq = 0 .. G-1, tiles t_first .. nt-1 of each; (cur_g, cur_j) is the tile in the buffer being computed
    int cur_g = 0, cur_j = t_first;
    const int ntl = nt - t_first, total = ntl > 0 ? G * ntl : 0;
    auto step = [&](auto bufc, bool more) {
        constexpr int BUF = decltype(bufc)::value;
        int nxt_g = cur_g, nxt_j = cur_j + 1;
        if (nxt_j == nt) {
            nxt_j = t_first;
            ++nxt_g;
        }
        bool live0 = true, live1 = true;       // does any key of this wave see a row of the tile's halves? (element-mask words)
        if constexpr (KMASK) {
            if (mcw0) {
                cur_cb = next_cb;
                if (more) next_cb = mcw0[(int64_t)nxt_g * p.cw_sh + nxt_j];
                live0 = __builtin_amdgcn_ballot_w64((uint32_t)cur_cb != 0u) != 0;
                live1 = __builtin_amdgcn_ballot_w64((uint32_t)(cur_cb >> 32) != 0u) != 0;
            }
        }
        if (more) {
            dma.issue(wave, nxt_j, qp0 + (int64_t)nxt_g * p.q_sh * 2, p.q_ss, q_slab, smem_base + (BUF ^ 1) * BUF_BYTES,
                      gp0 + (int64_t)nxt_g * p.do_sh * 2, p.do_ss, g_slab, smem_base + (BUF ^ 1) * BUF_BYTES + TILE_BYTES);
            stat_load(nxt_g, nxt_j);             // lands under this tile's math, written to LDS before the barrier
        }
        const int q_base = cur_j * BLOCK_N;
        // causal: a 32-row half whose last row is above this wave's first key contributes nothing
        if (live0 && (!CAUSAL || q_base + 31 >= wave_k0)) half(bufc, IC<0>{}, q_base, cur_g);
        if (live1 && q_base + 32 < p.Sq && (!CAUSAL || q_base + 63 >= wave_k0)) half(bufc, IC<1>{}, q_base + 32, cur_g);
        if (more) stat_store(BUF ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        cur_g = nxt_g;
        cur_j = nxt_j;
    };
    if (total > 0) {
        dma.issue(wave, t_first, qp0, p.q_ss, q_slab, smem_base, gp0, p.do_ss, g_slab, smem_base + TILE_BYTES);
        stat_load(0, t_first);
        stat_store(0);
        if constexpr (KMASK) {
            if (mcw0) next_cb = mcw0[t_first];
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        asm volatile("" : "+v"(kf[ks]));
        asm volatile("" : "+v"(vf[ks]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < total; i += 2) {
        step(IC<0>{}, i + 1 < total);
        if (i + 1 < total) step(IC<1>{}, i + 2 < total);
    }
    if (!key_ok) {                              // keys in [kv_len, Sk): gradients are exactly zero
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                dk[i][e] = 0.f;
                dv[i][e] = 0.f;
            }
    }
    if constexpr (sizeof(OT) == 2) {     // LDS is free after the loop's last barrier; dK then dV through the same region
        const uint32_t lb = smem_base + wave * (32 * D * 2);
        char* gk = (char*)((OT*)p.dk + (int64_t)b * p.dk_sb + (int64_t)hh * p.dk_sh + (int64_t)wave_k0 * p.dk_ss);
        char* gv = (char*)((OT*)p.dv + (int64_t)b * p.dv_sb + (int64_t)hh * p.dv_sh + (int64_t)wave_k0 * p.dv_ss);
        store_tile_rows_via_lds<T, D>(dk, p.scale, lb, lane, gk, p.dk_ss * 2, p.Sk - wave_k0);
        store_tile_rows_via_lds<T, D>(dv, 1.0f, lb, lane, gv, p.dv_ss * 2, p.Sk - wave_k0);
    } else if (my_key < p.Sk) {
        OT* krow_o = (OT*)p.dk + (int64_t)b * p.dk_sb + (int64_t)hh * p.dk_sh + (int64_t)my_key * p.dk_ss;
        OT* vrow_o = (OT*)p.dv + (int64_t)b * p.dv_sb + (int64_t)hh * p.dv_sh + (int64_t)my_key * p.dv_ss;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * h;
                f32x4 wk, wv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    wk[e] = dk[db][4 * g + e] * p.scale;
                    wv[e] = dv[db][4 * g + e];
                }
                *(f32x4*)(krow_o + d) = wk;
                *(f32x4*)(vrow_o + d) = wv;
            }
    }
}

}  // namespace pfa

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

constexpr int WAVE_M = 32;    // Q rows per wave
constexpr int BLOCK_N = 64;   // keys per K/V tile
constexpr int block_m(int nw) { return nw * WAVE_M; }   // Q rows per workgroup (nw waves)

struct FwdParams {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;
    const int32_t* seqlens_k;
    const uint8_t* mask;       // optional u8 mask, 0 = masked, addressed [b][h][q][k] by the four byte strides below
    int64_t q_sb, q_sh, q_ss;
    int64_t k_sb, k_sh, k_ss;
    int64_t v_sb, v_sh, v_ss;
    int64_t o_sb, o_sh, o_ss;
    int64_t m_sb, m_sh, m_sq, m_sk;   // a 2-D [B,Sk] key mask is (stride, 0, 0, 1)
    int32_t B, H, Sq, Sk;
    int32_t nqblk;        // ceil(Sq / BLOCK_M)
    int32_t kv_group;     // query heads per K/V head (>= 1): query head h reads K/V head h / kv_group
    int32_t xcd_group;    // 4-wave kernel: walk the heads of an XCD in groups of this many (0 = all of them side by side)
    uint32_t magic_h, magic_g;   // 4-wave kernel: floor(2^32 / H) + 1 and floor(2^32 / kv_group) + 1 -- n / d == mulhi(n, magic) while n * d < 2^32
    float scale_log2;     // softmax_scale * log2(e)
    unsigned long long* dbg;   // VAR_STAMP only: [workgroup][wave][8] cycle sums
    // mask condensed to 64-bit words by fa3_maskbits_kernel: bit i of word (b, h, q, j) = key 64 j + i is visible to row q of head h
    // of batch b; word address mbits + b*mb_sb + h*mb_sh + q*mb_sq + j (strides in words, 0 = broadcast: a [B,Sk] key mask has
    // mb_sh = mb_sq = 0).  null: the mask is read a byte per score.
    const unsigned long long* mbits;
    int64_t mb_sb, mb_sh, mb_sq;
    // ... and, per 256 mask rows, the first and last tile holding ANY visible key (fa3_maskrange_kernel): pairs (lo, hi) at
    // mrange + 2 * (b*mr_sb + h*mr_sh + (mr_q ? q / 256 : 0)).  A Q block runs only tiles lo .. hi: a structured mask (a band, a
    // triangle, padding) skips what it hides -- no fetch, no barrier -- instead of masking it.  null: every tile runs.
    const int* mrange = nullptr;
    int64_t mr_sb = 0, mr_sh = 0;
    int32_t mr_q = 0;
};

template <typename T> struct Elem;
template <> struct Elem<__bf16> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // TIMING-ONLY stand-in (ABL_MFMA16): the same MACs as two v_mfma_f32_16x16x32 on 8 of the 16 accumulator registers
    static __device__ __forceinline__ f32x16 mfma16x2(v8 a, v8 b, f32x16 c) {
        f32x4 c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c[e] = c0[e];
            c[4 + e] = c1[e];
        }
        return c;
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
    static __device__ __forceinline__ f32x16 mfma16x2(v8 a, v8 b, f32x16 c) {
        f32x4 c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c[e] = c0[e];
            c[4 + e] = c1[e];
        }
        return c;
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

// Row maximum over one 16-register S block as ONE asm statement: hipcc puts an s_nop between consecutive
// single-instruction asm statements that depend on each other (15 per tile), and the VALU issue port -- not the
// MFMA pipe -- is what this kernel runs out of (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost').
__device__ __forceinline__ float max16_first(const f32x16& a) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3\n\t"
        "v_max3_f32 %0, %0, %4, %5\n\t"
        "v_max3_f32 %0, %0, %6, %7\n\t"
        "v_max3_f32 %0, %0, %8, %9\n\t"
        "v_max3_f32 %0, %0, %10, %11\n\t"
        "v_max3_f32 %0, %0, %12, %13\n\t"
        "v_max3_f32 %0, %0, %14, %15\n\t"
        "v_max_f32 %0, %0, %16"
        : "=&v"(r)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
          "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]));
    return r;
}
__device__ __forceinline__ float max16_next(float r, const f32x16& a) {
    asm("v_max3_f32 %0, %0, %1, %2\n\t"
        "v_max3_f32 %0, %0, %3, %4\n\t"
        "v_max3_f32 %0, %0, %5, %6\n\t"
        "v_max3_f32 %0, %0, %7, %8\n\t"
        "v_max3_f32 %0, %0, %9, %10\n\t"
        "v_max3_f32 %0, %0, %11, %12\n\t"
        "v_max3_f32 %0, %0, %13, %14\n\t"
        "v_max3_f32 %0, %0, %15, %16"
        : "+&v"(r)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
          "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]));
    return r;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// One v_max3_f32.  fmaxf() on MFMA results makes hipcc emit a canonicalising v_max_f32 x,x per operand
// (3 VALU per 2 values instead of 0.5); the scores are never sNaN, so the raw instruction is safe.
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// max(x[lane], x[lane ^ 32]) in every lane: the two lanes that share a query row.
__device__ __forceinline__ float row_pair_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// same, without the canonicalising v_max x,x hipcc adds around fmaxf (s_nop 1: VALU write -> permlane read)
__device__ __forceinline__ float row_pair_max_asm(float x) {
    float t;
    asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(x), "=&v"(t));
    return x;
}
__device__ __forceinline__ float row_pair_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Variant bits (compile-time), selected through pfa_fa3_args.flags >> 8 for A/B runs:
constexpr int VAR_DEFER_MAX = 1;   // T13: rescale O only when a row max grew by > 2^8 (else exact lazy rescale)
constexpr int VAR_SETPRIO = 2;     // s_setprio(1) around the MFMA clusters
constexpr int VAR_SCHED = 4;       // pin the QK^T read/MFMA interleave with sched_group_barrier
// timing-only ablations (results are WRONG on purpose; cdna guide section 7 "ablate"): bits 16..19
constexpr int ABL_NO_DMA = 1 << 23;       // never refill LDS after the prologue (stale K/V tiles)
constexpr int ABL_NO_BARRIER = 1 << 24;   // no per-tile barrier
constexpr int VAR_XCDG2 = 1 << 28;        // block order: per XCD, heads in groups of 2 (all Q blocks of a group run together)
constexpr int VAR_XCDG4 = 1 << 29;        // ... groups of 4
constexpr int ABL_AGPR_OPND = 1 << 30;    // QK^T: LDS fragments land in AGPRs (asm ds_read "=a"), MFMA A operand from AGPR
constexpr int ABL_AGPR_ACC = 1 << 27;     // QK^T accumulators forced into the accumulator half (inline-asm MFMA, "+a")
constexpr int ABL_NO_MFMA = 1 << 26;      // QK^T: LDS reads only (operands consumed by an empty asm)
constexpr int ABL_NO_LDS = 1 << 25;       // QK^T A operands from registers instead of LDS
constexpr int VAR_RING3 = 1 << 22;        // 3-slot K/V ring: DMA for tile j+2 issued at the END of tile j (before the barrier wait)
constexpr int VAR_QKIL = 1 << 20;         // QK^T: alternate the two key blocks (two independent accumulator chains)
constexpr int VAR_PF8 = 1 << 21;          // QK^T: operand reads 8 deep instead of 4
constexpr int ABL_NO_SOFTMAX = 1 << 16;   // P = bf16(S): no max, no exp, no row sum
constexpr int ABL_NO_PV = 1 << 17;        // skip the PV MFMAs and V reads
constexpr int ABL_NO_QK = 1 << 18;        // skip the QK^T MFMAs and K reads (S = stale registers)
constexpr int ABL_MFMA16 = (int)(1u << 31); // every 32x32x16 MFMA replaced by two 16x16x32 (same MACs, half the accumulator traffic): DVFS probe
constexpr int ABL_NO_EXP = 1 << 19;       // softmax without the v_exp (p = fma result)
constexpr int VAR_ALTPRIO = 32768; // s_setprio alternates between the wave halves every tile (with VAR_STAGE2: each half wins once per barrier)
constexpr int VAR_DMA4 = 16384;    // only waves 0..NW/2-1 (the older half, which waits at the barrier anyway) issue the LDS-DMA
constexpr int VAR_DIET = 8192;     // VALU diet: row max as one asm block (no s_nop between), opaque LDS addresses (no per-tile v_add), no packed adds
constexpr int VAR_YPRIO = 2048;    // static s_setprio 1 for the younger wave half (waves NW/2..NW-1)
constexpr int VAR_LATEDMA = 4096;  // waves NW/2.. issue their DMA pieces after QK^T instead of at the top of the tile
constexpr int VAR_STAGGER = 1024;  // waves 4-7 one phase behind waves 0-3 (fa3_fwd_stagger_kernel.h)
constexpr int VAR_STAMP = 512;     // DIAGNOSTIC build: s_memtime phase stamps into FwdParams.dbg (never quote its run time)
constexpr int VAR_STAGE2 = 256;    // two 64-key tiles per barrier (128 KiB of LDS): half the barriers, waves drift further apart
constexpr int VAR_LSUM = 128;      // row sums on the matrix pipe: one extra MFMA per k-step against an all-ones A operand
constexpr int VAR_BUFDMA = 64;     // LDS-DMA by buffer_load ... lds: SRD rebuilt per tile on the SALU, no per-tile VALU
constexpr int VAR_PIPE = 32;       // software-pipelined half-tile schedule (fa3_fwd_pipe_kernel.h)
constexpr int VAR_NW4 = 16;        // 4-wave workgroups of 128 Q rows, two resident per CU (independent barriers)
constexpr int VAR_GLDS = 8;        // K/V tiles by LDS-DMA (global_load_lds_dwordx4), swizzle on the source address
constexpr int VAR_DEFAULT = VAR_DEFER_MAX | VAR_SCHED | VAR_GLDS | VAR_BUFDMA | VAR_DIET;

template <int N> struct IC { static constexpr int value = N; };

// s_memtime stamp as ONE statement with its own lgkmcnt(0) (cdna guide section 7, in-kernel stamps)
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
typedef __amdgpu_buffer_rsrc_t srd_t;

// Second half of the LDS-staged row store (every epilogue: forward O, backward dQ / dK / dV): read a [32 rows][RB bytes] image
// (16-byte chunk index XOR row) back so that one store instruction covers 64 / CPRW WHOLE rows.  All reads are issued before the
// first store: written as one loop, hipcc sinks each ds_read_b128 into its store's predicated block and the epilogue pays 8
// serial LDS round trips (ds_read, s_waitcnt lgkmcnt(0), store, next ...).  rows_valid is wave-uniform.
template <int RB>
__device__ __forceinline__ void store_rows_from_lds(uint32_t lbase, int lane, char* grow0, int64_t row_stride_bytes, int rows_valid) {
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;
    constexpr int CPRW = RB / 16, RPI = 64 / CPRW, NI = 32 / RPI;      // chunks per row, rows per store instruction, instructions
    const int cc = lane & (CPRW - 1), lr = lane / CPRW;
    u32x4 x[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = RPI * i + lr;
        x[i] = *(const lds_u32x4_t*)(uintptr_t)(lbase + row * RB + ((cc ^ (row & (CPRW - 1))) << 4));
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(x[i]));
    char* g = grow0 + 16 * cc + (int64_t)lr * row_stride_bytes;
    if (rows_valid >= 32) {
#pragma unroll
        for (int i = 0; i < NI; ++i) *(u32x4*)(g + (int64_t)(RPI * i) * row_stride_bytes) = x[i];
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (RPI * i + lr < rows_valid) *(u32x4*)(g + (int64_t)(RPI * i) * row_stride_bytes) = x[i];
    }
}

// LDS-DMA piece through a buffer descriptor: lane address = SRD base + voff, out-of-range lanes (rows past the
// end of the K/V slab) deliver zeros, so ragged tails need no clamp.  s_nop 4: SGPR-written-by-SALU -> VMEM.
__device__ __forceinline__ void lds_dma16_buf(srd_t srd, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(srd), "s"(lds_dst)
        : "memory");
}

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses into LDS at lds_dst + 16*lane.
// Inline asm on purpose: with the builtin, hipcc treats the DMA as a possibly-aliasing LDS store and puts
// s_waitcnt vmcnt(0) in front of the next ds_read of the OTHER buffer, exposing the whole HBM latency every
// tile.  Hidden in asm the DMA is invisible to the waitcnt pass; the kernel retires it itself with a counted
// s_waitcnt vmcnt(N) in front of the barrier that publishes the tile (cdna guide section 5.7 item 1).
// M0 (LDS base of the DMA) is saved/restored inside the statement; s_nop 0 = the M0-write -> LDS-DMA wait state.
__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// u8 mask (0 = masked; byte strides, 0 = broadcast) -> one 64-bit word per mask row and 64-key tile.  The forward then reads
// ONE word per row and tile: all ones = no masking work, all zero = the tile is skipped, anything else = bit tests -- where
// reading the mask itself cost 32 scattered byte loads per lane and tile (2-6 x the run time of the unmasked problem).
// One wave per word: grid (ceil(nt / 4), rows = Qm, Bm * Hm) with Bm / Hm / Qm = the mask's own (un-broadcast) extents.
template <int UNUSED = 0>   // (a template only so that the three translation units including this header do not each define it)
__global__ __launch_bounds__(256) void fa3_maskbits_kernel(const uint8_t* m, int64_t sb, int64_t sh, int64_t sq, int64_t sk, int Hm, int Sk,
                                                          int nt, unsigned long long* out, int64_t ob, int64_t oh, int64_t oq) {
    const int lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6), q = blockIdx.y;
    const int b = blockIdx.z / Hm, hh = blockIdx.z - b * Hm;
    if (tile >= nt) return;
    const int key = tile * 64 + lane;
    const bool on = key < Sk && m[(int64_t)b * sb + (int64_t)hh * sh + (int64_t)q * sq + (int64_t)key * sk] != 0;
    const unsigned long long bits = __builtin_amdgcn_ballot_w64(on);
    if (lane == 0) out[(int64_t)b * ob + (int64_t)hh * oh + (int64_t)q * oq + tile] = bits;
}

// The same for keys contiguous in memory (m_sk = 1, rows / bases / Sk multiples of 16 bytes): a lane reads 16 mask bytes at once, four
// lanes make a word, a wave 16 words of one row -- the byte-per-lane form above runs at 0.4 TB/s of mask, a 16 MB [S,S] mask cost
// 43 us in front of a 280 us forward.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void fa3_maskbits16_kernel(const uint8_t* m, int64_t sb, int64_t sh, int64_t sq, int Hm, int Sk, int nt,
                                                            unsigned long long* out, int64_t ob, int64_t oh, int64_t oq) {
    const int lane = threadIdx.x & 63, grp = blockIdx.x * 4 + (threadIdx.x >> 6), q = blockIdx.y;     // grp: 16 tiles = 1024 keys
    const int b = blockIdx.z / Hm, hh = blockIdx.z - b * Hm;
    if (grp * 16 >= nt) return;
    const int key0 = grp * 1024 + 16 * lane;
    uint32_t bits = 0;
    if (key0 < Sk) {
        const u32x4 x = *(const u32x4*)(m + (int64_t)b * sb + (int64_t)hh * sh + (int64_t)q * sq + key0);
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int i = 0; i < 4; ++i) bits |= (((x[w] >> (8 * i)) & 0xFFu) != 0 ? 1u : 0u) << (4 * w + i);
    }
    unsigned long long word = (unsigned long long)bits << (16 * (lane & 3));
    word |= __shfl_xor(word, 1);
    word |= __shfl_xor(word, 2);
    const int tile = grp * 16 + (lane >> 2);
    if ((lane & 3) == 0 && tile < nt) out[(int64_t)b * ob + (int64_t)hh * oh + (int64_t)q * oq + tile] = word;
}

// first / last tile with a visible key per 256 mask rows (see FwdParams::mrange): RANGE_PARTS workgroups per granule and mask (batch, head),
// each leaves the pair of its share of the rows; the forward kernel takes the minimum / maximum of the parts
constexpr int RANGE_PARTS = 16;
template <int GRAN = 256>      // rows per granule: 256 (a Q block of the forward / dQ kernels) or 128 (a key block of the dK/dV kernel, on transposed words)
__global__ __launch_bounds__(256) void fa3_maskrange_kernel(const unsigned long long* words, int64_t ob, int64_t oh, int64_t oq, int Hm, int Qm,
                                                           int nt, int* out, int ngran) {
    const int g = blockIdx.x / RANGE_PARTS, part = blockIdx.x - g * RANGE_PARTS, b = blockIdx.y / Hm, hh = blockIdx.y - b * Hm;
    const int r0 = min(Qm, g * GRAN + part * (GRAN / RANGE_PARTS)), r1 = min(Qm, r0 + GRAN / RANGE_PARTS);
    const unsigned long long* w = words + (int64_t)b * ob + (int64_t)hh * oh;
    int lo = nt, hi = -1;
    for (int idx = threadIdx.x; idx < (r1 - r0) * nt; idx += 256) {
        const int row = r0 + idx / nt, t = idx - (idx / nt) * nt;
        if (w[(int64_t)row * oq + t] != 0ull) {
            lo = min(lo, t);
            hi = max(hi, t);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off));
        hi = max(hi, __shfl_xor(hi, off));
    }
    __shared__ int slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) {
        slo[threadIdx.x >> 6] = lo;
        shi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int* o = out + 2 * ((((int64_t)b * Hm + hh) * ngran + g) * RANGE_PARTS + part);
        o[0] = min(min(slo[0], slo[1]), min(slo[2], slo[3]));
        o[1] = max(max(shi[0], shi[1]), max(shi[2], shi[3]));
    }
}

// The words transposed, for the key-stationary dK/dV kernel: bit i of word (b, h, key, tq) = row 64 tq + i sees `key`.  One wave per
// (64 rows, 64 keys): a lane holds its row's word of the key tile, 64 ballots turn the bit matrix over.  rows = the mask's own row
// extent (1: every row alike), Sq = the problem's.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void fa3_maskbitsT_kernel(const unsigned long long* roww, int64_t ob, int64_t oh, int64_t oq, int Hm, int rows,
                                                           int Sq, int Sk, int nt, int ntq, unsigned long long* colw) {
    const int lane = threadIdx.x & 63, kt = blockIdx.x * 4 + (threadIdx.x >> 6), tq = blockIdx.y;
    const int b = blockIdx.z / Hm, hh = blockIdx.z - b * Hm;
    if (kt >= nt) return;
    const int row = 64 * tq + lane;
    const unsigned long long w = row < Sq ? roww[(int64_t)b * ob + (int64_t)hh * oh + (int64_t)(rows > 1 ? row : 0) * oq + kt] : 0ull;
    unsigned long long kept = 0ull;
    for (int c = 0; c < 64; ++c) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(((w >> c) & 1ull) != 0ull);
        if (lane == c) kept = bal;
    }
    const int key = 64 * kt + lane;
    if (key < Sk) colw[(((int64_t)b * Hm + hh) * Sk + key) * ntq + tq] = kept;
}

template <typename T, int D, bool CAUSAL, bool SPLITP, bool KMASK, int VAR, typename OT>
__global__ __launch_bounds__(((VAR & VAR_NW4) ? 4 : 8) * 64, 2) void fa3_fwd_kernel(const FwdParams p) {
    constexpr int NW = (VAR & VAR_NW4) ? 4 : 8;
    constexpr int NTHREADS = NW * 64;
    constexpr int BLOCK_M = NW * WAVE_M;
    using E = Elem<T>;
    using v8 = typename E::v8;
    using v4 = typename E::v4;
    typedef __attribute__((address_space(3))) v8 lds_v8;
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    constexpr int KS = D / 16;                // k-steps of the QK^T product
    constexpr int DB = D / 32;                // 32-wide d blocks of the PV product
    constexpr int CPR = D / 8;                // 16-byte chunks per key row
    constexpr int TILE_BYTES = BLOCK_N * D * 2;
    constexpr int BUF_BYTES = 2 * TILE_BYTES; // K image + V image
    constexpr int HALF_TILE = TILE_BYTES / 2; // 32 keys
    constexpr int CHUNKS_PER_THREAD = (BLOCK_N * CPR) / NTHREADS;
    static_assert(CHUNKS_PER_THREAD >= 1, "tile too small for the workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* const smem_l = (lds_char*)smem;   // [buf][K|V][TILE_BYTES]
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem_l;   // LDS byte address (wave-uniform)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    // ---- block -> (q block, batch*head): heaviest (longest causal row) blocks first ----------------
    const int BH = p.B * p.H;
    const int n = blockIdx.x;
    int qrank = n / BH;
    int bh = n - qrank * BH;
    if constexpr ((VAR & VAR_XCDG2) || (VAR & VAR_XCDG4)) {
        // blocks n, n+8, ... share an XCD (round-robin dispatch; speed only).  Within an XCD walk the heads in
        // groups of G so that the ~32 resident workgroups are G heads x many Q blocks: every K/V tile is then
        // fetched into that XCD's L2 once and re-read by the other Q blocks of the head while it is still there.
        constexpr int G = (VAR & VAR_XCDG2) ? 2 : 4;
        const int hpx = BH / 8;                       // heads per XCD
        if ((BH % 8) == 0 && (hpx % G) == 0) {
            const int xcd = n & 7, idx = n >> 3;
            const int per_group = p.nqblk * G;
            const int g = idx / per_group, within = idx - g * per_group;
            qrank = within / G;
            bh = xcd + 8 * (g * G + (within - qrank * G));
        }
    }
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
    int nt = (kv_end + BLOCK_N - 1) / BLOCK_N;
    int j0 = 0;                              // first tile of the block (0 unless the mask's tile range says otherwise)
    if constexpr (KMASK && !(VAR & (VAR_RING3 | VAR_STAGE2))) {
        if (p.mrange) {
            const int* rg = p.mrange + 2 * RANGE_PARTS * ((int64_t)b * p.mr_sb + (int64_t)hh * p.mr_sh + (p.mr_q ? (q0 >> 8) : 0));
            int lo = rg[2 * (lane & (RANGE_PARTS - 1))], hi = rg[2 * (lane & (RANGE_PARTS - 1)) + 1];
#pragma unroll
            for (int off = RANGE_PARTS / 2; off >= 1; off >>= 1) {
                lo = min(lo, __shfl_xor(lo, off));
                hi = max(hi, __shfl_xor(hi, off));
            }
            lo = __builtin_amdgcn_readfirstlane(lo);
            hi = __builtin_amdgcn_readfirstlane(hi);
            nt = min(nt, hi + 1);                                    // nothing visible past tile hi (hi = -1: nothing at all -> zeros, LSE -inf)
            j0 = min(lo, max(nt, 0)) & ~1;                           // (even: the two LDS buffers keep their parity)
            nt = max(nt, 0);
        }
    }

    const T* __restrict__ qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)hh * p.q_sh;
    const T* __restrict__ kp = (const T*)p.k + (int64_t)b * p.k_sb + (int64_t)(hh / p.kv_group) * p.k_sh;
    const T* __restrict__ vp = (const T*)p.v + (int64_t)b * p.v_sb + (int64_t)(hh / p.kv_group) * p.v_sh;
    const uint8_t* __restrict__ kmp =
        KMASK ? p.mask + (int64_t)b * p.m_sb + (int64_t)hh * p.m_sh + (int64_t)min(my_q, p.Sq - 1) * p.m_sq : nullptr;

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
    const T* kld[CHUNKS_PER_THREAD];
    const T* vld[CHUNKS_PER_THREAD];
    int st_key[CHUNKS_PER_THREAD];
#pragma unroll
    for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
        const int c = tid + i * NTHREADS;
        st_key[i] = c / CPR;
        st_off[i] = tile_off<D>(st_key[i], c % CPR);
        kld[i] = kp + (c % CPR) * 8;
        vld[i] = vp + (c % CPR) * 8;
    }
    auto load_tile = [&](int j) {
#pragma unroll
        for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
            const int key = min(j * BLOCK_N + st_key[i], p.Sk - 1);
            kreg[i] = *(const u32x4*)(kld[i] + (int64_t)key * p.k_ss);
            vreg[i] = *(const u32x4*)(vld[i] + (int64_t)key * p.v_ss);
        }
    };
    auto store_tile = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
            *(lds_u32x4*)(smem_l + st_off[i] + BUF * BUF_BYTES) = kreg[i];
            *(lds_u32x4*)(smem_l + st_off[i] + BUF * BUF_BYTES + TILE_BYTES) = vreg[i];
        }
    };

    // ---- LDS-DMA staging (VAR_GLDS): one wave-instruction moves 64 x 16 B = 1 KiB into LINEAR LDS, so the
    // XOR swizzle is applied to the per-lane SOURCE chunk (guide rule 21).  Wave w issues pieces w, w+8, ...
    // piece i covers LDS rows 4i..4i+3 (256 B each); lane l -> row 4i + (l>>4), stored chunk l&15.
    constexpr int PIECES = TILE_BYTES / 1024;            // 16 (D=128) or 8 (D=64)
    constexpr int NDW = (VAR & VAR_DMA4) ? NW / 2 : NW;  // waves that issue DMA
    constexpr int PPW = PIECES / NDW;                    // pieces per issuing wave
    int dma_key[PPW];
    int dma_col;                                         // element offset of the source chunk in its key row
    {
        const int dw = wave % NDW;                       // issuing slot of this wave
        const int R0 = 4 * dw + (lane >> 4);             // LDS row of piece `dw`
        const int sw = ((R0 & 3) << 2) | ((R0 >> 2) & 3);   // rows 4*NDW apart share the swizzle term (NDW = 4, 8)
        const int cc = (lane & 15) ^ sw;                 // logical chunk stored at this lane's position
        if constexpr (D == 128) {
            dma_col = cc * 8;
#pragma unroll
            for (int t = 0; t < PPW; ++t) dma_key[t] = R0 + 4 * NDW * t;
        } else {
            dma_col = (cc & 7) * 8;
#pragma unroll
            for (int t = 0; t < PPW; ++t) dma_key[t] = 2 * (R0 + 4 * NDW * t) + (cc >> 3);
        }
    }
    // buffer-descriptor form: per-lane byte offsets are loop invariant, the tile steps the SRD base (SALU only)
    uint32_t kvoff[PPW], vvoff[PPW];
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
        kvoff[t] = (uint32_t)(dma_key[t] * (int)p.k_ss + dma_col) * 2u;
        vvoff[t] = (uint32_t)(dma_key[t] * (int)p.v_ss + dma_col) * 2u;
    }
    const int64_t k_slab = ((int64_t)(p.Sk - 1) * p.k_ss + D) * 2;   // bytes of this (b,h) K slab
    const int64_t v_slab = ((int64_t)(p.Sk - 1) * p.v_ss + D) * 2;
    auto dma_tile = [&](auto bufc, int j) {
        constexpr int BUF = decltype(bufc)::value;
        if constexpr (VAR & VAR_DMA4) {
            if (wave >= NDW) return;   // wave-uniform: the other half never touches the address path
        }
        if constexpr (VAR & VAR_BUFDMA) {
            const int64_t kstep = (int64_t)j * BLOCK_N * p.k_ss * 2, vstep = (int64_t)j * BLOCK_N * p.v_ss * 2;
            const srd_t ksrd = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const char*)kp + kstep), 0, (int)max((int64_t)0, k_slab - kstep), 0x00020000);
            const srd_t vsrd = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const char*)vp + vstep), 0, (int)max((int64_t)0, v_slab - vstep), 0x00020000);
#pragma unroll
            for (int t = 0; t < PPW; ++t) {
                const uint32_t kd = smem_base + BUF * BUF_BYTES + (wave % NDW + NDW * t) * 1024;
                lds_dma16_buf(ksrd, kvoff[t], kd);
                lds_dma16_buf(vsrd, vvoff[t], kd + TILE_BYTES);
            }
        } else {
#pragma unroll
            for (int t = 0; t < PPW; ++t) {
                const int key = min(j * BLOCK_N + dma_key[t], p.Sk - 1);
                const uint32_t kd = smem_base + BUF * BUF_BYTES + (wave + NW * t) * 1024;
                lds_dma16(kp + (int64_t)key * p.k_ss + dma_col, kd);
                lds_dma16(vp + (int64_t)key * p.v_ss + dma_col, kd + TILE_BYTES);
            }
        }
    };

    // ---- per-lane LDS read offsets, loop invariant; (buffer, key block, k-step) become immediates ----------
    // K row read (kb, ks): key = 32 kb + r, chunk 2 ks + h.   +32 keys leaves the swizzle term unchanged.
    uint32_t koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = smem_base + tile_off<D>(r, 2 * ks + h);   // absolute LDS address
    // opaque to the optimiser: otherwise it keeps row and chunk parts apart and re-adds them every tile (25 v_add_u32)
    if constexpr (VAR & VAR_DIET) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(koff[ks]));
    }
    // V transposed read (kb, s2, db, hi8): lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of the
    // 4-key x 16-d block at key0 = 32 kb + 16 s2 + 4 h (+8), d0 = 32 db + 16 ((lane>>4)&1)
    const int g1 = (lane >> 4) & 1;
    const int tq = (lane & 15) >> 2;
    const int tp = lane & 3;
    constexpr int NS2 = (D == 128) ? 1 : 2;   // D=64: two keys per LDS row, the swizzle term depends on s2
    uint32_t voff[NS2][DB][2];
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2)
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
            {
                voff[s2][db][hi] = smem_base + tile_off<D>(16 * s2 + 4 * h + tq + 8 * hi, db * 4 + 2 * g1 + (tp >> 1)) + 8 * (tp & 1);
                if constexpr (VAR & VAR_DIET) asm volatile("" : "+v"(voff[s2][db][hi]));
            }

    auto MF = [](v8 a, v8 b, f32x16 acc) {
        if constexpr (VAR & ABL_MFMA16) return E::mfma16x2(a, b, acc);
        else return E::mfma(a, b, acc);
    };
    f32x16 o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    float m_run = -1e30f;   // reference max of the exponentials, raw score units
    float l_run = 0.f;      // this lane's share of the row sum (VALU form)
    f32x16 lacc;            // VAR_LSUM: every register = the row sum of this lane's query row (matrix-pipe form)
#pragma unroll
    for (int e = 0; e < 16; ++e) lacc[e] = 0.f;
    v8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;
    const float c = p.scale_log2;
    const float thr = (VAR & VAR_DEFER_MAX) ? 8.0f / c : 0.0f;   // raw-score headroom before a rescale
    float m_thr = -1e30f, mc = -1e30f * c;                       // m_run + thr and m_run * c, updated with m_run

    unsigned long long st_qk_end = 0;
    // ---- one K/V tile: S^T = K Q^T, online softmax, O^T += V^T P^T -------------------------------------------
    // KMASK + mbits: the row's mask word of the current tile, fetched one tile ahead (consumed after the previous step's
    // s_waitcnt vmcnt(0), so the wait hipcc puts in front of its use never holds up the K/V prefetch issued behind it)
    unsigned long long tile_bits = ~0ull, next_bits = ~0ull;
    bool tile_all_ones = true;
    const unsigned long long* mrow = nullptr;
    if constexpr (KMASK) {
        if (p.mbits) mrow = p.mbits + (int64_t)b * p.mb_sb + (int64_t)hh * p.mb_sh + (int64_t)min(my_q, p.Sq - 1) * p.mb_sq;
    }
    auto fetch_bits = [&](int j) {
        if constexpr (KMASK) {
            if (mrow) next_bits = mrow[min(j, (p.Sk + BLOCK_N - 1) / BLOCK_N - 1)];
        }
    };
    auto tile_live = [&](int j) {              // does this wave compute tile j?  (wave-uniform)
        if (!(j * BLOCK_N < wave_kv_end)) return false;
        if constexpr (KMASK) {
            if (mrow) {
                tile_bits = next_bits;
                fetch_bits(j + 1);
                tile_all_ones = __builtin_amdgcn_ballot_w64(tile_bits != ~0ull) == 0;
                return __builtin_amdgcn_ballot_w64(tile_bits != 0ull) != 0;   // no key of the tile is visible to any row: skip it
            }
        }
        return true;
    };
    auto compute_tile = [&](auto bufc, int key_base, int jnext = -1) {
        constexpr int BUF = decltype(bufc)::value;
        if constexpr (VAR & VAR_ALTPRIO) {   // the half that lost the issue arbitration on the last tile wins this one
            if (((wave >= NW / 2) ? 1 : 0) ^ ((key_base / BLOCK_N) & 1)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        const lds_char* kimg = (const lds_char*)(uintptr_t)(BUF * BUF_BYTES);   // koff[]/voff[] carry the LDS base
        const lds_char* vimg = kimg + TILE_BYTES;

        // S^T = K Q^T: 16 MFMAs (2 key blocks x KS k-steps), A fragments prefetched PF deep from LDS so the
        // ds_read latency hides behind the MFMAs already issued (hipcc otherwise emits read -> wait -> mfma)
        f32x16 s[2];
        if constexpr (!(VAR & ABL_AGPR_ACC)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
        }
        constexpr int NQK = (VAR & ABL_NO_QK) ? 0 : 2 * KS;
        // VAR_PF8 alone: 8 deep; VAR_PF8 + VAR_SETPRIO: all 2*KS fragments first ("load cluster, then MFMA cluster")
        constexpr int PF = (VAR & ABL_NO_QK) ? 0 : ((VAR & VAR_PF8) ? ((VAR & VAR_SETPRIO) ? 2 * KS : 8) : 4);
        v8 afr[2 * KS];
        // step i -> (key block, k-step): blocked (kb = i / KS) or interleaved (kb = i & 1)
        auto kb_of = [](int i) { return (VAR & VAR_QKIL) ? (i & 1) : (i / KS); };
        auto ks_of = [](int i) { return (VAR & VAR_QKIL) ? (i >> 1) : (i % KS); };
        if constexpr (VAR & ABL_AGPR_OPND) {
            // timing experiment: does the LDS->register stream overlap the MFMAs when it lands in the accumulator half?
            auto rd = [&](int i, v8& dst) {
                const uint32_t addr = koff[ks_of(i)] + BUF * BUF_BYTES + kb_of(i) * HALF_TILE;
                asm volatile("ds_read_b128 %0, %1" : "=a"(dst) : "v"(addr));
            };
#pragma unroll
            for (int i = 0; i < 4; ++i) rd(i, afr[i]);
#pragma unroll
            for (int i = 0; i < 2 * KS; ++i) {
                const int left = (2 * KS - 1 - i) < 3 ? (2 * KS - 1 - i) : 3;
                if (left == 3) asm volatile("s_waitcnt lgkmcnt(3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[kb_of(i)]) : "a"(afr[i % 4]), "v"(qf[ks_of(i)]));
                else if (left == 2) asm volatile("s_waitcnt lgkmcnt(2)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[kb_of(i)]) : "a"(afr[i % 4]), "v"(qf[ks_of(i)]));
                else if (left == 1) asm volatile("s_waitcnt lgkmcnt(1)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[kb_of(i)]) : "a"(afr[i % 4]), "v"(qf[ks_of(i)]));
                else asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[kb_of(i)]) : "a"(afr[i % 4]), "v"(qf[ks_of(i)]));
                if (i + 4 < 2 * KS) rd(i + 4, afr[i % 4]);
            }
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(s[0]), "+v"(s[1]));
        } else {
#pragma unroll
        for (int i = 0; i < PF; ++i) afr[i] = *(const lds_v8*)(kimg + koff[ks_of(i)] + kb_of(i) * HALF_TILE);
        if (VAR & VAR_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NQK; ++i) {
            if constexpr (VAR & ABL_NO_LDS) {
                s[kb_of(i)] = E::mfma(qf[(ks_of(i) + 1) % KS], qf[ks_of(i)], s[kb_of(i)]);
            } else if constexpr (VAR & ABL_AGPR_ACC) {
                if (ks_of(i) == 0)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(s[kb_of(i)]) : "v"(afr[i % PF]), "v"(qf[ks_of(i)]));
                else
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(s[kb_of(i)]) : "v"(afr[i % PF]), "v"(qf[ks_of(i)]));
                if (i + PF < NQK) afr[i % PF] = *(const lds_v8*)(kimg + koff[ks_of(i + PF)] + kb_of(i + PF) * HALF_TILE);
            } else if constexpr (VAR & ABL_NO_MFMA) {
                asm volatile("" ::"v"(afr[i % PF]));
                if (i + PF < NQK) afr[i % PF] = *(const lds_v8*)(kimg + koff[ks_of(i + PF)] + kb_of(i + PF) * HALF_TILE);
            } else {
                s[kb_of(i)] = MF(afr[i % (PF ? PF : 1)], qf[ks_of(i)], s[kb_of(i)]);
                if (i + PF < NQK) afr[i % (PF ? PF : 1)] = *(const lds_v8*)(kimg + koff[ks_of(i + PF)] + kb_of(i + PF) * HALF_TILE);
            }
        }
        }
        if (VAR & VAR_SETPRIO) __builtin_amdgcn_s_setprio(0);
        if constexpr (VAR & ABL_NO_QK) {
            asm volatile("" : "+v"(s[0]), "+v"(s[1]));
        }
        if ((VAR & VAR_SCHED) && !(VAR & ABL_NO_QK) && !(VAR & ABL_NO_LDS) && !(VAR & ABL_NO_MFMA) && !(VAR & ABL_AGPR_ACC) &&
            !(VAR & ABL_AGPR_OPND)) {
            __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
            for (int i = 0; i < NQK - PF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);
        }

        if constexpr (VAR & ABL_AGPR_ACC) {   // timing only: consume the accumulators where they are and stop
            asm volatile("s_nop 15\n\ts_nop 15" ::"a"(s[0]), "a"(s[1]));
            return;
        }
        if constexpr (VAR & VAR_LATEDMA) {
            if (jnext >= 0) dma_tile(IC<BUF ^ 1>{}, jnext);   // late issue (younger half): off the post-barrier rush
        }
        if constexpr (VAR & VAR_STAMP) {   // QK^T segment: from compute start to the last QK MFMA issued
            const unsigned long long tq1 = stamp();
            st_qk_end = tq1;
        }
        // mask: wave-uniform test, only diagonal / tail / key-mask tiles pay
        const bool bits_mode = KMASK && mrow != nullptr;                         // mask as one word per row and tile (tile_bits)
        const bool need_mask = (key_base + BLOCK_N > kv_len) || (CAUSAL && key_base + BLOCK_N - 1 > wave_q0) ||
                               (KMASK && (!bits_mode || !tile_all_ones));
        if (need_mask) {
            asm volatile("" ::: "memory");   // keep this a real (wave-uniform) branch: hipcc otherwise if-converts
                                             // it into 32 v_cmp + 32 v_cndmask on EVERY tile
            // the lane's 32 mask bits of each key block: keys 32 kb + 4 h + {0..3, 8..11, 16..19, 24..27}
            const uint32_t mw[2] = {(uint32_t)(tile_bits >> (4 * h)), (uint32_t)(tile_bits >> (32 + 4 * h))};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = key_base + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * h;
                    bool ok = key < kv_len;
                    if (CAUSAL) ok = ok && (key <= my_q);
                    if (KMASK) {
                        if (bits_mode) ok = ok && (((mw[kb] >> ((e & 3) + 8 * (e >> 2))) & 1u) != 0);
                        else ok = ok && (kmp[(int64_t)min(key, p.Sk - 1) * p.m_sk] != 0);
                    }
                    s[kb][e] = ok ? s[kb][e] : -INFINITY;
                }
        }

        if constexpr (!(VAR & ABL_NO_SOFTMAX)) {
        // online softmax (flash_attention_3.py:239-246); a row lives in lanes (l, l^32)
        float mx;
        if constexpr (VAR & VAR_DIET) {
            mx = max16_first(s[0]);
            mx = max16_next(mx, s[1]);
        } else {
            mx = max3(s[0][0], s[1][0], s[0][1]);
            mx = max3(mx, s[1][1], s[0][2]);
#pragma unroll
            for (int e = 2; e < 16; e += 2) {
                mx = max3(mx, s[1][e], s[0][e + 1]);
                if (e + 2 < 16) mx = max3(mx, s[1][e + 1], s[0][e + 2]);
                else mx = fmaxf(mx, s[1][e + 1]);
            }
        }
        mx = (VAR & VAR_DIET) ? row_pair_max_asm(mx) : row_pair_max(mx);
        // rescale only when some row's max outgrew the headroom (thr = 0: whenever any max moved -> exact
        // lazy rescale; thr > 0: exponentials may reach 2^8, harmless in fp32/bf16 and cancelled by l)
        if (__builtin_amdgcn_ballot_w64(mx > m_thr) != 0) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = fast_exp2((m_run - m_new) * c);
            m_run = m_new;
            m_thr = m_new + thr;
            mc = m_new * c;
            l_run *= alpha;
            if (VAR & VAR_LSUM) {
#pragma unroll
                for (int e = 0; e < 16; ++e) lacc[e] *= alpha;
            }
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
        }
        float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if constexpr (VAR & ABL_NO_EXP) {
                s[0][e] = __builtin_fmaf(s[0][e], c, -mc);
                s[1][e] = __builtin_fmaf(s[1][e], c, -mc);
            } else {
                s[0][e] = fast_exp2(__builtin_fmaf(s[0][e], c, -mc));
                s[1][e] = fast_exp2(__builtin_fmaf(s[1][e], c, -mc));
            }
            if (!(VAR & VAR_LSUM)) {
                psum0 += s[0][e];
                if constexpr (VAR & VAR_DIET) asm volatile("" : "+v"(psum0));   // keeps SLP from pairing the sums into v_pk_add_f32
                psum1 += s[1][e];
            }
        }
        if (!(VAR & VAR_LSUM)) l_run += psum0 + psum1;
        }   // !ABL_NO_SOFTMAX

        // O^T += V^T P^T
        if (VAR & VAR_SETPRIO) __builtin_amdgcn_s_setprio(1);
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
                if (VAR & VAR_LSUM) {   // l += sum over this k-step's 16 keys of the ROUNDED p (matches the PV numerator)
                    lacc = E::mfma(ones, ph, lacc);
                    if (SPLITP) lacc = E::mfma(ones, pl, lacc);
                }
                constexpr int S2I = (D == 128) ? 0 : 1;
                const int koffs = kb * HALF_TILE + ((D == 128) ? s2 * 16 * 256 : 0);
                if constexpr (VAR & ABL_NO_PV) {
                    asm volatile("" ::"v"(ph));
                    continue;
                }
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
                    o[db] = MF(a, ph, o[db]);
                    if (SPLITP) o[db] = E::mfma(a, pl, o[db]);
                }
            }
        if (VAR & VAR_SETPRIO) __builtin_amdgcn_s_setprio(0);
    };

    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // VAR_STAMP: dma issue | compute | vmcnt wait | barrier | tiles
    unsigned long long st_t0 = 0, st_r0 = 0;
    if constexpr (VAR & VAR_STAMP) {   // shader clock vs the constant 100 MHz counter -> in-kernel clock (guide, DVFS item 6)
        st_t0 = __builtin_amdgcn_s_memtime();
        st_r0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }

    auto step = [&](auto bufc, int j) {
        constexpr int BUF = decltype(bufc)::value;
        if constexpr ((VAR & VAR_GLDS) && (VAR & VAR_STAMP)) {
            const unsigned long long t0 = stamp();
            if (j + 1 < nt) dma_tile(IC<BUF ^ 1>{}, j + 1);
            const unsigned long long t1 = stamp();
            if (tile_live(j)) compute_tile(bufc, j * BLOCK_N);
            const unsigned long long t2 = stamp();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t3 = stamp();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            const unsigned long long t4 = stamp();
            st_acc[5] += st_qk_end - t1;   // QK^T segment of the compute
            st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += t3 - t2; st_acc[3] += t4 - t3; st_acc[4] += 1;
        } else if constexpr ((VAR & VAR_GLDS) && (VAR & VAR_LATEDMA)) {
            const bool late = (wave >= NW / 2) && (j * BLOCK_N < wave_kv_end);
            if (j + 1 < nt && !late) dma_tile(IC<BUF ^ 1>{}, j + 1);
            if (tile_live(j)) compute_tile(bufc, j * BLOCK_N, (late && j + 1 < nt) ? j + 1 : -1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
        } else if constexpr (VAR & VAR_GLDS) {
            const bool live = tile_live(j);
            if constexpr (!(VAR & ABL_NO_DMA)) {
                if (j + 1 < nt) dma_tile(IC<BUF ^ 1>{}, j + 1);   // lands in the other buffer under this tile's math
            }
            if (live) compute_tile(bufc, j * BLOCK_N);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed ...
            __builtin_amdgcn_s_waitcnt(0xC07F);               // (lgkmcnt(0): this wave's LDS reads are done)
            if constexpr (!(VAR & ABL_NO_BARRIER)) __builtin_amdgcn_s_barrier();   // ... and so have everybody else's
        } else {
            if (j + 1 < nt) load_tile(j + 1);              // HBM latency hides under this tile's math
            if (tile_live(j)) compute_tile(bufc, j * BLOCK_N);   // wave-uniform causal skip
            if (j + 1 < nt) store_tile(IC<BUF ^ 1>{});
            __syncthreads();
        }
    };

    fetch_bits(j0);
    if (nt > j0) {
        if constexpr (VAR & VAR_GLDS) {
            dma_tile(IC<0>{}, j0);
            if constexpr ((VAR & VAR_STAGE2) || (VAR & VAR_RING3)) {
                if (nt > 1) dma_tile(IC<1>{}, 1);
            }
        } else {
            load_tile(j0);
            store_tile(IC<0>{});
        }
    }
    if constexpr (VAR & VAR_YPRIO) {
        if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);   // wave is an SGPR value: a real scalar branch
    }
    // Q must have LANDED before the loop: hipcc's waitcnt pass otherwise keeps vmcnt(7..0) waits for the Q
    // loads inside the loop body, where they drain the K/V prefetch of the next tile on every iteration.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (VAR & VAR_RING3) {
        // Waves reach the end of a tile at different times (the older half of each SIMD pair ~700 cycles early):
        // issuing there spreads the 32 DMA pieces of a tile over time instead of queueing all of them on the CU's
        // address path right after the barrier, and the early waves pay the issue latency out of their barrier wait.
        static_assert((VAR & VAR_GLDS) != 0, "VAR_RING3 needs the LDS-DMA path");
        auto ring_step = [&](auto slotc, int j) {
            constexpr int SLOT = decltype(slotc)::value;
            if (tile_live(j)) compute_tile(slotc, j * BLOCK_N);
            if (j + 2 < nt) {
                dma_tile(IC<(SLOT + 2) % 3>{}, j + 2);
                // tile j+1 (issued one tile ago) must have landed; tile j+2's 2*PPW pieces may stay in flight
                if constexpr (2 * PPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if constexpr (2 * PPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
        };
        for (int j = 0; j < nt; j += 3) {
            ring_step(IC<0>{}, j);
            if (j + 1 < nt) ring_step(IC<1>{}, j + 1);
            if (j + 2 < nt) ring_step(IC<2>{}, j + 2);
        }
    } else if constexpr (VAR & VAR_STAGE2) {
        // stage = two tiles in LDS slots {2P, 2P+1}; one barrier per stage
        static_assert((VAR & VAR_GLDS) != 0, "VAR_STAGE2 needs the LDS-DMA path");
        auto stage = [&](auto pc, int js) {
            constexpr int P = decltype(pc)::value;
            const int j0 = 2 * js;
            if (j0 + 2 < nt) dma_tile(IC<2 * (P ^ 1)>{}, j0 + 2);
            if (j0 + 3 < nt) dma_tile(IC<2 * (P ^ 1) + 1>{}, j0 + 3);
            if (j0 * BLOCK_N < wave_kv_end) compute_tile(IC<2 * P>{}, j0 * BLOCK_N);
            if (j0 + 1 < nt && (j0 + 1) * BLOCK_N < wave_kv_end) compute_tile(IC<2 * P + 1>{}, (j0 + 1) * BLOCK_N);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
        };
        const int ns = (nt + 1) / 2;
        for (int js = 0; js < ns; js += 2) {
            stage(IC<0>{}, js);
            if (js + 1 < ns) stage(IC<1>{}, js + 1);
        }
    } else {
        for (int j = j0; j < nt; j += 2) {
            step(IC<0>{}, j);
            if (j + 1 < nt) step(IC<1>{}, j + 1);
        }
    }

    if constexpr (VAR & VAR_STAMP) {
        st_acc[6] = __builtin_amdgcn_s_memtime() - st_t0;
        st_acc[7] = __builtin_amdgcn_s_memrealtime() - st_r0;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (lane == 0 && p.dbg) {
            unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = st_acc[i];
        }
    }
    // ---- epilogue: normalise (flash_attention_3.py:250 does it per tile; once is equivalent) ------------
    const float l_tot = (VAR & VAR_LSUM) ? lacc[0] : row_pair_sum(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;   // fully masked row -> zeros (documented divergence)
    if constexpr (sizeof(OT) == 2 && (VAR & VAR_DIET)) {
        // 16-bit store: lanes l and l+32 hold the two 8-byte halves of every 16-byte column group of one row.  One
        // v_permlane32_swap per dword pairs group k (even) with k+1 so that each lane owns 16 contiguous bytes:
        // 8 dwordx4 stores per lane instead of 16 dwordx2 -- the tail is store-issue bound (cdna guide T21).
        // The 16-byte chunks then go through LDS (free after the loop's last barrier; 32 rows x D*2 bytes per wave, chunk
        // index XOR-swizzled by the row) and come back so that one store instruction covers WHOLE rows: per-lane stores
        // at the row stride touch 64 cache lines per instruction, these 8 (D = 128) or 16 (D = 64).
        typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;
        constexpr int RB = D * 2, CPRW = RB / 16;      // row bytes, 16-byte chunks per row
        const uint32_t lbase = smem_base + wave * (32 * RB);
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                v4 wa, wb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    wa[e] = (T)(o[db][4 * g + e] * inv);
                    wb[e] = (T)(o[db][4 * g + 4 + e] * inv);
                }
                u32x2 a = __builtin_bit_cast(u32x2, wa), bq = __builtin_bit_cast(u32x2, wb);
                auto r0 = __builtin_amdgcn_permlane32_swap(a[0], bq[0], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(a[1], bq[1], false, false);
                u32x4 w = {r0[0], r1[0], r0[1], r1[1]};
                const uint32_t ch = 4 * db + g + h;
                *(lds_u32x4_t*)(uintptr_t)(lbase + r * RB + ((ch ^ (r & (CPRW - 1))) << 4)) = w;
            }
        store_rows_from_lds<RB>(lbase, lane, (char*)((OT*)p.o + (int64_t)b * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)wave_q0 * p.o_ss),
                                p.o_ss * 2, p.Sq - wave_q0);
    } else if (my_q < p.Sq) {
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
    }
    if (my_q < p.Sq) {
        if (p.lse && h == 0) {
            const float lse = l_tot > 0.f ? (m_run * c + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f : -INFINITY;
            p.lse[((int64_t)b * p.H + hh) * p.Sq + my_q] = lse;
        }
    }
}

}  // namespace pfa

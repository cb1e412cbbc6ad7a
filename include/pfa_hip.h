/*
 * pfa_hip.h -- C ABI of libpfa_hip.so: the MI355X (gfx950) Flash-Attention forward (and backward) that
 * replaces the body of the reference's electronic attention core.
 * ABI history: v1 forward; v2 general masks + weights; v3 backward; v4 grouped-query heads (kv_group); v5 fp32 operands (exact
 * fp32 kernels, forward and backward) and dense-branch attention dropout; v6 pfa_fa3_prepare, reserve_cus, pfa_probe_mfma;
 * v7 pfa_fa3_bwd_args.kv_group (grouped-query heads in the backward: dK / dV summed over the group in the kernel).
 *
 * Reference seam (danieleschmidt/Photonic-Flash-Attention, all paths under
 * src/photonic_flash_attention/):
 *
 *   pfa_fa3_fwd             replaces  core/flash_attention_3.py:120-150  _flash_attention_forward
 *                                     = :152-180 _standard_attention + :182-262 _tiled_attention
 *                                     (called from FlashAttention3.forward at :102)
 *   pfa_fa3_args.softmax_scale         the `q = q * self.scaling` pass at :138, folded into the kernel
 *   pfa_fa3_args.causal / seqlens_k /  the `attention_mask` argument (:165-168, :234-236): causal = 4-D
 *     key_mask / mask                  lower-triangular mask, seqlens_k / key_mask = 2-D [B,Sk] key mask,
 *                                      mask = any 4-D mask
 *   pfa_fa3_weights                    the `need_weights=True` outputs (:171,:180,:257-258)
 *   pfa_fa3_args.o strides             the `.transpose(1,2).contiguous()` copy at :107 (the kernel writes
 *                                     [B,S,H,D] directly, so the copy disappears)
 *   pfa_fa3_bwd                        autograd through :152-262 -- the reference's only backward (its modules train
 *                                     through the eager core; tests/unit/test_flash_attention_3.py:137-160)
 *   pfa_fa3_workspace_bytes           the tile-size memory budget of :264-293 (this path needs none; a key mask can use a few bytes)
 *   pfa_device_supported              the `torch.cuda.is_available()` probes at :71,:142
 *
 * The reference has no native code, so nothing binds an FFI today; INTEGRATION.md
 * shows the ctypes stub a maintainer adds at flash_attention_3.py:102.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer is a DEVICE pointer owned by the caller;
 *   - the library allocates nothing, frees nothing, keeps no reference after return;
 *   - asynchronous: work is enqueued on `stream` (a hipStream_t passed as void*), no implicit sync;
 *   - re-entrant and thread-safe; the only process-wide state is one code-object handle per device (the assembly kernels'
 *     hipModule_t), loaded once under a mutex by pfa_fa3_prepare / pfa_device_supported (or, failing that, by the first call)
 *     and never changed afterwards;
 *   - returns PFA_OK (0) or a negative pfa_status; never aborts the process.
 */
#ifndef PFA_HIP_H
#define PFA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFA_ABI_VERSION 7

typedef enum pfa_status {
    PFA_OK = 0,
    PFA_ERR_NULL = -1,          /* required pointer is NULL                         */
    PFA_ERR_STRUCT_SIZE = -2,   /* args->size does not match a known struct version */
    PFA_ERR_SHAPE = -3,         /* B,H,Sq,Sk <= 0 or grid limits exceeded           */
    PFA_ERR_HEAD_DIM = -4,      /* D not in {64, 128}                               */
    PFA_ERR_DTYPE = -5,         /* dtype_in / dtype_out unsupported                 */
    PFA_ERR_STRIDE = -6,        /* a stride is not a multiple of 8 elements         */
    PFA_ERR_ALIGN = -7,         /* a base pointer is not 16-byte aligned            */
    PFA_ERR_DEVICE = -8,        /* device is not gfx950 / cannot be selected        */
    PFA_ERR_LAUNCH = -9,        /* hipLaunchKernel failed (see pfa_last_hip_error)  */
    PFA_ERR_FLAGS = -10         /* unknown flag bits                                */
} pfa_status;

typedef enum pfa_dtype {
    PFA_DTYPE_BF16 = 0,
    PFA_DTYPE_FP16 = 1,
    PFA_DTYPE_FP32 = 2          /* output of the 16-bit kernels; as dtype_in: the EXACT fp32 kernel (fp32 modules, slow) */
} pfa_dtype;

/* flags */
#define PFA_FLAG_SPLIT_P   0x1u  /* carry P as bf16 hi+lo (two PV MFMA passes): the <=1e-3 parity mode        */
#define PFA_FLAG_NO_XCD_MAP 0x2u /* debugging: identity block->work mapping                                  */
#define PFA_FLAG_VARIANT_MASK 0xff00u /* bits 8..15: kernel selector for tests / A-B runs.  0 = the library chooses; 43 = the 4-wave HIP kernel,
                                         44 = the 8-wave HIP kernel, 45 = the persistent 4-wave assembly kernel (each only where it applies;
                                         otherwise the library's choice).  Anything else is PFA_ERR_FLAGS. */

/*
 * One attention problem: O[b,i,h,:] = softmax_j(scale * <Q[b,i,h,:], K[b,j,h,:]> + mask) V[b,j,h,:]
 *
 * Tensors are addressed as base + b*stride_b + h*stride_h + s*stride_s + d (strides in ELEMENTS of the
 * tensor's dtype, last dimension contiguous).  That covers [B,S,H,D] (what a fused QKV projection yields:
 * q = qkv[..., 0:E] has stride_s = 3E, stride_h = D), [B,H,S,D], and slices of either.
 */
typedef struct pfa_fa3_args {
    uint32_t size;              /* = sizeof(pfa_fa3_args); versions the struct       */
    uint32_t flags;             /* PFA_FLAG_*                                        */

    const void* q;              /* [B, Sq, H, D] by strides, dtype_in                */
    const void* k;              /* [B, Sk, H, D]                                     */
    const void* v;              /* [B, Sk, H, D]                                     */
    void*       o;              /* [B, Sq, H, D], dtype_out                          */
    float*      lse;            /* optional [B, H, Sq] fp32 natural-log LSE, or NULL */
    const int32_t* seqlens_k;   /* optional [B]: keys >= seqlens_k[b] are masked     */
    const uint8_t* key_mask;    /* optional [B, Sk] bytes, 0 = masked (2-D mask)     */

    int64_t q_stride_b, q_stride_h, q_stride_s;
    int64_t k_stride_b, k_stride_h, k_stride_s;
    int64_t v_stride_b, v_stride_h, v_stride_s;
    int64_t o_stride_b, o_stride_h, o_stride_s;
    int64_t key_mask_stride_b;  /* bytes between batches of key_mask                 */

    int32_t B, H, Sq, Sk, D;
    int32_t dtype_in;           /* PFA_DTYPE_BF16 | PFA_DTYPE_FP16 | PFA_DTYPE_FP32 (exact fp32 kernel: strides multiples of 4 elements, dtype_out fp32, no flags) */
    int32_t dtype_out;          /* = dtype_in, or PFA_DTYPE_FP32                     */
    int32_t causal;             /* 1: key j visible to row i iff j <= i (top-left)   */
    float   softmax_scale;      /* usually D^-0.5                                    */
    int32_t device_id;          /* HIP device ordinal the pointers live on           */

    void*   workspace;          /* pfa_fa3_workspace_bytes() bytes, may be NULL if 0 */
    size_t  workspace_bytes;

    /* ABI v2: general mask, the reference's 4-D `attention_mask` (flash_attention_3.py:168,235-236).
     * u8, 0 = masked, element (b,h,i,j) at mask + b*mask_stride_b + h*mask_stride_h + i*mask_stride_q +
     * j*mask_stride_k (BYTE strides; 0 broadcasts a dimension).  At most one of key_mask / mask may be set;
     * `causal` and `seqlens_k` combine with either. */
    const uint8_t* mask;
    int64_t mask_stride_b, mask_stride_h, mask_stride_q, mask_stride_k;

    /* ABI v4: grouped-query attention (not in the reference; what Hugging Face decoder models hand over).  k and v hold
     * H / kv_group heads and query head h reads K/V head h / kv_group, so nothing has to be expanded in memory.
     * 0 or 1 = one K/V head per query head.  H must be a multiple of kv_group.  (The backward takes the same field since ABI v7:
     * pfa_fa3_bwd_args.kv_group.) */
    int32_t kv_group;
    /* ABI v6: CUs to leave free (0 = none).  The persistent forward holds one workgroup on every CU for the whole launch, so a
     * kernel-based collective (RCCL) enqueued beside it only runs when it drains; a caller that overlaps such a collective passes
     * the CUs it needs (the grid shrinks to n_cu - reserve_cus, a multiple of 8).  Copy-engine transfers need none. */
    int32_t reserve_cus;

    /* ABI v5: attention dropout of the reference's dense branch (flash_attention_3.py:174-175: dropout(softmax(scores)) @ v).
     * drop_mask: keep-mask bytes [B][H][Sq][Sk] contiguous (non-zero = keep), drawn by the caller; kept weights are scaled by
     * drop_scale = 1 / (1 - p), the softmax normaliser is the un-dropped row sum.  Only with dtype_in = fp32 (the fp32 kernels);
     * NULL = no dropout. */
    const uint8_t* drop_mask;
    float   drop_scale;
    int32_t reserved1;          /* must be 0 */
} pfa_fa3_args;

/* ABI version of the loaded library (== PFA_ABI_VERSION of the header it was built from). */
int pfa_abi_version(void);

/* Human-readable text for a pfa_status. */
const char* pfa_status_string(int status);

/* 1 if HIP device `device_id` is a gfx950 part this library has code for, 0 if not, <0 on error. */
int pfa_device_supported(int device_id);

/* ABI v6: load the device's code objects NOW (idempotent, thread-safe): afterwards no call on that device loads a module, so a
 * first pfa_fa3_fwd may sit inside a hipStreamBeginCapture region or a timed loop.  pfa_device_supported does the same.
 * PFA_OK, or PFA_ERR_DEVICE (pfa_last_hip_error: why). */
int pfa_fa3_prepare(int device_id);

/* Last hipError_t seen by a failing call on this thread (0 = hipSuccess). */
int pfa_last_hip_error(void);

/* Scratch bytes pfa_fa3_fwd can use for `a`: 0 without a mask; with key_mask or mask, 8 bytes per un-broadcast mask row and
 * 64-key tile (+ 128 bytes per 256 mask rows) -- pfa_fa3_fwd first condenses the mask into one 64-bit word per row and tile
 * there (one word read per tile instead of a mask byte per score: 2-4 x faster) and notes, per 256 rows, the first and last
 * tile that holds a visible key: a Q block then runs only those tiles, so a structured mask (a band, a triangle, padding)
 * skips what it hides.  Optional: with workspace == NULL or too few bytes the mask is read byte-wise, every tile runs, and
 * the result is the same. */
size_t pfa_fa3_workspace_bytes(const pfa_fa3_args* a);

/* Validate `a` without launching: PFA_OK or the error pfa_fa3_fwd would return. */
int pfa_fa3_check(const pfa_fa3_args* a);

/* Enqueue the forward on `stream` (hipStream_t as void*, NULL = default stream). */
int pfa_fa3_fwd(const pfa_fa3_args* a, void* stream);

/*
 * Attention weights on request (the reference's need_weights=True, flash_attention_3.py:171,180,257-258):
 * W[b,h,i,j] = exp(scale*<q_i,k_j> + mask - lse[b,h,i]), the TRUE softmax row (the reference's tiled branch
 * returns un-renormalised tiles; documented divergence).  `a` is the argument block of the forward call that
 * produced a->lse (required; a->v and a->o are ignored).  W is addressed base + b*w_stride_b + h*w_stride_h +
 * i*w_stride_q + j (ELEMENT strides, last dim contiguous), dtype w_dtype = a->dtype_in or PFA_DTYPE_FP32.
 * Every element of W is written exactly once (masked ones as zeros): W may be uninitialised.
 */
int pfa_fa3_weights(const pfa_fa3_args* a, void* w, int32_t w_dtype, int64_t w_stride_b, int64_t w_stride_h,
                    int64_t w_stride_q, void* stream);

/*
 * Backward pass (ABI v3).  The reference obtains gradients from autograd through its eager forward
 * (flash_attention_3.py:152-262; its unit tests only require that gradients exist, tests/unit/
 * test_flash_attention_3.py:137-160); this entry point computes dQ, dK, dV from q, k, v, o, dO and the forward's
 * LSE by recomputation (two launches: dQ per query block, which also produces delta = rowsum(dO*O), then dK/dV per key block; no
 * atomics, bitwise reproducible).  Masks: `causal`, `seqlens_k` and the forward's general u8 `mask` (field below; a [B,Sk] key mask
 * is the broadcast form mask_stride_b = Sk, _h = 0, _q = 0, _k = 1).
 * All tensors [B,S,H,D] by element strides, last dim contiguous; gradients in `dtype_grad` (= dtype or fp32).
 * `delta` is caller-provided scratch of pfa_fa3_bwd_workspace_bytes() bytes ([B,H,Sq] fp32).
 */
typedef struct pfa_fa3_bwd_args {
    uint32_t size;              /* = sizeof(pfa_fa3_bwd_args) */
    uint32_t flags;             /* must be 0 */
    const void* q;
    const void* k;
    const void* v;
    const void* o;              /* forward output */
    const void* dout;           /* gradient of the forward output */
    const float* lse;           /* [B,H,Sq] from the forward (pfa_fa3_args.lse) */
    void* dq;
    void* dk;
    void* dv;
    float* delta;               /* scratch [B,H,Sq] */
    const int32_t* seqlens_k;   /* optional [B] */
    int64_t q_stride_b, q_stride_h, q_stride_s;
    int64_t k_stride_b, k_stride_h, k_stride_s;
    int64_t v_stride_b, v_stride_h, v_stride_s;
    int64_t o_stride_b, o_stride_h, o_stride_s;
    int64_t do_stride_b, do_stride_h, do_stride_s;
    int64_t dq_stride_b, dq_stride_h, dq_stride_s;
    int64_t dk_stride_b, dk_stride_h, dk_stride_s;
    int64_t dv_stride_b, dv_stride_h, dv_stride_s;
    int32_t B, H, Sq, Sk, D;
    int32_t dtype;              /* PFA_DTYPE_BF16 | PFA_DTYPE_FP16 | PFA_DTYPE_FP32 (v5): q,k,v,o,dout */
    int32_t dtype_grad;         /* = dtype or PFA_DTYPE_FP32: dq,dk,dv */
    int32_t causal;
    float   softmax_scale;
    int32_t device_id;
    /* optional element mask of the forward, same convention as pfa_fa3_args.mask (u8, 0 = masked, BYTE strides, 0
     * broadcasts a dimension; a [B,Sk] key mask is mask_stride_b = Sk, _h = 0, _q = 0, _k = 1).  Masked scores get no
     * gradient; rows the forward found fully masked (lse = -inf) get dq = 0 and contribute nothing to dk, dv. */
    const uint8_t* mask;
    int64_t mask_stride_b, mask_stride_h, mask_stride_q, mask_stride_k;
    /* ABI v5: dtype = PFA_DTYPE_FP32 (all tensors fp32, strides multiples of 4 elements, `delta` unused) runs the fp32 backward
     * kernels; only they take the forward's dropout keep-mask (see pfa_fa3_args.drop_mask). */
    const uint8_t* drop_mask;
    float   drop_scale;
    int32_t kv_group;           /* ABI v7 (was reserved1, must-be-0): grouped-query heads as in pfa_fa3_args.kv_group -- k, v, dk, dv hold
                                 * H / kv_group heads; dK / dV are summed over each group inside the kernel.  0 or 1 = none.  16-bit operands only. */
    /* ABI v7: optional scratch for an ELEMENT mask (16-bit operands; ignored for key-only masks): pfa_fa3_bwd_mask_workspace_bytes()
     * bytes.  pfa_fa3_bwd condenses the mask there into a word per row and 64-key tile, the same transposed (a word per key and 64-row
     * tile) and the tile ranges that hold visible entries: the kernels then read one word per tile instead of a mask byte per score and
     * run only the tiles a structured mask (a band, a triangle, documents) leaves visible.  NULL / too small: the byte paths, same result. */
    void*   mask_workspace;
    size_t  mask_workspace_bytes;
} pfa_fa3_bwd_args;

size_t pfa_fa3_bwd_workspace_bytes(const pfa_fa3_bwd_args* a);
size_t pfa_fa3_bwd_mask_workspace_bytes(const pfa_fa3_bwd_args* a);
int pfa_fa3_bwd(const pfa_fa3_bwd_args* a, void* stream);

/*
 * Kernel-selection introspection for tests/bench: writes the name of the kernel variant pfa_fa3_fwd
 * would launch for `a` into buf (NUL terminated, truncated to n) and returns the number of workgroups.
 */
int pfa_fa3_describe(const pfa_fa3_args* a, char* buf, size_t n);

/*
 * Measurement aid (bench.py, roofline.probe_tflops; not on the hot path): enqueue a bare v_mfma_f32_32x32x16_bf16 stream -- one wave
 * per SIMD on every CU, A operands re-read from LDS as the forward's tile loop reads its K / V^T fragments, random operands taken from
 * `random_64k` (64 KiB of device memory holding bf16 values) -- of `iters` x 64 MFMAs per wave.  `sink`: device scratch of
 * 4 B x 256 x the number of CUs.  Returns the number of workgroups launched (> 0) or a pfa_status; *flops = the launch's flops.
 */
int pfa_probe_mfma(const void* random_64k, float* sink, int iters, int device_id, void* stream, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* PFA_HIP_H */

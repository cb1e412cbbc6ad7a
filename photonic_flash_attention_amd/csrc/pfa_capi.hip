// pfa_capi.hip -- the C ABI of libpfa_hip.so (see include/pfa_hip.h for the contract and the
// reference lines each entry point replaces).  Host side only validates, picks a kernel variant
// and enqueues it; no allocation, no synchronisation, no global mutable state.
#include "pfa_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "fa3_fwd_kernel.h"

namespace {

thread_local int g_last_hip_error = 0;

struct Variant {
    const void* fn;
    const char* name;
    int lds_bytes;
};

template <typename T, int D, bool C, bool S, typename OT>
Variant make_variant(const char* name) {
    return Variant{(const void*)&pfa::fa3_fwd_kernel<T, D, C, S, OT>, name, 2 * 2 * pfa::BLOCK_N * D * 2};
}

#define PFA_V(T, TN, D, C, S, OT, ON) make_variant<T, D, C, S, OT>("fa3_fwd_" TN "_d" #D "_" #C "_" #S "_" ON)

// index: [dtype_in][D==128][causal][split][out_fp32]
const Variant& pick(int dtype_in, int D, int causal, int split, int out32) {
    static const Variant tbl[2][2][2][2][2] = {
        {{{{PFA_V(__bf16, "bf16", 64, false, false, __bf16, "o16"), PFA_V(__bf16, "bf16", 64, false, false, float, "o32")},
           {PFA_V(__bf16, "bf16", 64, false, true, __bf16, "o16"), PFA_V(__bf16, "bf16", 64, false, true, float, "o32")}},
          {{PFA_V(__bf16, "bf16", 64, true, false, __bf16, "o16"), PFA_V(__bf16, "bf16", 64, true, false, float, "o32")},
           {PFA_V(__bf16, "bf16", 64, true, true, __bf16, "o16"), PFA_V(__bf16, "bf16", 64, true, true, float, "o32")}}},
         {{{PFA_V(__bf16, "bf16", 128, false, false, __bf16, "o16"), PFA_V(__bf16, "bf16", 128, false, false, float, "o32")},
           {PFA_V(__bf16, "bf16", 128, false, true, __bf16, "o16"), PFA_V(__bf16, "bf16", 128, false, true, float, "o32")}},
          {{PFA_V(__bf16, "bf16", 128, true, false, __bf16, "o16"), PFA_V(__bf16, "bf16", 128, true, false, float, "o32")},
           {PFA_V(__bf16, "bf16", 128, true, true, __bf16, "o16"), PFA_V(__bf16, "bf16", 128, true, true, float, "o32")}}}},
        {{{{PFA_V(_Float16, "fp16", 64, false, false, _Float16, "o16"), PFA_V(_Float16, "fp16", 64, false, false, float, "o32")},
           {PFA_V(_Float16, "fp16", 64, false, true, _Float16, "o16"), PFA_V(_Float16, "fp16", 64, false, true, float, "o32")}},
          {{PFA_V(_Float16, "fp16", 64, true, false, _Float16, "o16"), PFA_V(_Float16, "fp16", 64, true, false, float, "o32")},
           {PFA_V(_Float16, "fp16", 64, true, true, _Float16, "o16"), PFA_V(_Float16, "fp16", 64, true, true, float, "o32")}}},
         {{{PFA_V(_Float16, "fp16", 128, false, false, _Float16, "o16"), PFA_V(_Float16, "fp16", 128, false, false, float, "o32")},
           {PFA_V(_Float16, "fp16", 128, false, true, _Float16, "o16"), PFA_V(_Float16, "fp16", 128, false, true, float, "o32")}},
          {{PFA_V(_Float16, "fp16", 128, true, false, _Float16, "o16"), PFA_V(_Float16, "fp16", 128, true, false, float, "o32")},
           {PFA_V(_Float16, "fp16", 128, true, true, _Float16, "o16"), PFA_V(_Float16, "fp16", 128, true, true, float, "o32")}}}}};
    return tbl[dtype_in][D == 128][causal ? 1 : 0][split ? 1 : 0][out32 ? 1 : 0];
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool mult8(int64_t s) { return (s % 8) == 0; }

int check(const pfa_fa3_args* a) {
    if (!a) return PFA_ERR_NULL;
    if (a->size != sizeof(pfa_fa3_args)) return PFA_ERR_STRUCT_SIZE;
    if (a->flags & ~(PFA_FLAG_SPLIT_P | PFA_FLAG_NO_XCD_MAP)) return PFA_ERR_FLAGS;
    if (!a->q || !a->k || !a->v || !a->o) return PFA_ERR_NULL;
    if (a->B <= 0 || a->H <= 0 || a->Sq <= 0 || a->Sk <= 0) return PFA_ERR_SHAPE;
    if (a->D != 64 && a->D != 128) return PFA_ERR_HEAD_DIM;
    if (a->dtype_in != PFA_DTYPE_BF16 && a->dtype_in != PFA_DTYPE_FP16) return PFA_ERR_DTYPE;
    if (a->dtype_out != a->dtype_in && a->dtype_out != PFA_DTYPE_FP32) return PFA_ERR_DTYPE;
    if (!(a->softmax_scale > 0.f) || !isfinite(a->softmax_scale)) return PFA_ERR_SHAPE;
    const int64_t st[] = {a->q_stride_b, a->q_stride_h, a->q_stride_s, a->k_stride_b, a->k_stride_h, a->k_stride_s,
                          a->v_stride_b, a->v_stride_h, a->v_stride_s};
    for (int64_t s : st)
        if (!mult8(s)) return PFA_ERR_STRIDE;
    const int64_t ost[] = {a->o_stride_b, a->o_stride_h, a->o_stride_s};
    for (int64_t s : ost)
        if (s % 4 != 0) return PFA_ERR_STRIDE;
    if (!aligned16(a->q) || !aligned16(a->k) || !aligned16(a->v) || !aligned16(a->o)) return PFA_ERR_ALIGN;
    if (a->lse && (reinterpret_cast<uintptr_t>(a->lse) & 3u)) return PFA_ERR_ALIGN;
    const int64_t nq = (a->Sq + pfa::BLOCK_M - 1) / pfa::BLOCK_M;
    if (nq * a->B * a->H > 0x7fffffffLL) return PFA_ERR_SHAPE;
    return PFA_OK;
}

}  // namespace

extern "C" {

int pfa_abi_version(void) { return PFA_ABI_VERSION; }

const char* pfa_status_string(int status) {
    switch (status) {
        case PFA_OK: return "ok";
        case PFA_ERR_NULL: return "required pointer is NULL";
        case PFA_ERR_STRUCT_SIZE: return "pfa_fa3_args.size does not match this library's struct";
        case PFA_ERR_SHAPE: return "invalid shape or softmax_scale (B,H,Sq,Sk must be > 0, scale finite > 0)";
        case PFA_ERR_HEAD_DIM: return "head dim must be 64 or 128";
        case PFA_ERR_DTYPE: return "dtype_in must be bf16/fp16 and dtype_out the same or fp32";
        case PFA_ERR_STRIDE: return "q/k/v strides must be multiples of 8 elements (o: 4)";
        case PFA_ERR_ALIGN: return "q/k/v/o base pointers must be 16-byte aligned";
        case PFA_ERR_DEVICE: return "device is not a supported gfx950 part";
        case PFA_ERR_LAUNCH: return "HIP kernel launch failed";
        case PFA_ERR_FLAGS: return "unknown flag bits";
        default: return "unknown pfa_status";
    }
}

int pfa_last_hip_error(void) { return g_last_hip_error; }

int pfa_device_supported(int device_id) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        (void)hipGetLastError();
        return PFA_ERR_DEVICE;
    }
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

size_t pfa_fa3_workspace_bytes(const pfa_fa3_args* a) {
    (void)a;
    return 0;
}

int pfa_fa3_check(const pfa_fa3_args* a) { return check(a); }

int pfa_fa3_describe(const pfa_fa3_args* a, char* buf, size_t n) {
    const int st = check(a);
    if (st != PFA_OK) return st;
    const Variant& v = pick(a->dtype_in, a->D, a->causal, a->flags & PFA_FLAG_SPLIT_P, a->dtype_out == PFA_DTYPE_FP32);
    if (buf && n) {
        strncpy(buf, v.name, n - 1);
        buf[n - 1] = 0;
    }
    const int nq = (a->Sq + pfa::BLOCK_M - 1) / pfa::BLOCK_M;
    return nq * a->B * a->H;
}

int pfa_fa3_fwd(const pfa_fa3_args* a, void* stream) {
    const int st = check(a);
    if (st != PFA_OK) return st;

    pfa::FwdParams p;
    p.q = a->q; p.k = a->k; p.v = a->v; p.o = a->o;
    p.lse = a->lse; p.seqlens_k = a->seqlens_k; p.key_mask = a->key_mask;
    p.q_sb = a->q_stride_b; p.q_sh = a->q_stride_h; p.q_ss = a->q_stride_s;
    p.k_sb = a->k_stride_b; p.k_sh = a->k_stride_h; p.k_ss = a->k_stride_s;
    p.v_sb = a->v_stride_b; p.v_sh = a->v_stride_h; p.v_ss = a->v_stride_s;
    p.o_sb = a->o_stride_b; p.o_sh = a->o_stride_h; p.o_ss = a->o_stride_s;
    p.km_sb = a->key_mask_stride_b;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.nqblk = (a->Sq + pfa::BLOCK_M - 1) / pfa::BLOCK_M;
    p.scale_log2 = a->softmax_scale * 1.4426950408889634f;

    const Variant& v = pick(a->dtype_in, a->D, a->causal, a->flags & PFA_FLAG_SPLIT_P, a->dtype_out == PFA_DTYPE_FP32);
    const unsigned grid = (unsigned)(p.nqblk * a->B * a->H);
    void* kargs[] = {&p};
    int prev_dev = -1;
    hipError_t e = hipGetDevice(&prev_dev);
    if (e == hipSuccess && prev_dev != a->device_id) e = hipSetDevice(a->device_id);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        (void)hipGetLastError();
        return PFA_ERR_DEVICE;
    }
    e = hipLaunchKernel(v.fn, dim3(grid), dim3(pfa::NTHREADS), kargs, (size_t)v.lds_bytes, (hipStream_t)stream);
    if (prev_dev != a->device_id) (void)hipSetDevice(prev_dev);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        (void)hipGetLastError();
        return PFA_ERR_LAUNCH;
    }
    return PFA_OK;
}

}  // extern "C"

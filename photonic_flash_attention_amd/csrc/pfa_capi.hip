// pfa_capi.hip -- the C ABI of libpfa_hip.so (see include/pfa_hip.h for the contract and the
// reference lines each entry point replaces).  Host side only validates, picks a kernel variant
// and enqueues it; no allocation, no synchronisation; the only process-wide state is pfa_p4.hip's per-device module handle.
#include "pfa_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "fa3_fwd_kernel.h"
#include "fa3_weights_kernel.h"
#include "fa3_fwd_f32_kernel.h"
#include "pfa_p4.h"

namespace pfa { const void* w4_kernel(int dtype, bool causal, bool out32); }   // pfa_w4.hip

namespace {

thread_local int g_last_hip_error = 0;

struct Variant {
    const void* fn;
    char name[96];
    int lds_bytes;
    bool grid3 = false;    // 4-wave kernel: (head-in-XCD-group, Q block, group) arrive as blockIdx.x/y/z -- no divisions in its prologue
    int nthreads;
    int block_m;
    int xcd_group = 0;
    bool p4 = false;       // the persistent assembly kernel (pfa_p4.hip): launched through its own module, fn unused
    int p4_grid = 0;
};

template <typename T, int D, bool C, bool S, bool K, int VAR, typename OT>
Variant mk(const char* tn, const char* on) {
    Variant v;
    v.fn = (const void*)&pfa::fa3_fwd_kernel<T, D, C, S, K, VAR, OT>;
    snprintf(v.name, sizeof(v.name), "fa3_fwd_%s_d%d_%s%s%s_%s_v%d", tn, D, C ? "causal" : "full", S ? "_splitp" : "",
             K ? "_kmask" : "", on, VAR);
    v.lds_bytes = ((VAR & pfa::VAR_STAGE2) ? 4 : 2) * 2 * pfa::BLOCK_N * D * 2;
    if (VAR & pfa::VAR_RING3) v.lds_bytes = 3 * 2 * pfa::BLOCK_N * D * 2;
    if (VAR & pfa::VAR_STAGGER) v.lds_bytes = 5 * pfa::BLOCK_N * D * 2;   // K ring 2 + V ring 3
    v.nthreads = ((VAR & pfa::VAR_NW4) ? 4 : 8) * 64;
    v.block_m = v.nthreads / 2;
    return v;
}

// Production variants: every (dtype, D, causal, split, kmask, out) at VAR_DEFAULT.
template <typename T, int D, bool C, bool S, bool K>
Variant by_out(bool out32, const char* tn) {
    return out32 ? mk<T, D, C, S, K, pfa::VAR_DEFAULT, float>(tn, "o32") : mk<T, D, C, S, K, pfa::VAR_DEFAULT, T>(tn, "o16");
}
template <typename T, int D, bool C, bool S>
Variant by_kmask(bool kmask, bool out32, const char* tn) {
    return kmask ? by_out<T, D, C, S, true>(out32, tn) : by_out<T, D, C, S, false>(out32, tn);
}
template <typename T, int D, bool C>
Variant by_split(bool split, bool kmask, bool out32, const char* tn) {
    return split ? by_kmask<T, D, C, true>(kmask, out32, tn) : by_kmask<T, D, C, false>(kmask, out32, tn);
}
template <typename T, int D>
Variant by_causal(bool causal, bool split, bool kmask, bool out32, const char* tn) {
    return causal ? by_split<T, D, true>(split, kmask, out32, tn) : by_split<T, D, false>(split, kmask, out32, tn);
}
template <typename T>
Variant by_d(int D, bool causal, bool split, bool kmask, bool out32, const char* tn) {
    return D == 128 ? by_causal<T, 128>(causal, split, kmask, out32, tn) : by_causal<T, 64>(causal, split, kmask, out32, tn);
}


// persistent 4 waves x 64 rows in assembly (gen_fa3_fwd_p4.py / pfa_p4.hip)
Variant p4_variant(const pfa_fa3_args* a, bool causal) {
    Variant v;
    v.fn = nullptr;
    snprintf(v.name, sizeof(v.name), "fa3_fwd_p4_%s_d%d_%s%s_%s", a->dtype_in == PFA_DTYPE_BF16 ? "bf16" : "fp16", a->D, causal ? "causal" : "full",
             pfa::p4_flavour(a) == 1 ? "_km" : (pfa::p4_flavour(a) == 2 ? "_kl" : ""), a->dtype_out == PFA_DTYPE_FP32 ? "splitp_o32" : "o16");
    v.p4 = true;
    v.p4_grid = pfa::p4_workgroups(a);
    v.lds_bytes = 0;
    v.nthreads = 256;
    v.block_m = 256;
    return v;
}

// 4 waves x 64 rows (fa3_fwd_w4_kernel.h): D = 128, single P, no element mask
Variant w4_variant(const pfa_fa3_args* a, bool causal, bool out32) {
    Variant v;
    v.fn = pfa::w4_kernel(a->dtype_in == PFA_DTYPE_BF16 ? 0 : 1, causal, out32);
    snprintf(v.name, sizeof(v.name), "fa3_fwd_w4_%s_d128_%s_%s", a->dtype_in == PFA_DTYPE_BF16 ? "bf16" : "fp16",
             causal ? "causal" : "full", out32 ? "o32" : "o16");
    v.grid3 = true;
    v.lds_bytes = 8 * pfa::BLOCK_N * 128 * 2;      // K ring 2 + V ring 2 tiles, then the Q block's 64-KiB landing zone
    v.nthreads = 256;
    v.block_m = 256;
    // block order: an XCD walks its heads in groups of 4 (2 if 4 does not divide them): fewer heads' K/V live in its 4-MiB L2
    // at a time -- FETCH_SIZE -30 % at C3, S = 2048 x 256 heads +5 %, C3 +1 %, C4 / C5 unchanged (variant 49: the old order)
    const int hpx = (a->B * a->H) % 8 == 0 ? (a->B * a->H) / 8 : 0;
    v.xcd_group = (hpx > 0 && hpx % 4 == 0) ? 4 : ((hpx > 0 && hpx % 2 == 0) ? 2 : 0);
    return v;
}

Variant pick(const pfa_fa3_args* a) {
    const bool causal = a->causal != 0, split = (a->flags & PFA_FLAG_SPLIT_P) != 0;
    const bool kmask = a->key_mask != nullptr || a->mask != nullptr;
    const bool out32 = a->dtype_out == PFA_DTYPE_FP32;
    const unsigned var = (a->flags >> 8) & 0xffu;   // 0 = production default
    // The 4-wave x 64-row kernel (D = 128, single P, no element mask) wins once a workgroup streams enough key tiles to
    // amortise its fill and drain (one workgroup per CU: nothing overlaps them): from ~16 tiles per workgroup on
    // (tools/ab_bench.py, one box, against the 8-wave kernel: S1K +2.6 %, S2Kc +0.5 %, C3 +4 %, C4 / C5 / S8Kc +5 %, S2K +8 %;
    // below: S512 +1 %, S1Kc -6 %, S512c -6 %, S256 -5 %).  Variant 43 forces it, 44 forces the 8-wave kernel (A/B).
    const bool w4_ok = a->D == 128 && !split && !kmask && (int64_t)a->B * a->H * a->H < (1ll << 32);   // last: its multiply-high head index
    const int64_t avg_tiles = (causal ? (int64_t)a->Sk / 2 : (int64_t)a->Sk) / pfa::BLOCK_N;
    // Production selectors (A/B and tests): 43 = the 4-wave HIP kernel, 44 = the 8-wave kernel, 45 = the persistent assembly kernel.
    // The persistent kernel takes every aligned problem (pfa::p4_eligible): without a cold fill per Q block it also wins on short
    // sequences (same box, against the better HIP kernel: S256 +26 %, S512 +22 %, S512 causal +18 %, S1024 +12 %, S1024 causal
    // +24 %, S2048 causal +12 %: profiles/r02_p4_experiments.txt); the 4-wave HIP kernel keeps the other long problems.
    if (pfa::p4_eligible(a) && (var == 45 || var == 0)) {
        Variant v = p4_variant(a, causal);
        // D = 64 too since round 3: with the fast loop (no row max) and the mid-phase barrier the persistent kernel beats the 8-wave HIP kernel
        // on every D = 64 shape tried, also with several units per CU (same box, selector 44 -> 45: C2 527 -> 615 TFLOP/s, B16 H16 S2048 922 ->
        // 1000, B4 H12 S2048 causal 599 -> 681, B4 H16 S4096 causal 904 -> 925: profiles/r03_p4_experiments.txt).
        if (v.p4_grid > 0) return v;   // (0: the code object did not load on this device -- fall through to the HIP kernels)
    }
    if (w4_ok && (var == 43 || ((var == 0 || var == 45) && avg_tiles >= 16))) return w4_variant(a, causal, out32);
    return a->dtype_in == PFA_DTYPE_BF16 ? by_d<__bf16>(a->D, causal, split, kmask, out32, "bf16")
                                         : by_d<_Float16>(a->D, causal, split, kmask, out32, "fp16");
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool mult8(int64_t s) { return (s % 8) == 0; }

int check(const pfa_fa3_args* a) {
    if (!a) return PFA_ERR_NULL;
    if (a->size != sizeof(pfa_fa3_args)) return PFA_ERR_STRUCT_SIZE;
    if (a->flags & ~(PFA_FLAG_SPLIT_P | PFA_FLAG_NO_XCD_MAP | PFA_FLAG_VARIANT_MASK)) return PFA_ERR_FLAGS;
    {   // the library knows three kernel selectors (43 / 44 / 45, see pick())
        const unsigned var = (a->flags & PFA_FLAG_VARIANT_MASK) >> 8;
        if (var != 0 && var != 43 && var != 44 && var != 45) return PFA_ERR_FLAGS;
    }
    if (!a->q || !a->k || !a->v || !a->o) return PFA_ERR_NULL;
    if (a->key_mask && a->mask) return PFA_ERR_FLAGS;
    if (a->kv_group < 0 || a->reserve_cus < 0 || a->reserved1 != 0 || (a->kv_group > 1 && a->H % a->kv_group != 0)) return PFA_ERR_SHAPE;
    if (a->drop_mask && (a->dtype_in != PFA_DTYPE_FP32 || !(a->drop_scale >= 1.f) || !isfinite(a->drop_scale))) return PFA_ERR_FLAGS;
    if (a->B <= 0 || a->H <= 0 || a->Sq <= 0 || a->Sk <= 0) return PFA_ERR_SHAPE;
    if (a->D != 64 && a->D != 128) return PFA_ERR_HEAD_DIM;
    if (a->dtype_in != PFA_DTYPE_BF16 && a->dtype_in != PFA_DTYPE_FP16 && a->dtype_in != PFA_DTYPE_FP32) return PFA_ERR_DTYPE;
    if (a->dtype_out != a->dtype_in && a->dtype_out != PFA_DTYPE_FP32) return PFA_ERR_DTYPE;
    if (a->dtype_in == PFA_DTYPE_FP32) {       // the exact fp32 kernel (fa3_fwd_f32_kernel.h): 16-byte rows, no kernel selector, no split P
        if (a->flags & (PFA_FLAG_SPLIT_P | PFA_FLAG_VARIANT_MASK)) return PFA_ERR_FLAGS;
        const int64_t st4[] = {a->q_stride_b, a->q_stride_h, a->q_stride_s, a->k_stride_b, a->k_stride_h, a->k_stride_s,
                               a->v_stride_b, a->v_stride_h, a->v_stride_s};
        for (int64_t s : st4)
            if (s % 4 != 0) return PFA_ERR_STRIDE;
        if (!a->q || !a->k || !a->v || !a->o) return PFA_ERR_NULL;
        if (!aligned16(a->q) || !aligned16(a->k) || !aligned16(a->v) || (reinterpret_cast<uintptr_t>(a->o) & 3u)) return PFA_ERR_ALIGN;
        if (!(a->softmax_scale > 0.f) || !isfinite(a->softmax_scale)) return PFA_ERR_SHAPE;
        if ((int64_t)((a->Sq + 63) / 64) * a->B * a->H > 0x7fffffffLL) return PFA_ERR_SHAPE;
        if (a->lse && (reinterpret_cast<uintptr_t>(a->lse) & 3u)) return PFA_ERR_ALIGN;
        const int64_t rows[] = {a->q_stride_s, a->k_stride_s, a->v_stride_s, a->o_stride_s};       // rows at least D apart, 32-bit slabs
        for (int64_t s : rows)
            if (s < a->D) return PFA_ERR_STRIDE;
        if (((int64_t)(a->Sq - 1) * a->q_stride_s + a->D) * 4 > 0x7fffffffLL || ((int64_t)(a->Sk - 1) * a->k_stride_s + a->D) * 4 > 0x7fffffffLL ||
            ((int64_t)(a->Sk - 1) * a->v_stride_s + a->D) * 4 > 0x7fffffffLL || ((int64_t)(a->Sq - 1) * a->o_stride_s + a->D) * 4 > 0x7fffffffLL)
            return PFA_ERR_SHAPE;
        return PFA_OK;
    }
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
    const int64_t nq = (a->Sq + 127) / 128;
    if (nq * a->B * a->H > 0x7fffffffLL) return PFA_ERR_SHAPE;
    // K/V slabs of one (batch, head) are addressed through 32-bit buffer descriptors
    if (((int64_t)(a->Sk - 1) * a->k_stride_s + a->D) * 2 > 0x7fffffffLL) return PFA_ERR_SHAPE;
    if (((int64_t)(a->Sk - 1) * a->v_stride_s + a->D) * 2 > 0x7fffffffLL) return PFA_ERR_SHAPE;
    if (((int64_t)(a->Sq - 1) * a->q_stride_s + a->D) * 2 > 0x7fffffffLL) return PFA_ERR_SHAPE;   // the 4-wave kernel's Q DMA
    if (a->k_stride_s < 0 || a->v_stride_s < 0 || a->k_stride_s * 64 > 0x3fffffffLL || a->v_stride_s * 64 > 0x3fffffffLL)
        return PFA_ERR_STRIDE;
    return PFA_OK;
}

}  // namespace

namespace {
template <typename T, int D, bool C, bool K>
const void* weights_fn(bool w32) {
    return w32 ? (const void*)&pfa::fa3_weights_kernel<T, D, C, K, float> : (const void*)&pfa::fa3_weights_kernel<T, D, C, K, T>;
}
template <typename T, int D>
const void* weights_fn_ck(bool causal, bool kmask, bool w32) {
    if (causal) return kmask ? weights_fn<T, D, true, true>(w32) : weights_fn<T, D, true, false>(w32);
    return kmask ? weights_fn<T, D, false, true>(w32) : weights_fn<T, D, false, false>(w32);
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
        case PFA_ERR_FLAGS: return "unknown flag bits, both key_mask and mask set, or drop_mask without fp32 operands";
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
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return 0;
    int herr = 0;                   // load the assembly kernels' code object now: no later call (or graph capture) has to
    if (pfa::p4_prepare(device_id, &herr) != PFA_OK) g_last_hip_error = herr;     // (the HIP kernels still serve the device)
    return 1;
}

int pfa_fa3_prepare(int device_id) {
    int herr = 0;
    const int st = pfa::p4_prepare(device_id, &herr);
    if (st != PFA_OK) g_last_hip_error = herr;
    return st;
}

// Masks are condensed into one 64-bit word per mask row and 64-key tile before the forward (fa3_maskbits_kernel): geometry of
// the word array for `a` -- the mask's own (un-broadcast) extents and the word strides the kernels use
struct MaskBits {
    int Bm = 0, Hm = 0, Qm = 0, nt = 0;
    int64_t sb = 0, sh = 0, sq = 0, sk = 0;      // byte strides of the source mask
    int64_t ob = 0, oh = 0, oq = 0;              // word strides of the result (0 = broadcast)
    const uint8_t* src = nullptr;
    int ngran = 0;                               // 256-row granules of the mask (fa3_maskrange_kernel: first / last visible tile of each)
    size_t word_bytes() const { return src ? (size_t)Bm * Hm * Qm * nt * sizeof(unsigned long long) : 0; }
    size_t bytes() const { return src ? word_bytes() + (size_t)Bm * Hm * ngran * pfa::RANGE_PARTS * 2 * sizeof(int) : 0; }
};
static MaskBits mask_bits(const pfa_fa3_args* a) {
    MaskBits m;
    m.nt = (a->Sk + 63) / 64;
    if (a->key_mask) {
        m.src = a->key_mask; m.Bm = a->B; m.Hm = 1; m.Qm = 1;
        m.sb = a->key_mask_stride_b; m.sk = 1;
    } else if (a->mask) {
        m.src = a->mask;
        m.Bm = a->mask_stride_b ? a->B : 1; m.Hm = a->mask_stride_h ? a->H : 1; m.Qm = a->mask_stride_q ? a->Sq : 1;
        m.sb = a->mask_stride_b; m.sh = a->mask_stride_h; m.sq = a->mask_stride_q; m.sk = a->mask_stride_k;
    } else {
        return m;
    }
    if (m.Qm > 65535 || (int64_t)m.Bm * m.Hm > 65535) { m.src = nullptr; return m; }      // launch limits: the byte path serves these
    m.ngran = (m.Qm + 255) / 256;
    m.oq = m.Qm > 1 ? m.nt : 0;
    m.oh = m.Hm > 1 ? (int64_t)m.Qm * m.nt : 0;
    m.ob = m.Bm > 1 ? (int64_t)m.Hm * m.Qm * m.nt : 0;
    return m;
}

static bool launch_mask_bits(const MaskBits& mb, const pfa_fa3_args* a, void* stream) {
    const bool wide = mb.sk == 1 && a->Sk % 16 == 0 && mb.sb % 16 == 0 && mb.sh % 16 == 0 && mb.sq % 16 == 0 && ((uintptr_t)mb.src & 15) == 0;
    if (wide)       // keys contiguous and 16-byte aligned: 16 mask bytes per lane
        hipLaunchKernelGGL(pfa::fa3_maskbits16_kernel<0>, dim3((unsigned)(((mb.nt + 15) / 16 + 3) / 4), (unsigned)mb.Qm, (unsigned)(mb.Bm * mb.Hm)),
                           dim3(256), 0, (hipStream_t)stream, mb.src, mb.sb, mb.sh, mb.sq, mb.Hm, a->Sk, mb.nt, (unsigned long long*)a->workspace,
                           mb.ob, mb.oh, mb.oq);
    else
        hipLaunchKernelGGL(pfa::fa3_maskbits_kernel<0>, dim3((unsigned)((mb.nt + 3) / 4), (unsigned)mb.Qm, (unsigned)(mb.Bm * mb.Hm)), dim3(256), 0,
                           (hipStream_t)stream, mb.src, mb.sb, mb.sh, mb.sq, mb.sk, mb.Hm, a->Sk, mb.nt, (unsigned long long*)a->workspace,
                           mb.ob, mb.oh, mb.oq);
    if (hipGetLastError() != hipSuccess) return false;
    hipLaunchKernelGGL(pfa::fa3_maskrange_kernel<256>, dim3((unsigned)(mb.ngran * pfa::RANGE_PARTS), (unsigned)(mb.Bm * mb.Hm)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)a->workspace, mb.ob, mb.oh, mb.oq, mb.Hm, mb.Qm, mb.nt,
                       (int*)((char*)a->workspace + mb.word_bytes()), mb.ngran);
    return hipGetLastError() == hipSuccess;
}

size_t pfa_fa3_workspace_bytes(const pfa_fa3_args* a) { return (a && a->dtype_in != PFA_DTYPE_FP32) ? mask_bits(a).bytes() : 0; }

int pfa_fa3_check(const pfa_fa3_args* a) { return check(a); }

int pfa_fa3_describe(const pfa_fa3_args* a, char* buf, size_t n) {
    const int st = check(a);
    if (st != PFA_OK) return st;
    if (a->dtype_in == PFA_DTYPE_FP32) {
        if (buf && n) snprintf(buf, n, "fa3_fwd_f32_mfma_d%d_exact", a->D);
        return ((a->Sq + pfa::F32_BM - 1) / pfa::F32_BM) * a->B * a->H;
    }
    const Variant v = pick(a);
    if (buf && n) {
        strncpy(buf, v.name, n - 1);
        buf[n - 1] = 0;
    }
    if (v.p4) return v.p4_grid;
    const int nq = (a->Sq + v.block_m - 1) / v.block_m;
    return nq * a->B * a->H;
}

static int launch_f32(const pfa_fa3_args* a, void* stream) {
    pfa::F32Params p;
    p.q = (const float*)a->q; p.k = (const float*)a->k; p.v = (const float*)a->v; p.o = (float*)a->o;
    p.lse = a->lse; p.seqlens_k = a->seqlens_k;
    if (a->mask) {
        p.mask = a->mask; p.m_sb = a->mask_stride_b; p.m_sh = a->mask_stride_h; p.m_sq = a->mask_stride_q; p.m_sk = a->mask_stride_k;
    } else {
        p.mask = a->key_mask; p.m_sb = a->key_mask_stride_b; p.m_sh = 0; p.m_sq = 0; p.m_sk = 1;
    }
    p.q_sb = a->q_stride_b; p.q_sh = a->q_stride_h; p.q_ss = a->q_stride_s;
    p.k_sb = a->k_stride_b; p.k_sh = a->k_stride_h; p.k_ss = a->k_stride_s;
    p.v_sb = a->v_stride_b; p.v_sh = a->v_stride_h; p.v_ss = a->v_stride_s;
    p.o_sb = a->o_stride_b; p.o_sh = a->o_stride_h; p.o_ss = a->o_stride_s;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.kv_group = a->kv_group > 1 ? a->kv_group : 1;
    p.causal = a->causal != 0;
    p.scale = a->softmax_scale;
    p.drop_mask = a->drop_mask;
    p.drop_scale = a->drop_scale;
    const void* fn = a->D == 128 ? (const void*)&pfa::fa3_fwd_f32_kernel<128> : (const void*)&pfa::fa3_fwd_f32_kernel<64>;
    const int lds = a->D == 128 ? pfa::f32_lds_bytes<128>() : pfa::f32_lds_bytes<64>();
    int prev_dev = -1;
    hipError_t e = hipGetDevice(&prev_dev);
    if (e == hipSuccess && prev_dev != a->device_id) e = hipSetDevice(a->device_id);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        (void)hipGetLastError();
        return PFA_ERR_DEVICE;
    }
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    void* kargs[] = {&p};
    e = hipLaunchKernel(fn, dim3((unsigned)(((a->Sq + pfa::F32_BM - 1) / pfa::F32_BM) * a->B * a->H)), dim3(256), kargs, (size_t)lds, (hipStream_t)stream);
    if (prev_dev != a->device_id) (void)hipSetDevice(prev_dev);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        (void)hipGetLastError();
        return PFA_ERR_LAUNCH;
    }
    return PFA_OK;
}

int pfa_fa3_fwd(const pfa_fa3_args* a, void* stream) {
    const int st = check(a);
    if (st != PFA_OK) return st;
    if (a->dtype_in == PFA_DTYPE_FP32) return launch_f32(a, stream);

    pfa::FwdParams p;
    p.q = a->q; p.k = a->k; p.v = a->v; p.o = a->o;
    p.lse = a->lse; p.seqlens_k = a->seqlens_k;
    if (a->mask) {
        p.mask = a->mask; p.m_sb = a->mask_stride_b; p.m_sh = a->mask_stride_h; p.m_sq = a->mask_stride_q; p.m_sk = a->mask_stride_k;
    } else {
        p.mask = a->key_mask; p.m_sb = a->key_mask_stride_b; p.m_sh = 0; p.m_sq = 0; p.m_sk = 1;
    }
    p.q_sb = a->q_stride_b; p.q_sh = a->q_stride_h; p.q_ss = a->q_stride_s;
    p.k_sb = a->k_stride_b; p.k_sh = a->k_stride_h; p.k_ss = a->k_stride_s;
    p.v_sb = a->v_stride_b; p.v_sh = a->v_stride_h; p.v_ss = a->v_stride_s;
    p.o_sb = a->o_stride_b; p.o_sh = a->o_stride_h; p.o_ss = a->o_stride_s;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.dbg = (unsigned long long*)a->workspace;   // only the diagnostic VAR_STAMP variant writes it
    // mask + enough workspace: one word per row and tile instead of a mask byte per score (without workspace the byte path runs)
    const MaskBits mb = mask_bits(a);
    const bool use_mbits = mb.src && a->workspace && a->workspace_bytes >= mb.bytes();
    p.mbits = use_mbits ? (const unsigned long long*)a->workspace : nullptr;
    p.mb_sb = mb.ob; p.mb_sh = mb.oh; p.mb_sq = mb.oq;
    if (use_mbits) {
        p.mrange = (const int*)((const char*)a->workspace + mb.word_bytes());
        p.mr_sb = mb.Bm > 1 ? (int64_t)mb.Hm * mb.ngran : 0;
        p.mr_sh = mb.Hm > 1 ? mb.ngran : 0;
        p.mr_q = mb.Qm > 1 ? 1 : 0;
    }
    const Variant v = pick(a);
    if (v.p4) {
        int herr = 0;
        const int st4 = pfa::p4_launch(a, stream, &herr);
        if (st4 != PFA_OK) g_last_hip_error = herr;
        return st4;
    }
    p.nqblk = (a->Sq + v.block_m - 1) / v.block_m;
    p.kv_group = a->kv_group > 1 ? a->kv_group : 1;
    p.xcd_group = v.xcd_group;
    p.scale_log2 = a->softmax_scale * 1.4426950408889634f;
    p.magic_h = (uint32_t)((1ull << 32) / (uint64_t)a->H) + 1u;
    p.magic_g = (uint32_t)((1ull << 32) / (uint64_t)p.kv_group) + 1u;

    dim3 grid((unsigned)(p.nqblk * a->B * a->H));
    if (v.grid3) {
        // linear dispatch order is x fastest and workgroup n runs on XCD n % 8: x = XCD + 8 * (head within the XCD's current
        // group of G), y = Q block rank (heaviest first), z = group  ==  the order the kernel used to derive with divisions
        const int BH = a->B * a->H;
        if (p.xcd_group > 0 && (BH % (8 * p.xcd_group) != 0 || BH / (8 * p.xcd_group) > 65535)) p.xcd_group = 0;
        if (p.nqblk > 65535) return PFA_ERR_SHAPE;
        grid = p.xcd_group > 0 ? dim3(8u * p.xcd_group, (unsigned)p.nqblk, (unsigned)(BH / (8 * p.xcd_group)))
                               : dim3((unsigned)BH, (unsigned)p.nqblk, 1u);
    }
    void* kargs[] = {&p};
    int prev_dev = -1;
    hipError_t e = hipGetDevice(&prev_dev);
    if (e == hipSuccess && prev_dev != a->device_id) e = hipSetDevice(a->device_id);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        (void)hipGetLastError();
        return PFA_ERR_DEVICE;
    }
    if (use_mbits && !launch_mask_bits(mb, a, stream)) {
        if (prev_dev != a->device_id) (void)hipSetDevice(prev_dev);
        return PFA_ERR_LAUNCH;
    }
    if (v.lds_bytes > 64 * 1024)   // opt in to > 64 KiB of dynamic LDS (idempotent, per function)
        (void)hipFuncSetAttribute(v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds_bytes);
    e = hipLaunchKernel(v.fn, grid, dim3(v.nthreads), kargs, (size_t)v.lds_bytes, (hipStream_t)stream);
    if (prev_dev != a->device_id) (void)hipSetDevice(prev_dev);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        (void)hipGetLastError();
        return PFA_ERR_LAUNCH;
    }
    return PFA_OK;
}

int pfa_fa3_weights(const pfa_fa3_args* a, void* w, int32_t w_dtype, int64_t w_stride_b, int64_t w_stride_h,
                    int64_t w_stride_q, void* stream) {
    if (!a || !w) return PFA_ERR_NULL;
    pfa_fa3_args probe = *a;             // same validation as the forward; v/o are not read here
    if (!probe.v) probe.v = probe.k;
    if (!probe.o) probe.o = w;
    const int st = check(&probe);
    if (st != PFA_OK) return st;
    if (a->dtype_in == PFA_DTYPE_FP32) return PFA_ERR_DTYPE;      // the weights pass is MFMA only: hand it the 16-bit operands
    if (!a->lse) return PFA_ERR_NULL;
    if (w_dtype != a->dtype_in && w_dtype != PFA_DTYPE_FP32) return PFA_ERR_DTYPE;

    pfa::WeightsParams p;
    p.q = a->q; p.k = a->k; p.lse = a->lse; p.w = w; p.seqlens_k = a->seqlens_k;
    if (a->mask) {
        p.mask = a->mask; p.m_sb = a->mask_stride_b; p.m_sh = a->mask_stride_h; p.m_sq = a->mask_stride_q; p.m_sk = a->mask_stride_k;
    } else {
        p.mask = a->key_mask; p.m_sb = a->key_mask_stride_b; p.m_sh = 0; p.m_sq = 0; p.m_sk = 1;
    }
    p.q_sb = a->q_stride_b; p.q_sh = a->q_stride_h; p.q_ss = a->q_stride_s;
    p.k_sb = a->k_stride_b; p.k_sh = a->k_stride_h; p.k_ss = a->k_stride_s;
    p.w_sb = w_stride_b; p.w_sh = w_stride_h; p.w_sq = w_stride_q;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.nqblk = (a->Sq + 127) / 128;
    p.kv_group = a->kv_group > 1 ? a->kv_group : 1;
    p.scale_log2 = a->softmax_scale * 1.4426950408889634f;
    const MaskBits mb = mask_bits(a);          // as in pfa_fa3_fwd: the mask as words when the caller gave workspace
    const bool use_mbits = mb.src && a->workspace && a->workspace_bytes >= mb.bytes();
    p.mbits = use_mbits ? (const unsigned long long*)a->workspace : nullptr;
    p.mb_sb = mb.ob; p.mb_sh = mb.oh; p.mb_sq = mb.oq;
    const bool causal = a->causal != 0, kmask = p.mask != nullptr, w32 = w_dtype == PFA_DTYPE_FP32;
    const void* fn;
    if (a->dtype_in == PFA_DTYPE_BF16)
        fn = a->D == 128 ? weights_fn_ck<__bf16, 128>(causal, kmask, w32) : weights_fn_ck<__bf16, 64>(causal, kmask, w32);
    else
        fn = a->D == 128 ? weights_fn_ck<_Float16, 128>(causal, kmask, w32) : weights_fn_ck<_Float16, 64>(causal, kmask, w32);
    int prev_dev = -1;
    hipError_t e = hipGetDevice(&prev_dev);
    if (e == hipSuccess && prev_dev != a->device_id) e = hipSetDevice(a->device_id);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        (void)hipGetLastError();
        return PFA_ERR_DEVICE;
    }
    if (use_mbits && !launch_mask_bits(mb, a, stream)) {      // (again: the call may come without a forward before it)
        if (prev_dev != a->device_id) (void)hipSetDevice(prev_dev);
        return PFA_ERR_LAUNCH;
    }
    void* kargs[] = {&p};
    e = hipLaunchKernel(fn, dim3((unsigned)(p.nqblk * a->B * a->H)), dim3(256), kargs, 0, (hipStream_t)stream);
    if (prev_dev != a->device_id) (void)hipSetDevice(prev_dev);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        (void)hipGetLastError();
        return PFA_ERR_LAUNCH;
    }
    return PFA_OK;
}

}  // extern "C"

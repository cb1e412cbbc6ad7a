// pfa_bwd_capi.hip -- C ABI of the backward pass (include/pfa_hip.h: pfa_fa3_bwd).  Separate translation unit:
// the key-stationary dK/dV kernel runs one wave per SIMD on the whole register file and is compiled with
// -mllvm -amdgpu-mfma-vgpr-form (see Makefile) so that hipcc keeps its MFMA accumulators in place.
#include "pfa_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include "fa3_bwd_kernels.h"
#include "fa3_bwd_f32_kernel.h"

namespace {

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_bwd(const pfa_fa3_bwd_args* a) {
    if (!a) return PFA_ERR_NULL;
    if (a->size != sizeof(pfa_fa3_bwd_args)) return PFA_ERR_STRUCT_SIZE;
    if (a->flags) return PFA_ERR_FLAGS;
    if (a->kv_group < 0 || (a->kv_group > 1 && (a->H % a->kv_group || a->dtype == PFA_DTYPE_FP32))) return PFA_ERR_FLAGS;   // (fp32 kernels: one K/V head per query head)
    if (a->dtype == PFA_DTYPE_FP32) {          // fp32 backward kernels (fa3_bwd_f32_kernel.h): fp32 everything, the only path with dropout
        if (!a->q || !a->k || !a->v || !a->o || !a->dout || !a->lse || !a->dq || !a->dk || !a->dv) return PFA_ERR_NULL;
        if (a->B <= 0 || a->H <= 0 || a->Sq <= 0 || a->Sk <= 0) return PFA_ERR_SHAPE;
        if (a->D != 64 && a->D != 128) return PFA_ERR_HEAD_DIM;
        if (a->dtype_grad != PFA_DTYPE_FP32) return PFA_ERR_DTYPE;
        if (!(a->softmax_scale > 0.f) || !isfinite(a->softmax_scale)) return PFA_ERR_SHAPE;
        if (a->drop_mask && (!(a->drop_scale >= 1.f) || !isfinite(a->drop_scale))) return PFA_ERR_FLAGS;
        const int64_t st4[] = {a->q_stride_b, a->q_stride_h, a->q_stride_s, a->k_stride_b, a->k_stride_h, a->k_stride_s,
                               a->v_stride_b, a->v_stride_h, a->v_stride_s};
        for (int64_t s : st4)
            if (s % 4) return PFA_ERR_STRIDE;
        // dout is read in 16-byte pieces like q / k / v (fa3_bwd_f32_kernel MODE 0); o and the gradients are 4-byte accesses
        const int64_t do4[] = {a->do_stride_b, a->do_stride_h, a->do_stride_s};
        for (int64_t s : do4)
            if (s % 4) return PFA_ERR_STRIDE;
        if (!al16(a->q) || !al16(a->k) || !al16(a->v) || !al16(a->dout)) return PFA_ERR_ALIGN;
        const void* p4[] = {a->o, a->dq, a->dk, a->dv, a->lse};
        for (const void* p : p4)
            if (reinterpret_cast<uintptr_t>(p) & 3u) return PFA_ERR_ALIGN;
        // row strides: non-negative, rows at least D apart, every tensor addressable with 32-bit element offsets inside one (b, h) slab
        const int64_t rows[] = {a->q_stride_s, a->k_stride_s, a->v_stride_s, a->o_stride_s, a->do_stride_s, a->dq_stride_s, a->dk_stride_s,
                                a->dv_stride_s};
        for (int64_t s : rows)
            if (s < a->D) return PFA_ERR_STRIDE;
        const int64_t ext[] = {(int64_t)(a->Sq - 1) * a->q_stride_s, (int64_t)(a->Sk - 1) * a->k_stride_s, (int64_t)(a->Sk - 1) * a->v_stride_s,
                               (int64_t)(a->Sq - 1) * a->o_stride_s, (int64_t)(a->Sq - 1) * a->do_stride_s, (int64_t)(a->Sq - 1) * a->dq_stride_s,
                               (int64_t)(a->Sk - 1) * a->dk_stride_s, (int64_t)(a->Sk - 1) * a->dv_stride_s};
        for (int64_t e : ext)
            if ((e + a->D) * 4 > 0x7fffffffLL) return PFA_ERR_SHAPE;
        return PFA_OK;
    }
    if (a->drop_mask) return PFA_ERR_FLAGS;
    if (!a->q || !a->k || !a->v || !a->o || !a->dout || !a->lse || !a->dq || !a->dk || !a->dv || !a->delta) return PFA_ERR_NULL;
    if (a->B <= 0 || a->H <= 0 || a->Sq <= 0 || a->Sk <= 0) return PFA_ERR_SHAPE;
    if (a->D != 64 && a->D != 128) return PFA_ERR_HEAD_DIM;
    if (a->dtype != PFA_DTYPE_BF16 && a->dtype != PFA_DTYPE_FP16) return PFA_ERR_DTYPE;
    if (a->dtype_grad != a->dtype && a->dtype_grad != PFA_DTYPE_FP32) return PFA_ERR_DTYPE;
    if (!(a->softmax_scale > 0.f) || !isfinite(a->softmax_scale)) return PFA_ERR_SHAPE;
    const int64_t in_st[] = {a->q_stride_b, a->q_stride_h, a->q_stride_s, a->k_stride_b, a->k_stride_h, a->k_stride_s,
                             a->v_stride_b, a->v_stride_h, a->v_stride_s, a->o_stride_b, a->o_stride_h, a->o_stride_s,
                             a->do_stride_b, a->do_stride_h, a->do_stride_s};
    for (int64_t s : in_st)
        if (s % 8) return PFA_ERR_STRIDE;
    const int64_t out_st[] = {a->dq_stride_b, a->dq_stride_h, a->dq_stride_s, a->dk_stride_b, a->dk_stride_h,
                              a->dk_stride_s, a->dv_stride_b, a->dv_stride_h, a->dv_stride_s};
    for (int64_t s : out_st)      // gradient rows leave in 16-byte pieces
        if (s % (a->dtype_grad == PFA_DTYPE_FP32 ? 4 : 8)) return PFA_ERR_STRIDE;
    const void* ptrs[] = {a->q, a->k, a->v, a->o, a->dout, a->dq, a->dk, a->dv};
    for (const void* p : ptrs)
        if (!al16(p)) return PFA_ERR_ALIGN;
    const int64_t slabs[] = {((int64_t)(a->Sk - 1) * a->k_stride_s + a->D) * 2, ((int64_t)(a->Sk - 1) * a->v_stride_s + a->D) * 2,
                             ((int64_t)(a->Sq - 1) * a->q_stride_s + a->D) * 2, ((int64_t)(a->Sq - 1) * a->do_stride_s + a->D) * 2};
    for (int64_t s : slabs)
        if (s > 0x7fffffffLL || s <= 0) return PFA_ERR_SHAPE;
    return PFA_OK;
}

template <typename T, int D, bool C, bool K>
void pick_ck(bool g32, const void*& dq, const void*& dkdv) {
    dq = g32 ? (const void*)&pfa::fa3_bwd_dq_kernel<T, D, C, K, float> : (const void*)&pfa::fa3_bwd_dq_kernel<T, D, C, K, T>;
    dkdv = g32 ? (const void*)&pfa::fa3_bwd_dkdv_kernel<T, D, C, K, float> : (const void*)&pfa::fa3_bwd_dkdv_kernel<T, D, C, K, T>;
}
template <typename T, int D>
void pick_kernels(bool causal, bool kmask, bool g32, const void*& delta, const void*& dq, const void*& dkdv) {
    delta = (const void*)&pfa::fa3_bwd_delta_kernel<T, D>;
    if (causal) kmask ? pick_ck<T, D, true, true>(g32, dq, dkdv) : pick_ck<T, D, true, false>(g32, dq, dkdv);
    else kmask ? pick_ck<T, D, false, true>(g32, dq, dkdv) : pick_ck<T, D, false, false>(g32, dq, dkdv);
}

}  // namespace

// geometry of the condensed element mask in pfa_fa3_bwd_args.mask_workspace (see BwdParams): [row words][row ranges][column words][column ranges]
struct BwdMaskWs {
    int Bm = 0, Hm = 0, Qm = 0, nt = 0, ntq = 0, ngq = 0, nkb = 0;
    size_t roww = 0, rowr = 0, colw = 0, colr = 0;
    size_t bytes() const { return roww + rowr + colw + colr; }
};
static BwdMaskWs bwd_mask_ws(const pfa_fa3_bwd_args* a) {
    BwdMaskWs w;
    if (!a || !a->mask || a->dtype == PFA_DTYPE_FP32) return w;
    if (a->mask_stride_h == 0 && a->mask_stride_q == 0) return w;              // key-only masks never reach the element-mask kernels
    w.Bm = a->mask_stride_b ? a->B : 1; w.Hm = a->mask_stride_h ? a->H : 1; w.Qm = a->mask_stride_q ? a->Sq : 1;
    if (w.Qm > 65535 || (int64_t)w.Bm * w.Hm > 65535) return BwdMaskWs();     // launch limits: the byte paths serve these
    w.nt = (a->Sk + 63) / 64; w.ntq = (a->Sq + 63) / 64;
    if (w.ntq > 65535) return BwdMaskWs();
    w.ngq = (w.Qm + 255) / 256; w.nkb = (a->Sk + 127) / 128;
    const size_t bh = (size_t)w.Bm * w.Hm;
    w.roww = bh * w.Qm * w.nt * 8; w.rowr = bh * w.ngq * pfa::RANGE_PARTS * 8;
    w.colw = bh * (size_t)a->Sk * w.ntq * 8; w.colr = bh * w.nkb * pfa::RANGE_PARTS * 8;
    return w;
}

extern "C" {

size_t pfa_fa3_bwd_mask_workspace_bytes(const pfa_fa3_bwd_args* a) { return bwd_mask_ws(a).bytes(); }


size_t pfa_fa3_bwd_workspace_bytes(const pfa_fa3_bwd_args* a) {
    if (!a || a->B <= 0 || a->H <= 0 || a->Sq <= 0) return 0;
    return (size_t)a->B * a->H * a->Sq * sizeof(float);
}

int pfa_fa3_bwd(const pfa_fa3_bwd_args* a, void* stream) {
    const int st = check_bwd(a);
    if (st != PFA_OK) return st;
    if (a->dtype == PFA_DTYPE_FP32) {
        pfa::F32BwdParams p;
        p.q = (const float*)a->q; p.k = (const float*)a->k; p.v = (const float*)a->v; p.o = (const float*)a->o;
        p.dout = (const float*)a->dout; p.lse = a->lse;
        p.dq = (float*)a->dq; p.dk = (float*)a->dk; p.dv = (float*)a->dv;
        p.seqlens_k = a->seqlens_k; p.mask = a->mask; p.drop_mask = a->drop_mask;
        p.q_sb = a->q_stride_b; p.q_sh = a->q_stride_h; p.q_ss = a->q_stride_s;
        p.k_sb = a->k_stride_b; p.k_sh = a->k_stride_h; p.k_ss = a->k_stride_s;
        p.v_sb = a->v_stride_b; p.v_sh = a->v_stride_h; p.v_ss = a->v_stride_s;
        p.o_sb = a->o_stride_b; p.o_sh = a->o_stride_h; p.o_ss = a->o_stride_s;
        p.do_sb = a->do_stride_b; p.do_sh = a->do_stride_h; p.do_ss = a->do_stride_s;
        p.dq_sb = a->dq_stride_b; p.dq_sh = a->dq_stride_h; p.dq_ss = a->dq_stride_s;
        p.dk_sb = a->dk_stride_b; p.dk_sh = a->dk_stride_h; p.dk_ss = a->dk_stride_s;
        p.dv_sb = a->dv_stride_b; p.dv_sh = a->dv_stride_h; p.dv_ss = a->dv_stride_s;
        p.m_sb = a->mask_stride_b; p.m_sh = a->mask_stride_h; p.m_sq = a->mask_stride_q; p.m_sk = a->mask_stride_k;
        p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk; p.causal = a->causal != 0;
        p.scale = a->softmax_scale; p.drop_scale = a->drop_scale;
        const void* f0 = a->D == 128 ? (const void*)&pfa::fa3_bwd_f32_kernel<128, 0> : (const void*)&pfa::fa3_bwd_f32_kernel<64, 0>;
        const void* f1 = a->D == 128 ? (const void*)&pfa::fa3_bwd_f32_kernel<128, 1> : (const void*)&pfa::fa3_bwd_f32_kernel<64, 1>;
        const int lds = a->D == 128 ? pfa::f32_bwd_lds_bytes<128>() : pfa::f32_bwd_lds_bytes<64>();
        int prev = -1;
        hipError_t e = hipGetDevice(&prev);
        if (e == hipSuccess && prev != a->device_id) e = hipSetDevice(a->device_id);
        if (e != hipSuccess) { (void)hipGetLastError(); return PFA_ERR_DEVICE; }
        if (lds > 64 * 1024) {
            (void)hipFuncSetAttribute(f0, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        }
        void* kargs[] = {&p};
        e = hipLaunchKernel(f0, dim3((unsigned)(((a->Sq + pfa::F32B_BM - 1) / pfa::F32B_BM) * a->B * a->H)), dim3(256), kargs, (size_t)lds, (hipStream_t)stream);
        if (e == hipSuccess)
            e = hipLaunchKernel(f1, dim3((unsigned)(((a->Sk + pfa::F32B_BM - 1) / pfa::F32B_BM) * a->B * a->H)), dim3(256), kargs, (size_t)lds, (hipStream_t)stream);
        if (prev != a->device_id) (void)hipSetDevice(prev);
        if (e != hipSuccess) { (void)hipGetLastError(); return PFA_ERR_LAUNCH; }
        return PFA_OK;
    }
    pfa::BwdParams p;
    p.q = a->q; p.k = a->k; p.v = a->v; p.o = a->o; p.dout = a->dout; p.lse = a->lse; p.delta = a->delta;
    p.dq = a->dq; p.dk = a->dk; p.dv = a->dv; p.seqlens_k = a->seqlens_k;
    p.mask = a->mask; p.m_sb = a->mask_stride_b; p.m_sh = a->mask_stride_h; p.m_sq = a->mask_stride_q; p.m_sk = a->mask_stride_k;
    p.q_sb = a->q_stride_b; p.q_sh = a->q_stride_h; p.q_ss = a->q_stride_s;
    p.k_sb = a->k_stride_b; p.k_sh = a->k_stride_h; p.k_ss = a->k_stride_s;
    p.v_sb = a->v_stride_b; p.v_sh = a->v_stride_h; p.v_ss = a->v_stride_s;
    p.o_sb = a->o_stride_b; p.o_sh = a->o_stride_h; p.o_ss = a->o_stride_s;
    p.do_sb = a->do_stride_b; p.do_sh = a->do_stride_h; p.do_ss = a->do_stride_s;
    p.dq_sb = a->dq_stride_b; p.dq_sh = a->dq_stride_h; p.dq_ss = a->dq_stride_s;
    p.dk_sb = a->dk_stride_b; p.dk_sh = a->dk_stride_h; p.dk_ss = a->dk_stride_s;
    p.dv_sb = a->dv_stride_b; p.dv_sh = a->dv_stride_h; p.dv_ss = a->dv_stride_s;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.kv_group = a->kv_group > 1 ? a->kv_group : 1;
    // a mask of the keys only (the reference's 2-D [B,Sk] mask arrives as [B,1,1,Sk]) runs on the unmasked kernels: see BwdParams::keymask
    const bool key_only = a->mask && a->mask_stride_h == 0 && a->mask_stride_q == 0 && a->mask_stride_k == 1 && a->Sk % 4 == 0 &&
                          a->mask_stride_b % 4 == 0 && ((uintptr_t)a->mask & 3) == 0;
    p.keymask = key_only ? a->mask : nullptr;
    p.km_sb = a->mask_stride_b;
    if (key_only) p.mask = nullptr;
    p.mask_dw = (p.mask && a->mask_stride_k == 1 && a->Sk % 4 == 0 && a->mask_stride_b % 4 == 0 && a->mask_stride_h % 4 == 0 &&
                 a->mask_stride_q % 4 == 0 && ((uintptr_t)a->mask & 3) == 0) ? 1 : 0;
    p.scale = a->softmax_scale;
    p.scale_log2 = a->softmax_scale * 1.4426950408889634f;
    // element mask + scratch: words, transposed words and tile ranges (launched below, in front of the two kernels)
    const BwdMaskWs mw = p.mask ? bwd_mask_ws(a) : BwdMaskWs();
    const bool use_words = mw.bytes() > 0 && a->mask_workspace && a->mask_workspace_bytes >= mw.bytes();
    if (use_words) {
        char* ws = (char*)a->mask_workspace;
        p.mw_row = (const unsigned long long*)ws;
        p.mw_sq = mw.Qm > 1 ? mw.nt : 0; p.mw_sh = mw.Hm > 1 ? (int64_t)mw.Qm * mw.nt : 0; p.mw_sb = mw.Bm > 1 ? (int64_t)mw.Hm * mw.Qm * mw.nt : 0;
        p.rg_row = (const int*)(ws + mw.roww);
        p.rg_q = mw.Qm > 1 ? 1 : 0; p.rg_sh = mw.Hm > 1 ? mw.ngq : 0; p.rg_sb = mw.Bm > 1 ? (int64_t)mw.Hm * mw.ngq : 0;
        p.mw_col = (const unsigned long long*)(ws + mw.roww + mw.rowr);
        p.ntq = mw.ntq; p.cw_sh = mw.Hm > 1 ? (int64_t)a->Sk * mw.ntq : 0; p.cw_sb = mw.Bm > 1 ? (int64_t)mw.Hm * a->Sk * mw.ntq : 0;
        p.rg_col = (const int*)(ws + mw.roww + mw.rowr + mw.colw);
        p.crg_sh = mw.Hm > 1 ? mw.nkb : 0; p.crg_sb = mw.Bm > 1 ? (int64_t)mw.Hm * mw.nkb : 0;
    }

    const void *kdelta, *kdq, *kdkdv;
    const bool causal = a->causal != 0, g32 = a->dtype_grad == PFA_DTYPE_FP32, kmask = p.mask != nullptr;     // (not for key-only masks)
    if (a->dtype == PFA_DTYPE_BF16) {
        if (a->D == 128) pick_kernels<__bf16, 128>(causal, kmask, g32, kdelta, kdq, kdkdv);
        else pick_kernels<__bf16, 64>(causal, kmask, g32, kdelta, kdq, kdkdv);
    } else {
        if (a->D == 128) pick_kernels<_Float16, 128>(causal, kmask, g32, kdelta, kdq, kdkdv);
        else pick_kernels<_Float16, 64>(causal, kmask, g32, kdelta, kdq, kdkdv);
    }
    int prev = -1;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess && prev != a->device_id) e = hipSetDevice(a->device_id);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return PFA_ERR_DEVICE;
    }
    const int lds = 2 * 2 * pfa::BLOCK_N * a->D * 2;
    const int BH = a->B * a->H;
    void* args[] = {&p};
    const int rows_per_block = 256 / (a->D / 8);
    // (the dQ kernel computes delta = rowsum(dO o O) for its own rows and publishes it for the dK/dV kernel behind it;
    //  fa3_bwd_delta_kernel is kept for reference / diagnostics but is no longer launched)
    (void)kdelta;
    (void)rows_per_block;
    if (use_words) {
        char* ws = (char*)a->mask_workspace;
        const dim3 bh((unsigned)1, (unsigned)1, (unsigned)(mw.Bm * mw.Hm));
        const bool wide = a->mask_stride_k == 1 && a->Sk % 16 == 0 && a->mask_stride_b % 16 == 0 && a->mask_stride_h % 16 == 0 &&
                          a->mask_stride_q % 16 == 0 && ((uintptr_t)a->mask & 15) == 0;
        if (wide)
            hipLaunchKernelGGL(pfa::fa3_maskbits16_kernel<0>, dim3((unsigned)(((mw.nt + 15) / 16 + 3) / 4), (unsigned)mw.Qm, bh.z), dim3(256), 0,
                               (hipStream_t)stream, a->mask, a->mask_stride_b, a->mask_stride_h, a->mask_stride_q, mw.Hm, a->Sk, mw.nt,
                               (unsigned long long*)ws, p.mw_sb, p.mw_sh, p.mw_sq);
        else
            hipLaunchKernelGGL(pfa::fa3_maskbits_kernel<0>, dim3((unsigned)((mw.nt + 3) / 4), (unsigned)mw.Qm, bh.z), dim3(256), 0, (hipStream_t)stream,
                               a->mask, a->mask_stride_b, a->mask_stride_h, a->mask_stride_q, a->mask_stride_k, mw.Hm, a->Sk, mw.nt,
                               (unsigned long long*)ws, p.mw_sb, p.mw_sh, p.mw_sq);
        hipLaunchKernelGGL(pfa::fa3_maskrange_kernel<256>, dim3((unsigned)(mw.ngq * pfa::RANGE_PARTS), bh.z), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned long long*)ws, p.mw_sb, p.mw_sh, p.mw_sq, mw.Hm, mw.Qm, mw.nt, (int*)(ws + mw.roww), mw.ngq);
        // (the transposed words always have the problem's own row count: a mask without a row dimension is the same word in every row)
        hipLaunchKernelGGL(pfa::fa3_maskbitsT_kernel<0>, dim3((unsigned)((mw.nt + 3) / 4), (unsigned)mw.ntq, bh.z), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned long long*)ws, p.mw_sb, p.mw_sh, p.mw_sq, mw.Hm, mw.Qm, a->Sq, a->Sk, mw.nt, mw.ntq,
                           (unsigned long long*)(ws + mw.roww + mw.rowr));
        hipLaunchKernelGGL(pfa::fa3_maskrange_kernel<128>, dim3((unsigned)(mw.nkb * pfa::RANGE_PARTS), bh.z), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned long long*)(ws + mw.roww + mw.rowr), (int64_t)mw.Hm * a->Sk * mw.ntq, (int64_t)a->Sk * mw.ntq,
                           (int64_t)mw.ntq, mw.Hm, a->Sk, mw.ntq, (int*)(ws + mw.roww + mw.rowr + mw.colw), mw.nkb);
        if (hipGetLastError() != hipSuccess) {
            if (prev != a->device_id) (void)hipSetDevice(prev);
            return PFA_ERR_LAUNCH;
        }
    }
    p.nblk = (a->Sq + 255) / 256;
    e = hipLaunchKernel(kdq, dim3((unsigned)(p.nblk * BH)), dim3(512), args, (size_t)lds, (hipStream_t)stream);
    if (e == hipSuccess) {
        p.nblk = (a->Sk + 127) / 128;
        if (lds + 1024 > 64 * 1024)   // tile stages + the per-row constants exceed the default 64 KiB dynamic-LDS limit
            (void)hipFuncSetAttribute(kdkdv, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 1024);
        e = hipLaunchKernel(kdkdv, dim3((unsigned)(p.nblk * (BH / p.kv_group))), dim3(256), args, (size_t)lds + 1024, (hipStream_t)stream);   // a workgroup per key block and K/V head
    }
    if (prev != a->device_id) (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return PFA_ERR_LAUNCH;
    }
    return PFA_OK;
}

}  // extern "C"

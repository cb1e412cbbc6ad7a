// pfa_p4.hip -- host side of the persistent 4-wave forward (gen_fa3_fwd_p4.py): the kernel is gfx950 assembly assembled into a code
// object of its own (build/fa3_fwd_p4.hsaco), embedded here byte for byte and loaded once per device with hipModuleLoadData.
// The only process-wide state of the library: the per-device module handles, filled once under a mutex (pfa_fa3_prepare /
// pfa_device_supported do it eagerly) and never changed afterwards.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "pfa_hip.h"
#include "pfa_p4.h"
#include "build/fa3_fwd_p4_offsets.h"          // generated: P4_KA_* byte offsets of the kernarg block, P4_LDS_BYTES

// the code object, embedded by the assembler
__asm__(".section .rodata\n"
        ".global pfa_p4_hsaco_begin\n.global pfa_p4_hsaco_end\n"
        ".balign 4096\n"
        "pfa_p4_hsaco_begin:\n"
        ".incbin \"" PFA_P4_HSACO_PATH "\"\n"
        "pfa_p4_hsaco_end:\n"
        ".byte 0\n"
        ".previous\n");
extern "C" const unsigned char pfa_p4_hsaco_begin[];
extern "C" const unsigned char pfa_p4_hsaco_end[];

namespace pfa {

// kernarg block of fa3_fwd_p4_* (gen_fa3_fwd_p4.py `KA`); byte strides are 32-bit (the host refuses anything larger)
struct P4Params {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;
    uint32_t q_sb, q_sh, k_sb, k_sh, v_sb, v_sh, o_sb, o_sh;
    uint32_t q_ss, k_ss, v_ss, o_ss;
    uint32_t H, Sq, Sk, NB, NU, magic_NU, magic_H, kv_group, magic_G;
    float scale_log2, thr;
    uint32_t hx, xcd_mode, SL, nt_full, pad;
    unsigned long long* dbg;
    const int32_t* seqlens_k;      // not preloaded by the kernel: fetched in its item decode (null = every batch has Sk keys)
};
#define P4_CHECK(field, macro) static_assert(offsetof(P4Params, field) == macro, "kernarg layout of " #field)
P4_CHECK(q, P4_KA_Q); P4_CHECK(k, P4_KA_K); P4_CHECK(v, P4_KA_V); P4_CHECK(o, P4_KA_O); P4_CHECK(lse, P4_KA_LSE);
P4_CHECK(q_sb, P4_KA_Q_SB); P4_CHECK(q_sh, P4_KA_Q_SH); P4_CHECK(k_sb, P4_KA_K_SB); P4_CHECK(k_sh, P4_KA_K_SH);
P4_CHECK(v_sb, P4_KA_V_SB); P4_CHECK(v_sh, P4_KA_V_SH); P4_CHECK(o_sb, P4_KA_O_SB); P4_CHECK(o_sh, P4_KA_O_SH);
P4_CHECK(q_ss, P4_KA_Q_SS); P4_CHECK(k_ss, P4_KA_K_SS); P4_CHECK(v_ss, P4_KA_V_SS); P4_CHECK(o_ss, P4_KA_O_SS);
P4_CHECK(H, P4_KA_H); P4_CHECK(Sq, P4_KA_SQ); P4_CHECK(Sk, P4_KA_SK); P4_CHECK(NB, P4_KA_NB); P4_CHECK(NU, P4_KA_NU);
P4_CHECK(magic_NU, P4_KA_MAGIC_NU); P4_CHECK(magic_H, P4_KA_MAGIC_H); P4_CHECK(kv_group, P4_KA_KV_GROUP);
P4_CHECK(magic_G, P4_KA_MAGIC_G); P4_CHECK(scale_log2, P4_KA_SCALE_LOG2); P4_CHECK(thr, P4_KA_THR); P4_CHECK(hx, P4_KA_HX);
P4_CHECK(xcd_mode, P4_KA_XCD_MODE); P4_CHECK(SL, P4_KA_SL); P4_CHECK(nt_full, P4_KA_NT_FULL); P4_CHECK(dbg, P4_KA_DBG);
P4_CHECK(seqlens_k, P4_KA_SEQLENS);
static_assert(sizeof(P4Params) == P4_KARG_BYTES, "kernarg size");

namespace {
constexpr int MAX_DEV = 64;
struct DevMod {
    hipModule_t mod = nullptr;
    hipFunction_t fn[2][2][2][3][2] = {};  // [dtype][D: 128, 64][causal][plain, key mask, ragged][parity variant: fp32 store + split P]
    int n_cu = 0;
    int state = 0;                   // 0 = not loaded (yet), 1 = ready, -1 = permanently unavailable on this device
    int last_err = 0;                // hipError_t of the last failed attempt
};
DevMod g_mod[MAX_DEV];
std::mutex g_mu;

uint32_t magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d) + 1u; }   // n / d == mulhi(n, magic) while n * d < 2^32

// Load the code object on `dev` (once).  Permanent failures (the device is no gfx950 part, a kernel symbol is missing) latch
// state = -1; transient ones (a failing HIP call, e.g. a first call issued under stream capture) leave state = 0 so that a later
// call tries again; either way the hipError_t is kept for pfa_last_hip_error.  pfa_device_supported / pfa_fa3_prepare call this
// eagerly, so that no launch, describe or graph capture ever loads a module.
const DevMod* module_for(int dev, int* hip_err = nullptr) {
    if (dev < 0 || dev >= MAX_DEV) return nullptr;
    DevMod& m = g_mod[dev];
    if (__atomic_load_n(&m.state, __ATOMIC_ACQUIRE) == 1) return &m;          // the common case takes no lock
    std::lock_guard<std::mutex> lk(g_mu);
    if (m.state == 0) {
        auto fail = [&](hipError_t e, bool permanent) -> const DevMod* {
            m.last_err = (int)e;
            if (hip_err) *hip_err = (int)e;
            (void)hipGetLastError();
            if (permanent) m.state = -1;
            return nullptr;
        };
        int cur = -1;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess) return fail(e, false);
        struct Restore {       // the module belongs to the device that is current while it is loaded
            int cur, dev;
            Restore(int c, int d) : cur(c), dev(d) { if (c != d) (void)hipSetDevice(d); }
            ~Restore() { if (cur != dev) (void)hipSetDevice(cur); }
        } restore(cur, dev);
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) return fail(e, false);
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(hipErrorNoBinaryForGpu, true);
        m.n_cu = prop.multiProcessorCount;
        hipModule_t mod = nullptr;
        e = hipModuleLoadData(&mod, pfa_p4_hsaco_begin);
        if (e != hipSuccess) return fail(e, e == hipErrorNoBinaryForGpu || e == hipErrorInvalidImage || e == hipErrorSharedObjectInitFailed);
        static const char* dt[2] = {"bf16", "fp16"};
        static const char* cz[2] = {"full", "causal"};
        static const char* pv[2] = {"o16", "splitp_o32"};
        for (int d = 0; d < 2; ++d)
            for (int hd = 0; hd < 2; ++hd)
                for (int c = 0; c < 2; ++c)
                    for (int km = 0; km < 3; ++km)
                        for (int v = 0; v < 2; ++v) {
                            char name[64];
                            snprintf(name, sizeof(name), "fa3_fwd_p4_%s_d%d_%s%s_%s", dt[d], hd ? 64 : 128, cz[c], km == 1 ? "_km" : (km == 2 ? "_kl" : ""), pv[v]);
                            e = hipModuleGetFunction(&m.fn[d][hd][c][km][v], mod, name);
                            if (e != hipSuccess) {         // the production code object carries all 48 kernels: a missing one is a broken build
                                (void)hipModuleUnload(mod);
                                return fail(e, true);
                            }
                        }
        m.mod = mod;
        __atomic_store_n(&m.state, 1, __ATOMIC_RELEASE);
    } else if (m.state < 0 && hip_err) {
        *hip_err = m.last_err;
    }
    return m.state == 1 ? &m : nullptr;
}

bool fits_u32(int64_t x) { return x >= 0 && x <= 0xffffffffLL; }
}  // namespace

// 0 = the plain kernels (whole blocks and tiles), 1 = *_km_* ([B, Sk] key mask, whole blocks and tiles), 2 = *_kl_* (ragged: Sq no
// multiple of 256 or Sk no multiple of 128 -- rows past the end are kept out by buffer descriptors, keys past Sk by a computed mask word)
int p4_flavour(const pfa_fa3_args* a) {
    if (a->key_mask) return 1;
    const bool whole = a->Sq % 256 == 0 && a->Sk % 128 == 0;
    return whole && !a->seqlens_k ? 0 : 2;       // (per-batch key counts: the ragged kernels' length word, and tile counts cut to the length)
}

// Shapes the persistent kernel takes (everything else stays on the HIP kernels): D = 128 or 64; the fast variant (one P operand, 16-bit
// store) or the parity variant (split P AND fp32 store); no element mask; Sq >= 128 and Sk >= 193 of any length (ragged: *_kl_*); a
// [B, Sk] key mask with contiguous rows on whole blocks / tile pairs (*_km_*: the kernel reads the bytes itself); seqlens_k without the
// causal mask (an item's tile count is cut to its batch's keys), and with it on the ragged kernels (round 3: a block behind its batch's cut
// runs the tiles of the visible keys without a diagonal); under the causal mask Sq == Sk (units are heavy + light block pairs, with an odd
// block count the middle block is a unit of its own).
bool p4_eligible(const pfa_fa3_args* a) {
    const bool split = (a->flags & PFA_FLAG_SPLIT_P) != 0, out32 = a->dtype_out == PFA_DTYPE_FP32;
    if ((a->D != 128 && a->D != 64) || split != out32) return false;
    const int64_t osz = out32 ? 4 : 2;
    if (a->mask) return false;                                         // (element masks: the HIP kernels)
    if (a->key_mask && (a->key_mask_stride_b != a->Sk || (int64_t)a->B * a->Sk > 0x7fffffffLL)) return false;   // its 32-bit running byte offset
    const int64_t NBq = ((int64_t)a->Sq + 255) / 256;
    if (a->Sq < 128 || a->Sk < 193) return false;      // at least half a Q block of rows; at least four key tiles, the first three of them whole
    if (a->causal && a->Sq != a->Sk) return false;      // (units are (heavy, light) block pairs; an odd block count leaves the middle block alone)
    const int64_t st[] = {a->q_stride_b, a->q_stride_h, a->q_stride_s, a->k_stride_b, a->k_stride_h, a->k_stride_s,
                          a->v_stride_b, a->v_stride_h, a->v_stride_s};
    for (int64_t s : st)
        if (!fits_u32(s * 2)) return false;
    if (!fits_u32(a->o_stride_b * osz) || !fits_u32(a->o_stride_h * osz) || !fits_u32(a->o_stride_s * osz)) return false;
    // row strides: a wave's DMA offsets (row * stride + 256) and an item's output rows stay below 2^31; rows at least D elements apart
    if (a->q_stride_s < a->D || a->k_stride_s < a->D || a->v_stride_s < a->D || a->o_stride_s < a->D) return false;
    if ((int64_t)a->Sk * a->k_stride_s * 2 > 0x7fffffffLL || (int64_t)a->Sk * a->v_stride_s * 2 > 0x7fffffffLL) return false;
    if (256 * a->q_stride_s * 2 > 0x7fffffffLL || 256 * a->o_stride_s * osz > 0x7fffffffLL) return false;
    if ((a->o_stride_s * osz) % 16 != 0 || (a->o_stride_h * osz) % 16 != 0 || (a->o_stride_b * osz) % 16 != 0 ||
        (reinterpret_cast<uintptr_t>(a->o) & 15u)) return false;                                       // 16-byte row stores
    const int64_t BH = (int64_t)a->B * a->H, NB = NBq, NU = a->causal ? (NB + 1) / 2 : NB;
    if (BH * NU > (1ll << 24) || BH * a->H >= (1ll << 32)) return false;                                 // multiply-high ranges
    return true;
}

// Grid of the persistent kernel: one workgroup per CU, minus the CUs the caller wants left to another kernel (pfa_fa3_args.reserve_cus:
// a collective running beside the attention needs CUs of its own, the persistent workgroups never yield theirs), a multiple of 8 when
// the heads map onto the XCDs.
static int grid_for(const DevMod* m, const pfa_fa3_args* a) {
    int n = m->n_cu - (a->reserve_cus > 0 ? a->reserve_cus : 0);
    if (n < 8) n = 8;
    const int BH = a->B * a->H;
    return BH % 8 == 0 ? (n / 8) * 8 : n;
}

int p4_prepare(int device_id, int* hip_err) { return module_for(device_id, hip_err) ? PFA_OK : PFA_ERR_DEVICE; }

int p4_workgroups(const pfa_fa3_args* a) {
    const DevMod* m = module_for(a->device_id);
    return m ? grid_for(m, a) : 0;
}

// Enqueue.  Returns PFA_OK, or PFA_ERR_LAUNCH / PFA_ERR_DEVICE (hip error in *hip_err).
int p4_launch(const pfa_fa3_args* a, void* stream, int* hip_err) {
    const DevMod* m = module_for(a->device_id, hip_err);
    if (!m) return PFA_ERR_DEVICE;
    P4Params p;
    memset(&p, 0, sizeof(p));
    p.q = a->q; p.k = a->k; p.v = a->v; p.o = a->o; p.lse = a->lse;
    p.q_sb = (uint32_t)(a->q_stride_b * 2); p.q_sh = (uint32_t)(a->q_stride_h * 2); p.q_ss = (uint32_t)(a->q_stride_s * 2);
    p.k_sb = (uint32_t)(a->k_stride_b * 2); p.k_sh = (uint32_t)(a->k_stride_h * 2); p.k_ss = (uint32_t)(a->k_stride_s * 2);
    p.v_sb = (uint32_t)(a->v_stride_b * 2); p.v_sh = (uint32_t)(a->v_stride_h * 2); p.v_ss = (uint32_t)(a->v_stride_s * 2);
    const bool parity = a->dtype_out == PFA_DTYPE_FP32;
    const int64_t osz = parity ? 4 : 2;
    p.o_sb = (uint32_t)(a->o_stride_b * osz); p.o_sh = (uint32_t)(a->o_stride_h * osz); p.o_ss = (uint32_t)(a->o_stride_s * osz);
    p.H = (uint32_t)a->H; p.Sq = (uint32_t)a->Sq; p.Sk = (uint32_t)a->Sk;
    p.NB = (uint32_t)((a->Sq + 255) / 256);
    p.NU = a->causal ? (p.NB + 1) / 2 : p.NB;
    p.magic_NU = magic(p.NU); p.magic_H = magic(p.H);
    p.kv_group = a->kv_group > 1 ? (uint32_t)a->kv_group : 1u;
    p.magic_G = magic(p.kv_group);
    p.scale_log2 = a->softmax_scale * 1.4426950408889634f;
    p.thr = 8.0f / p.scale_log2;
    const int BH = a->B * a->H;
    const int grid = grid_for(m, a);
    p.xcd_mode = BH % 8 == 0 ? 1u : 0u;
    p.hx = p.xcd_mode ? (uint32_t)(BH / 8) : (uint32_t)BH;
    p.SL = p.xcd_mode ? (uint32_t)(grid / 8) : (uint32_t)grid;
    p.nt_full = (uint32_t)(((a->Sk + 127) / 128) * 2);          // 64-key tiles, an even number of them (the two S buffers alternate from tile 0)
    // diagnostic build only (P4_STAMP=1 make): per-wave cycle buckets go to the caller's workspace; the production kernel never reads it
    p.dbg = (a->workspace && a->workspace_bytes >= (size_t)grid * 4 * 16 * 4) ? (unsigned long long*)a->workspace : nullptr;
    // key-mask kernels: the same kernarg slot carries the mask bytes, `pad` is the kernel's running byte offset into them
    if (a->key_mask) p.dbg = (unsigned long long*)a->key_mask;
    p.seqlens_k = a->seqlens_k;

    size_t sz = sizeof(p);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    hipFunction_t fn = m->fn[a->dtype_in == PFA_DTYPE_BF16 ? 0 : 1][a->D == 64 ? 1 : 0][a->causal ? 1 : 0][p4_flavour(a)][parity ? 1 : 0];
    int prev = -1;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess && prev != a->device_id) e = hipSetDevice(a->device_id);
    if (e != hipSuccess) {
        if (hip_err) *hip_err = (int)e;
        (void)hipGetLastError();
        return PFA_ERR_DEVICE;
    }
    e = hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, nullptr, config);
    if (prev != a->device_id) (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        if (hip_err) *hip_err = (int)e;
        (void)hipGetLastError();
        return PFA_ERR_LAUNCH;
    }
    return PFA_OK;
}

}  // namespace pfa

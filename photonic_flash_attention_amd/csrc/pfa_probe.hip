// pfa_probe.hip -- measurement aid of bench.py (roofline.probe_tflops): what the matrix pipes of THIS device deliver under the load of
// the forward's tile loop, without its softmax -- a bare v_mfma_f32_32x32x16_bf16 stream on random operands, one wave per SIMD on every
// CU, every A operand re-read from LDS exactly as the tile loop reads its K fragments (16 ds_read_b128 per 64 MFMAs, each feeding two
// MFMAs) and its V^T fragments (32 ds_read_b64_tr_b16 per 64 MFMAs).  The chip lowers its clock under such a load (MI355X_MICROARCH.md,
// 'DVFS give-back'), so the nominal 2.5 PFLOP/s is not what a kernel can reach on random data; this number is.  Not part of the hot path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfa_hip.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PROBE_LDS = 64 * 1024;       // K ring + V ring of the forward at D = 128
constexpr int PROBE_WG_LDS = 96 * 1024;    // more than half the CU's LDS: one workgroup per CU

constexpr int AHEAD = 6;                   // fragments requested ahead of their MFMAs (<= 15 LDS reads outstanding)
constexpr int reads_of(int f) { return (f & 31) < 16 ? 1 : 2; }          // fragments 0..15: K by ds_read_b128; 16..31: V^T by two tr reads
constexpr int behind(int f) {              // LDS reads issued after fragment f's own at the time its MFMAs want it
    int n = 0;
    for (int g = f + 1; g < f + AHEAD; ++g) n += reads_of(g);
    return n;
}

struct Frag { uint64_t lo, hi; };

template <int F>
__device__ __forceinline__ void request(Frag& fr, uint32_t kbase, uint32_t vbase, uint32_t flip) {
    constexpr int f = F & 31;
    if constexpr (f < 16) {
        const uint32_t addr = (kbase ^ (uint32_t)((f & 7) << 5)) + (uint32_t)((f >> 3) * 8192) + flip;
        __uint128_t w;
        asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(addr));
        fr.lo = (uint64_t)w;
        fr.hi = (uint64_t)(w >> 64);
    } else {
        const uint32_t addr = vbase + (uint32_t)((f & 3) * 64 + ((f - 16) >> 2) * 4096) + flip;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fr.lo) : "v"(addr));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(fr.hi) : "v"(addr));
    }
}

template <int F>
__device__ __forceinline__ void step(Frag (&ring)[AHEAD], f32x16 (&acc)[8], const bf16x8 (&b)[8], uint32_t kbase, uint32_t vbase,
                                     uint32_t flip_now, uint32_t flip_next) {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(behind(F)) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    Frag& fr = ring[F % AHEAD];
    bf16x8 a;
    __builtin_memcpy(&a, &fr, 16);
    acc[(2 * F) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[F & 7], acc[(2 * F) & 7], 0, 0, 0);
    acc[(2 * F + 1) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[(F + 3) & 7], acc[(2 * F + 1) & 7], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    request<F + AHEAD>(fr, kbase, vbase, F + AHEAD >= 32 ? flip_next : flip_now);      // (the slot just consumed)
    if constexpr (F + 1 < 32) step<F + 1>(ring, acc, b, kbase, vbase, flip_now, flip_next);
}

template <int F>
__device__ __forceinline__ void prime(Frag (&ring)[AHEAD], uint32_t kbase, uint32_t vbase) {
    request<F>(ring[F], kbase, vbase, 0);
    if constexpr (F + 1 < AHEAD) prime<F + 1>(ring, kbase, vbase);
}

__global__ __launch_bounds__(256, 1) void pfa_probe_mfma_kernel(const uint4* __restrict__ rnd, float* __restrict__ sink, int iters) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < PROBE_LDS / 16; i += 256) lds[i] = rnd[i];
    __syncthreads();
    // B operands: 8 random fragments per wave (the forward keeps Q / P in registers)
    bf16x8 b[8];
    for (int i = 0; i < 8; ++i) {
        const uint4 w = rnd[(tid * 8 + i) & (PROBE_LDS / 16 - 1)];
        __builtin_memcpy(&b[i], &w, 16);
    }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    // conflict-free images: a lane's 16-byte chunk index XOR-ed with its row bits, as in the forward's tile image
    const int r = lane & 31, h = lane >> 5;
    const uint32_t kbase = (uint32_t)(r * 256 + ((h ^ ((r & 3) << 2 | ((r >> 2) & 3))) << 4));
    const uint32_t vbase = (uint32_t)(32 * 1024 + ((lane & 15) >> 2) * 256 + (lane >> 4) * 1024 + ((lane & 3) << 3));
    Frag ring[AHEAD];
    prime<0>(ring, kbase, vbase);
    for (int it = 0; it < iters; ++it) {
        const uint32_t flip_now = (uint32_t)(it & 1) * 16384u, flip_next = (uint32_t)((it + 1) & 1) * 16384u;
        step<0>(ring, acc, b, kbase, vbase, flip_now, flip_next);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    sink[blockIdx.x * 256 + tid] = s;
}
}  // namespace

extern "C" int pfa_probe_mfma(const void* random_64k, float* sink, int iters, int device_id, void* stream, double* flops) {
    if (!random_64k || !sink || iters <= 0) return PFA_ERR_NULL;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { (void)hipGetLastError(); return PFA_ERR_DEVICE; }
    const int n_wg = prop.multiProcessorCount;
    const void* fn = (const void*)&pfa_probe_mfma_kernel;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PROBE_WG_LDS);
    const uint4* rnd = (const uint4*)random_64k;
    void* kargs[] = {&rnd, &sink, &iters};
    if (hipLaunchKernel(fn, dim3((unsigned)n_wg), dim3(256), kargs, (size_t)PROBE_WG_LDS, (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        return PFA_ERR_LAUNCH;
    }
    if (flops) *flops = (double)n_wg * 4.0 * (double)iters * 64.0 * 32768.0;
    return n_wg;
}

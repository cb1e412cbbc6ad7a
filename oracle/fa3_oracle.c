/*
 * ORACLE (test infrastructure, never shipped, never linked by the product):
 * plain-C fp32 restatement of the reference's electronic attention core,
 * /root/reference/src/photonic_flash_attention/core/flash_attention_3.py
 *
 *   oracle_fa3_fwd_f32  <-  _flash_attention_forward (:120-150)
 *                            dense branch  _standard_attention (:152-180)  when Sq,Sk <= tile
 *                            tiled branch  _tiled_attention    (:182-262)  otherwise
 *
 * It exists beside oracle/fa3_oracle.py (same algorithm on torch CPU ops) as an independent
 * second statement with no BLAS/softmax library underneath: tests/test_oracle_golden.py checks
 * BOTH against the golden vectors produced by the real reference (parity: pinned).
 *
 * Layout: q [B,Sq,H,D], k/v [B,Sk,H,D], out [B,Sq,H,D], contiguous fp32.
 * Mask: causal flag (the reference's 4-D lower-triangular mask, key j visible to row i iff
 * j <= i) and/or per-batch valid key counts (a 2-D key-padding mask); NULL = none.
 * One OpenMP task per (batch, head, q-tile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int optimal_tile(int sq, int sk) {
    /* :264-293 with a >= 8 GB budget: the memory bound never binds -> min(Sq,Sk,512), floor 32 */
    int t = sq < sk ? sq : sk;
    if (t > 512) t = 512;
    if (t < 32) t = 32;
    return t;
}

int oracle_fa3_tile(int sq, int sk) { return optimal_tile(sq, sk); }

/* one q-tile x all kv-tiles of one (b,h); rows [i0,i1) */
static void one_q_tile(const float* q, const float* k, const float* v, float* out, int H, int Sq, int Sk, int D,
                       int h, int i0, int i1, int tile, float scaling, int causal, int kv_valid, int dense) {
    const int nq = i1 - i0;
    const size_t rs = (size_t)H * D; /* row stride */
    float* qs = (float*)malloc((size_t)nq * D * sizeof(float));
    float* o = (float*)calloc((size_t)nq * D, sizeof(float));
    float* m = (float*)malloc(nq * sizeof(float));
    float* l = (float*)calloc(nq, sizeof(float));
    float* s = (float*)malloc((size_t)tile * sizeof(float));
    float* pv = (float*)malloc((size_t)D * sizeof(float));
    for (int i = 0; i < nq; ++i) {
        m[i] = -INFINITY;
        for (int d = 0; d < D; ++d) qs[(size_t)i * D + d] = q[(size_t)(i0 + i) * rs + (size_t)h * D + d] * scaling; /* :138 */
    }
    const int step = dense ? Sk : tile;
    for (int j0 = 0; j0 < Sk; j0 += step) {
        const int j1 = j0 + step < Sk ? j0 + step : Sk;
        const int nk = j1 - j0;
        for (int i = 0; i < nq; ++i) {
            const float* qi = qs + (size_t)i * D;
            float mx = -INFINITY;
            for (int j = 0; j < nk; ++j) {
                const float* kj = k + (size_t)(j0 + j) * rs + (size_t)h * D;
                float acc = 0.f;
                for (int d = 0; d < D; ++d) acc += qi[d] * kj[d];
                const int key = j0 + j;
                if ((causal && key > i0 + i) || key >= kv_valid) acc = -INFINITY; /* :168 / :236 */
                s[j] = acc;
                if (acc > mx) mx = acc;
            }
            const float m_new = m[i] > mx ? m[i] : mx;                      /* :239-240 */
            if (m_new == -INFINITY) continue;                               /* nothing visible yet */
            const float e_old = expf(m[i] - m_new) * l[i];                  /* :244 */
            float psum = 0.f;
            memset(pv, 0, (size_t)D * sizeof(float));
            for (int j = 0; j < nk; ++j) {
                const float p = expf(s[j] - m_new);                         /* :243 */
                psum += p;
                if (p != 0.f) {
                    const float* vj = v + (size_t)(j0 + j) * rs + (size_t)h * D;
                    for (int d = 0; d < D; ++d) pv[d] += p * vj[d];
                }
            }
            const float l_new = e_old + psum;                               /* :246 */
            float* oi = o + (size_t)i * D;
            for (int d = 0; d < D; ++d) oi[d] = (e_old * oi[d] + pv[d]) / l_new; /* :250 */
            m[i] = m_new;
            l[i] = l_new;
        }
    }
    for (int i = 0; i < nq; ++i)
        memcpy(out + (size_t)(i0 + i) * rs + (size_t)h * D, o + (size_t)i * D, (size_t)D * sizeof(float));
    free(qs); free(o); free(m); free(l); free(s); free(pv);
}

/* returns 0; seqlens_k may be NULL */
int oracle_fa3_fwd_f32(const float* q, const float* k, const float* v, float* out, int B, int H, int Sq, int Sk,
                       int D, float scaling, int causal, const int32_t* seqlens_k) {
    const int tile = optimal_tile(Sq, Sk);
    const int dense = (Sq <= tile && Sk <= tile);
    const int qstep = dense ? Sq : tile;
    const int nqt = (Sq + qstep - 1) / qstep;
    const long total = (long)B * H * nqt;
#pragma omp parallel for schedule(dynamic, 1)
    for (long w = 0; w < total; ++w) {
        const int qt = (int)(w % nqt);
        const int h = (int)((w / nqt) % H);
        const int b = (int)(w / ((long)nqt * H));
        const int i0 = qt * qstep;
        const int i1 = i0 + qstep < Sq ? i0 + qstep : Sq;
        int kvv = Sk;
        if (seqlens_k && seqlens_k[b] < kvv) kvv = seqlens_k[b] < 0 ? 0 : seqlens_k[b];
        one_q_tile(q + (size_t)b * Sq * H * D, k + (size_t)b * Sk * H * D, v + (size_t)b * Sk * H * D,
                   out + (size_t)b * Sq * H * D, H, Sq, Sk, D, h, i0, i1, tile, scaling, causal, kvv, dense);
    }
    return 0;
}

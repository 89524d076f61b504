/* Oracle: exact flat inner-product search in plain C.  TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of what the reference asks of faiss at test_amazon_filterd.py:578
 * (IndexFlatIP.search; semantics SURVEY.md Appendix A.5) under the canonical contract of
 * score(a,b) = sum_{k=0..d-1} q[a][k]*c[b][k] accumulated sequentially in double
 * (each float*float product is exact in double) and rounded once to float; results ordered by
 * (score descending, id ascending); missing results id -1, score -FLT_MAX.
 * PARITY UNPINNED: faiss is not installed and the reference holds no fixture for this path.
 *
 * Built by oracle/Makefile into oracle/libsss_oracle.so; loaded by oracle/search_ref.py.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* insert (s,id) into a list sorted by (score desc, id asc); ids arrive ascending per query,
 * so a strict '>' keeps equal scores in id order. */
static inline void insert_sorted(float* D, int64_t* I, int k, float s, int64_t id) {
    if (I[k - 1] >= 0 && !(s > D[k - 1])) return; /* full list and not better than its tail */
    int p = k - 1;                                /* empty slots (id -1) lose to any entry */
    while (p > 0 && (I[p - 1] < 0 || s > D[p - 1])) {
        D[p] = D[p - 1];
        I[p] = I[p - 1];
        --p;
    }
    D[p] = s;
    I[p] = id;
}

static int search_impl(const float* q, int64_t nq, const float* c, int64_t n, int d, int k,
                       int64_t id_offset, float* D, int64_t* I, int threads, int mode) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t a = 0; a < nq; ++a) {
        float* Da = D + a * k;
        int64_t* Ia = I + a * k;
        for (int j = 0; j < k; ++j) { Da[j] = -FLT_MAX; Ia[j] = -1; }
        const float* qa = q + a * d;
        for (int64_t b = 0; b < n; ++b) {
            const float* cb = c + b * d;
            float s;
            if (mode == 0) {
                double acc = 0.0;
                for (int t = 0; t < d; ++t) acc += (double)qa[t] * (double)cb[t];
                s = (float)acc;
            } else {
                float acc = 0.0f; /* k-ordered float fma chain == gfx950 f32 MFMA numerics */
                for (int t = 0; t < d; ++t) acc = fmaf(qa[t], cb[t], acc);
                s = acc;
            }
            insert_sorted(Da, Ia, k, s, b + id_offset);
        }
    }
    return 0;
}

int oracle_search_exact(const float* q, int64_t nq, const float* c, int64_t n, int d, int k,
                        int64_t id_offset, float* D, int64_t* I, int threads) {
    return search_impl(q, nq, c, n, d, k, id_offset, D, I, threads, 0);
}

int oracle_search_fp32(const float* q, int64_t nq, const float* c, int64_t n, int d, int k,
                       int64_t id_offset, float* D, int64_t* I, int threads) {
    return search_impl(q, nq, c, n, d, k, id_offset, D, I, threads, 1);
}

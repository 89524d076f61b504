// Exhaustive exact search: canonical float64 scores of selected queries against EVERY corpus
// row, then an exact top-k.  This is the correctness backstop behind k_ip_topk_f32 (queries
// whose fused result could not be proven exact, tiny corpora, k larger than the fused path,
// the IndexFlatL2 metric of reference test_amazon_filterd.py:215-217, any d % 4 == 0 such as
// the reference's D = 1600).  It favours simplicity over speed; DESIGN.md "exactness".
#include "scan.h"

namespace sss {

constexpr int EX_ROWS = 64;   // rows per wave pass
constexpr int EX_KC = 128;    // k-chunk staged in LDS
constexpr int EX_LD = EX_KC + 4;

// scores[f][row] = float32( sum_k q[qsel[f]][k] * c[row][k] ) accumulated sequentially in
// float64 (metric 0), or sum_k (q_k - c_k)^2 with one rounding per multiply and per add (1).
// DT_BF16: q and c hold bf16; every element converts exactly to float32 on the way into LDS.
template <int DT>
__global__ __launch_bounds__(64) void k_exact_scores(const void* __restrict__ Qv,
                                                     const int* __restrict__ qsel,
                                                     const void* __restrict__ Cv, long n, int d,
                                                     int metric, float* __restrict__ scores) {
    __shared__ __attribute__((aligned(16))) float tile[EX_ROWS * EX_LD];
    __shared__ __attribute__((aligned(16))) float qs[EX_KC];
    constexpr int EB = DT == DT_F32 ? 4 : 2;          // bytes per element
    constexpr int EPV = 16 / EB;                      // elements per 16-byte load
    const int lane = threadIdx.x;
    const int f = blockIdx.y;
    const char* C = reinterpret_cast<const char*>(Cv);
    const char* q = reinterpret_cast<const char*>(Qv) + (size_t)qsel[f] * d * EB;
    auto to_f32 = [](const char* p, int i) -> float {
        if (DT == DT_F32) return reinterpret_cast<const float*>(p)[i];
        return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(p)[i] << 16);
    };
    for (long row0 = (long)blockIdx.x * EX_ROWS; row0 < n; row0 += (long)gridDim.x * EX_ROWS) {
        double acc = 0.0;
        for (int k0 = 0; k0 < d; k0 += EX_KC) {
            const int kc = d - k0 < EX_KC ? d - k0 : EX_KC;   // multiple of EPV
            const int nv = kc / EPV;
            __syncthreads();
            for (int i = lane; i < EX_ROWS * nv; i += 64) {
                const int rr = i / nv, cc = i % nv;
                long grow = row0 + rr;
                if (grow > n - 1) grow = n - 1;
                const f32x4 v = *reinterpret_cast<const f32x4*>(C + ((size_t)grow * d + k0) * EB + cc * 16);
#pragma unroll
                for (int e = 0; e < EPV; ++e) tile[rr * EX_LD + cc * EPV + e] = to_f32(reinterpret_cast<const char*>(&v), e);
            }
            for (int i = lane; i < kc; i += 64) qs[i] = to_f32(q, k0 + i);
            __syncthreads();
            const float* mine = &tile[lane * EX_LD];
            if (metric == 0) {
                for (int kk = 0; kk < kc; ++kk) acc += (double)qs[kk] * (double)mine[kk];
            } else {
                for (int kk = 0; kk < kc; ++kk) {
                    const double dl = __dsub_rn((double)qs[kk], (double)mine[kk]);
                    acc = __dadd_rn(acc, __dmul_rn(dl, dl));
                }
            }
        }
        if (row0 + lane < n) scores[(size_t)f * n + row0 + lane] = (float)acc;
    }
}

// One block per selected query: k rounds of "largest key strictly below the previous one"
// over the n scores (keys are unique because ids are).  O(k*n) reads -- backstop only.
constexpr int FULL_THREADS = 1024;
__global__ __launch_bounds__(FULL_THREADS) void k_topk_full(const float* __restrict__ scores,
                                                            const int* __restrict__ qsel, long n,
                                                            int k, long id_offset, int metric,
                                                            float* __restrict__ D_out,
                                                            long* __restrict__ I_out) {
    __shared__ unsigned long long wbest[FULL_THREADS / 64];
    __shared__ unsigned long long s_prev;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* s = scores + (size_t)f * n;
    const size_t out = (size_t)qsel[f] * k;
    unsigned long long prev = ~0ull;
    for (int it = 0; it < k; ++it) {
        unsigned long long best = 0;
        for (long i = tid; i < n; i += FULL_THREADS) {
            const float v = metric == 0 ? s[i] : -s[i];
            const unsigned long long key = make_key(v, (int)i);
            if (key < prev && key > best) best = key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o);
            best = other > best ? other : best;
        }
        if (lane == 0) wbest[wv] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long g = 0;
            for (int w = 0; w < FULL_THREADS / 64; ++w) g = wbest[w] > g ? wbest[w] : g;
            s_prev = g;
            if (g == 0) {
                D_out[out + it] = metric == 0 ? -3.4028234663852886e38f : 3.4028234663852886e38f;
                I_out[out + it] = -1;
            } else {
                const float v = key_score(g);
                D_out[out + it] = metric == 0 ? v : -v;
                I_out[out + it] = (long)key_id(g) + id_offset;
            }
        }
        __syncthreads();
        prev = s_prev;
        if (prev == 0) prev = 0;   // exhausted: every later round also yields 0 -> padding
    }
}

size_t ip_topk_exhaustive_workspace_bytes(long nsel, long n) { return (size_t)nsel * n * 4 + 256; }

int ip_topk_exhaustive(const void* q, const int* qsel, long nsel, const void* c, long n, int d,
                       int k, int dtype, long id_offset, int metric, float* D_out, long* I_out, void* ws,
                       size_t ws_bytes, hipStream_t st) {
    if (nsel <= 0 || n <= 0 || k <= 0 || d <= 0 || (dtype != DT_F32 && dtype != DT_BF16) ||
        d % (dtype == DT_F32 ? 4 : 8) || (metric != 0 && metric != 1)) {
        set_error("ip_topk_exhaustive: need nsel, n, k > 0, d %% 4 == 0 (f32) / d %% 8 == 0 (bf16), metric in {0,1}");
        return SSS_EINVAL;
    }
    if (n >= (1L << 31) || nsel > 65535) { set_error("ip_topk_exhaustive: n < 2^31, nsel <= 65535"); return SSS_EINVAL; }
    if (ws_bytes < ip_topk_exhaustive_workspace_bytes(nsel, n)) {
        set_error("ip_topk_exhaustive: workspace too small");
        return SSS_EWORKSPACE;
    }
    float* scores = reinterpret_cast<float*>(ws);
    long gx = (n + EX_ROWS - 1) / EX_ROWS;
    if (gx > 8192) gx = 8192;
    if (dtype == DT_F32)
        hipLaunchKernelGGL(k_exact_scores<DT_F32>, dim3((unsigned)gx, (unsigned)nsel), dim3(64), 0, st, q, qsel, c, n, d, metric, scores);
    else
        hipLaunchKernelGGL(k_exact_scores<DT_BF16>, dim3((unsigned)gx, (unsigned)nsel), dim3(64), 0, st, q, qsel, c, n, d, metric, scores);
    int rc = check_launch("k_exact_scores");
    if (rc) return rc;
    hipLaunchKernelGGL(k_topk_full, dim3((unsigned)nsel), dim3(FULL_THREADS), 0, st, scores, qsel, n, k, id_offset, metric, D_out, I_out);
    return check_launch("k_topk_full");
}

}  // namespace sss

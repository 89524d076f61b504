// Exhaustive exact search: canonical float64 scores of selected queries against EVERY corpus
// row, then an exact top-k.  This is the correctness backstop behind the fused scans (queries
// whose fused result could not be proven exact, tiny corpora, k larger than the fused path,
// the IndexFlatL2 metric of reference test_amazon_filterd.py:215-217, any d % 4 == 0 such as
// the reference's D = 1600).  O(n d) scoring + O(n) radix selection per query (long rows: after a
// multi-workgroup compaction of the rows that can still matter); DESIGN.md "exactness".
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

// Fast scorer for the fused kernel's shapes (d = 64 / 128 / 256 elements per row): a lane keeps
// ONE corpus row in registers and walks all selected queries, so the corpus is read once however
// many queries need the backstop (k_exact_scores above re-reads it per query).  The query elements are
// wave-uniform: they come in through the scalar cache (s_load) and enter the FMAs as SGPR operands --
// no LDS, no barrier.  Same canonical score: sequential float64 sum.
// lb (optional, one float per selected query): a LOWER bound of the query's k-th best score (the fused
// path's k-th re-scored candidate).  A cheap float32 pass first bounds every row's score from above
// (float32 fma chain: error <= d 2^-24 |q||c|); rows that provably stay below lb -- almost all of them --
// skip the float64 chain and get -FLT_MAX, which the selection ignores.
template <int D, int DT>
__device__ __forceinline__ float q_elem(const void* qrow, int kx) {        // element kx of a (wave-uniform) query row
    if (DT == DT_F32) return reinterpret_cast<const float*>(qrow)[kx];
    const unsigned w = reinterpret_cast<const unsigned*>(qrow)[kx >> 1];
    return __builtin_bit_cast(float, (kx & 1) ? (w & 0xFFFF0000u) : (w << 16));
}

template <int D, int DT>
__global__ __launch_bounds__(256) void k_exact_scores_rows(const void* __restrict__ Qv, const int* __restrict__ qsel, int nsel,
                                                           const void* __restrict__ Cv, long n, float* __restrict__ scores,
                                                           const float* __restrict__ lb) {
    constexpr int QB = 1024;                           // queries whose norms are kept in LDS at a time
    __shared__ float qn[QB];
    constexpr int EB = DT == DT_F32 ? 4 : 2;
    const char* C = reinterpret_cast<const char*>(Cv);
    const char* Q = reinterpret_cast<const char*>(Qv);
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    const long rl = row < n ? row : n - 1;
    float r[D];
    if (DT == DT_F32) {
#pragma unroll
        for (int v = 0; v < D / 4; ++v) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(C + ((size_t)rl * D + v * 4) * EB);
            r[4 * v] = c4.x; r[4 * v + 1] = c4.y; r[4 * v + 2] = c4.z; r[4 * v + 3] = c4.w;
        }
    } else {
#pragma unroll
        for (int v = 0; v < D / 8; ++v) {
            const uint4 c4 = *reinterpret_cast<const uint4*>(C + ((size_t)rl * D + v * 8) * EB);
            const unsigned u[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r[8 * v + 2 * e] = __builtin_bit_cast(float, u[e] << 16);
                r[8 * v + 2 * e + 1] = __builtin_bit_cast(float, u[e] & 0xFFFF0000u);
            }
        }
    }
    float rnorm = 0.f;
    if (lb) {
#pragma unroll
        for (int kx = 0; kx < D; ++kx) rnorm = fmaf(r[kx], r[kx], rnorm);
        rnorm = sqrtf(rnorm) * 1.0001f;
    }
    for (int f0 = 0; f0 < nsel; f0 += QB) {
        const int nf = nsel - f0 < QB ? nsel - f0 : QB;
        if (lb) {                                          // query norms of this block (for the pre-test's error margin)
            __syncthreads();
            for (int t = threadIdx.x; t < nf; t += 256) {
                const char* qrow = Q + (size_t)qsel[f0 + t] * D * EB;
                float s2 = 0.f;
                for (int kx = 0; kx < D; ++kx) { const float v = q_elem<D, DT>(qrow, kx); s2 = fmaf(v, v, s2); }
                qn[t] = sqrtf(s2) * 1.0001f;
            }
            __syncthreads();
        }
        for (int f = 0; f < nf; ++f) {
            const char* qrow = Q + (size_t)qsel[f0 + f] * D * EB;      // wave-uniform address: scalar loads
            bool need = true;
            if (lb) {                                      // float32 upper bound first
                float s0 = 0.f;
#pragma unroll
                for (int kx = 0; kx < D; ++kx) s0 = fmaf(q_elem<D, DT>(qrow, kx), r[kx], s0);
                const float l0 = lb[f0 + f];
                need = !(s0 + (float)D * 6.3e-8f * rnorm * qn[f] + 2.4e-7f * fabsf(l0) < l0);   // (NaN anywhere: keep the row)
            }
            double acc = 0.0;
            if (need) {
#pragma unroll
                for (int kx = 0; kx < D; ++kx) acc += (double)q_elem<D, DT>(qrow, kx) * (double)r[kx];
            }
            if (row < n) scores[(size_t)(f0 + f) * n + row] = need ? (float)acc : -3.4028234663852886e38f;
        }
    }
}

// One block per selected query: exact top-k of its n canonical scores in O(n):
//   1. radix select (4 passes of 8 bits over the order-preserving uint of the score, LDS
//      histograms) -> T = the k-th best score and how many rows tied at T are still needed;
//   2. collect every row better than T (unordered) and, scanning ids in increasing order, the
//      lowest-id rows equal to T;
//   3. bitonic sort of the k collected (score desc, id asc) keys, write.
// Correct for any amount of ties (mass duplicates, identical rows).
constexpr int RS_THREADS = 1024;
constexpr int RS_MAX_K = 1024;
// With a compacted input (cs != nullptr and the query's survivors fit `cap`): the select runs over the
// query's `ctotal` survivors (scores cs, row ids cids, in increasing id order) instead of all n rows.
__global__ __launch_bounds__(RS_THREADS) void k_topk_radix(const float* __restrict__ scores,
                                                           const int* __restrict__ qsel, long n_rows, int k,
                                                           long id_offset, int metric,
                                                           float* __restrict__ D_out, long* __restrict__ I_out,
                                                           const float* __restrict__ cs, const int* __restrict__ cids,
                                                           const unsigned* __restrict__ ctotal, int cap) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long keys[RS_MAX_K];
    __shared__ unsigned s_prefix, s_need, s_count, s_wave[RS_THREADS / 64], s_taken;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* s = scores + (size_t)f * n_rows;
    const int* ids = nullptr;
    long n = n_rows;
    if (cs && ctotal[f] <= (unsigned)cap) { s = cs + (size_t)f * cap; ids = cids + (size_t)f * cap; n = (long)ctotal[f]; }
    const size_t out = (size_t)qsel[f] * k;
    auto key_of = [&](long i) { return f2ord(metric == 0 ? s[i] : -s[i]); };
    auto id_of = [&](long i) { return ids ? (unsigned)ids[i] : (unsigned)i; };
    const int kk = (long)k < n_rows ? k : (int)n_rows;  // rows actually returned (a compacted input holds at least kk)
    int K2 = 64;
    while (K2 < kk) K2 <<= 1;
    for (int i = tid; i < K2; i += RS_THREADS) keys[i] = 0ull;
    unsigned T = 0, need = kk;
    if (kk < n) {                                        // (kk == n: every row is returned, no selection needed)
        unsigned prefix = 0, mask = 0;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            for (int i = tid; i < 256; i += RS_THREADS) hist[i] = 0;
            __syncthreads();
            // four independent loads in flight per thread (the passes are latency-bound otherwise);
            // whole waves stay in the loop so the ballots below are convergent
            const long n_up = (n + 4 * RS_THREADS - 1) / (4 * RS_THREADS) * (4 * RS_THREADS);
            for (long i0 = tid; i0 < n_up; i0 += 4 * RS_THREADS) {
                unsigned keyv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long i = i0 + (long)u * RS_THREADS;
                    keyv[u] = i < n ? key_of(i) : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long i = i0 + (long)u * RS_THREADS;
                    const unsigned key = keyv[u];
                    const bool act = i < n && (key & mask) == prefix;
                    const unsigned b = (key >> shift) & 255u;
                    if (pass == 0) {
                        // sign and exponent bits are shared by most scores: one LDS atomic per
                        // DISTINCT bucket of the wave instead of 64 serialised ones on the same address
                        unsigned long long todo = __builtin_amdgcn_ballot_w64(act);
                        while (todo) {
                            const int leader = __builtin_ctzll(todo);
                            const unsigned lb = (unsigned)__builtin_amdgcn_readlane((int)b, leader);
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(act && b == lb);
                            if (lane == leader) atomicAdd(&hist[lb], (unsigned)__builtin_popcountll(m));
                            todo &= ~m;
                        }
                    } else if (act) {
                        atomicAdd(&hist[b], 1u);
                    }
                }
            }
            __syncthreads();
            if (tid == 0) {
                unsigned cum = 0;
                int b = 255;
                for (; b > 0; --b) {                     // best bucket first
                    if (cum + hist[b] >= need) break;
                    cum += hist[b];
                }
                s_prefix = prefix | ((unsigned)b << shift);
                s_need = need - cum;                     // rank still to find inside the chosen bucket
            }
            __syncthreads();
            prefix = s_prefix; need = s_need;
            mask |= 255u << shift;
            __syncthreads();
        }
        T = prefix;                                      // the kk-th best key; `need` rows equal to it are wanted
    }
    if (tid == 0) { s_count = 0; s_taken = 0; }
    __syncthreads();
    // ---- rows strictly better than T (fewer than kk of them), in any order
    if (kk < n) {
        for (long i0 = tid; i0 < n; i0 += 4 * RS_THREADS) {
            unsigned keyv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long i = i0 + (long)u * RS_THREADS;
                keyv[u] = i < n ? key_of(i) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long i = i0 + (long)u * RS_THREADS;
                if (i < n && keyv[u] > T) {
                    const unsigned p = atomicAdd(&s_count, 1u);
                    keys[p] = ((unsigned long long)keyv[u] << 32) | (unsigned)(~id_of(i));
                }
            }
        }
    }
    __syncthreads();
    // ---- rows equal to T (every row when kk == n), lowest ids first: chunks in id order + block prefix sum
    const unsigned want = kk < n ? need : (unsigned)kk;
    for (long base = 0; base < n; base += RS_THREADS) {
        if (s_taken >= want) break;                      // uniform: s_taken only changes between barriers
        const long i = base + tid;
        unsigned key = 0;
        bool hit = false;
        if (i < n) { key = key_of(i); hit = kk < n ? key == T : true; }
        const unsigned long long b = __builtin_amdgcn_ballot_w64(hit);
        if (lane == 0) s_wave[wv] = (unsigned)__builtin_popcountll(b);
        __syncthreads();
        unsigned before = 0;
        for (int w = 0; w < wv; ++w) before += s_wave[w];
        const unsigned rank = s_taken + before + (unsigned)__builtin_popcountll(b & ((1ull << lane) - 1ull));
        if (hit && rank < want) keys[s_count + rank] = ((unsigned long long)key << 32) | (unsigned)(~id_of(i));
        __syncthreads();
        if (tid == 0) {
            unsigned tot = 0;
            for (int w = 0; w < RS_THREADS / 64; ++w) tot += s_wave[w];
            s_taken += tot;
        }
        __syncthreads();
    }
    __syncthreads();
    // ---- order by (score desc, id asc) == key descending; padding keys are 0
    for (int kq = 2; kq <= K2; kq <<= 1) {
        for (int j = kq >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < K2; i += RS_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], bq = keys[ixj];
                    const bool desc = (i & kq) == 0;
                    if (desc ? a < bq : a > bq) { keys[i] = bq; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += RS_THREADS) {
        if (i < kk) {
            const float v = key_score(keys[i]);
            D_out[out + i] = metric == 0 ? v : -v;
            I_out[out + i] = (long)key_id(keys[i]) + id_offset;
        } else {                                         // faiss pads missing results
            D_out[out + i] = metric == 0 ? -3.4028234663852886e38f : 3.4028234663852886e38f;
            I_out[out + i] = -1;
        }
    }
}


// ---- compaction pre-pass for long rows (n >= CP_MIN_N): the single-workgroup select above is bound by
// its own load latency (6 sweeps over n).  The kk-th best key of CP_HEAD evenly spread scores is a lower
// bound T0 of the true kk-th best, so every row of the answer (ties included) has key >= T0; those
// rows are compacted IN ID ORDER by many workgroups (count per slab, scan, ordered write) and the select
// then runs over the survivors.  More survivors than CP_CAP (rows sorted ascending, say): the select
// falls back to the full row -- speed only, never correctness.
constexpr int CP_HEAD = 32768;        // sampled scores (their ordinals are staged in 128 KB of LDS: one fetch, four radix passes)
constexpr int CP_SLAB = 65536;
constexpr int CP_CAP = 131072;
constexpr long CP_MIN_N = 262144;

__global__ __launch_bounds__(RS_THREADS) void k_head_threshold(const float* __restrict__ scores, long n, int k, int metric,
                                                               unsigned* __restrict__ T0) {
    extern __shared__ __attribute__((aligned(16))) unsigned ords[];     // [head] ordinals of the sampled scores
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_need;
    const int f = blockIdx.x, tid = threadIdx.x;
    const float* s = scores + (size_t)f * n;
    const long head = n < CP_HEAD ? n : CP_HEAD;         // sample size; rows i * stride: an evenly spread sample
    const long stride = n / head;                        // (corpora are often ordered -- by session length, by prefix ...)
    // The sample is fetched ONCE (every element its own cache line: with the four passes reading global memory this single
    // workgroup spent 0.40 ms per query on 262 k scattered loads at 4M rows).  k <= 1024 (always, on the paths that reach
    // this kernel today): a thread keeps only the MAXIMUM of its 32 samples -- 1024 distinct rows, so their k-th largest is
    // still a value that k rows reach -- and the radix passes run over 1024 words instead of 32768 (the first pass of the
    // full form put all 32768 LDS atomics on two or three bins: 0.23 ms per query even from LDS).  The bound is a little
    // looser (about the top 2 % of the corpus survive the compaction instead of 1.5 %).
    unsigned need = (unsigned)((long)k < head ? k : head), prefix = 0, mask = 0;
    long cnt = head;
    if (need <= RS_THREADS && head >= RS_THREADS) {
        unsigned best = 0u;
        for (long i = tid; i < head; i += RS_THREADS) {
            const float v = s[i * stride];
            best = max(best, f2ord(metric == 0 ? v : -v));
        }
        ords[tid] = best;
        cnt = RS_THREADS;
    } else {
        for (long i = tid; i < head; i += RS_THREADS) {
            const float v = s[i * stride];
            ords[i] = f2ord(metric == 0 ? v : -v);
        }
    }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 256; i += RS_THREADS) hist[i] = 0;
        __syncthreads();
        for (long i = tid; i < cnt; i += RS_THREADS) {
            const unsigned key = ords[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {                                  // the bin holding the need-th key from the top: four bins a lane, suffix sum by shuffles
            const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const unsigned mine = h0 + h1 + h2 + h3;
            unsigned suf = mine;                         // sum over this lane and every higher one
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = (unsigned)__shfl_down((int)suf, o);
                if (tid + o < 64) suf += v;
            }
            const unsigned above = suf - mine;
            if (above < need && suf >= need) {           // (exactly one lane)
                unsigned cum = above;
                int b2 = 3;
                if (cum + h3 < need) { cum += h3; b2 = 2; if (cum + h2 < need) { cum += h2; b2 = 1; if (cum + h1 < need) { cum += h1; b2 = 0; } } }
                s_prefix = prefix | ((unsigned)(4 * tid + b2) << shift);
                s_need = need - cum;
            }
        }
        __syncthreads();
        prefix = s_prefix; need = s_need;
        mask |= 255u << shift;
        __syncthreads();
    }
    if (tid == 0) T0[f] = prefix;
}

// grid (slabs, nsel): rows of one slab of one query with key > T0 and with key == T0.  Of the rows EQUAL to
// T0 only the kk lowest ids can ever be needed (the true kk-th best is >= T0; if it is T0 the answer takes
// the lowest ids among its ties), so a boundary inside a huge group of identical rows still compacts.
__global__ __launch_bounds__(256) void k_count_ge(const float* __restrict__ scores, long n, int metric,
                                                  const unsigned* __restrict__ T0, int nslabs, unsigned* __restrict__ cnt_gt,
                                                  unsigned* __restrict__ cnt_eq) {
    __shared__ unsigned s_g[4], s_e[4];
    const int f = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
    const float* s = scores + (size_t)f * n;
    const unsigned t0 = T0[f];
    const long lo = (long)slab * CP_SLAB, hi = lo + CP_SLAB < n ? lo + CP_SLAB : n;
    unsigned g = 0, e = 0;
    for (long i = lo + tid; i < hi; i += 256) {
        const unsigned key = f2ord(metric == 0 ? s[i] : -s[i]);
        g += key > t0 ? 1u : 0u;
        e += key == t0 ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { g += __shfl_xor(g, o); e += __shfl_xor(e, o); }
    if ((tid & 63) == 0) { s_g[tid >> 6] = g; s_e[tid >> 6] = e; }
    __syncthreads();
    if (tid == 0) {
        cnt_gt[(size_t)f * nslabs + slab] = s_g[0] + s_g[1] + s_g[2] + s_g[3];
        cnt_eq[(size_t)f * nslabs + slab] = s_e[0] + s_e[1] + s_e[2] + s_e[3];
    }
}

// one thread block per query: exclusive scans of its slab counts (in place) + the number of rows kept
__global__ __launch_bounds__(64) void k_scan_slabs(unsigned* __restrict__ cnt_gt, unsigned* __restrict__ cnt_eq, int nslabs,
                                                   long n, int k, unsigned* __restrict__ total) {
    const int f = blockIdx.x;
    if (threadIdx.x != 0) return;
    const unsigned kk = (unsigned)((long)k < n ? k : n);
    unsigned rg = 0, re = 0;
    for (int j = 0; j < nslabs; ++j) {
        const unsigned g = cnt_gt[(size_t)f * nslabs + j], e = cnt_eq[(size_t)f * nslabs + j];
        cnt_gt[(size_t)f * nslabs + j] = rg;
        cnt_eq[(size_t)f * nslabs + j] = re;
        rg += g; re += e;
    }
    total[f] = rg + (re < kk ? re : kk);
}

// grid (slabs, nsel): ordered write of the slab's kept rows (key > T0, or key == T0 among the first kk such
// rows of the query) at position (#greater before) + min(#equal before, kk): ids ascending.
__global__ __launch_bounds__(RS_THREADS) void k_compact_ge(const float* __restrict__ scores, long n, int k, int metric,
                                                           const unsigned* __restrict__ T0, int nslabs,
                                                           const unsigned* __restrict__ off_gt, const unsigned* __restrict__ off_eq,
                                                           const unsigned* __restrict__ total, int cap, float* __restrict__ cs,
                                                           int* __restrict__ cids) {
    __shared__ unsigned s_wg[RS_THREADS / 64], s_we[RS_THREADS / 64];
    __shared__ unsigned s_bg, s_be;
    const int f = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (total[f] > (unsigned)cap) return;                 // does not fit: the select reads the full row
    const float* s = scores + (size_t)f * n;
    const unsigned t0 = T0[f];
    const unsigned kk = (unsigned)((long)k < n ? k : n);
    const long lo = (long)slab * CP_SLAB, hi = lo + CP_SLAB < n ? lo + CP_SLAB : n;
    if (tid == 0) { s_bg = off_gt[(size_t)f * nslabs + slab]; s_be = off_eq[(size_t)f * nslabs + slab]; }
    __syncthreads();
    for (long base = lo; base < hi; base += RS_THREADS) {
        const long i = base + tid;
        float v = 0.f;
        bool gt = false, eq = false;
        if (i < hi) {
            v = s[i];
            const unsigned key = f2ord(metric == 0 ? v : -v);
            gt = key > t0; eq = key == t0;
        }
        const unsigned long long bg = __builtin_amdgcn_ballot_w64(gt), be = __builtin_amdgcn_ballot_w64(eq);
        if (lane == 0) { s_wg[wv] = (unsigned)__builtin_popcountll(bg); s_we[wv] = (unsigned)__builtin_popcountll(be); }
        __syncthreads();
        unsigned g_before = s_bg, e_before = s_be, g_tot = 0, e_tot = 0;
        for (int w = 0; w < RS_THREADS / 64; ++w) {
            const unsigned cg = s_wg[w], ce = s_we[w];
            if (w < wv) { g_before += cg; e_before += ce; }
            g_tot += cg; e_tot += ce;
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        g_before += (unsigned)__builtin_popcountll(bg & below);
        e_before += (unsigned)__builtin_popcountll(be & below);
        if (gt || (eq && e_before < kk)) {
            const unsigned pos = g_before + (e_before < kk ? e_before : kk);
            cs[(size_t)f * cap + pos] = v;
            cids[(size_t)f * cap + pos] = (int)i;
        }
        __syncthreads();
        if (tid == 0) { s_bg += g_tot; s_be += e_tot; }
        __syncthreads();
    }
}

static size_t compact_bytes(long nsel, long n) {
    if (n < CP_MIN_N) return 0;
    const long nslabs = (n + CP_SLAB - 1) / CP_SLAB;
    return (size_t)nsel * ((size_t)CP_CAP * 8 + (size_t)nslabs * 8 + 8) + 1024;
}

size_t ip_topk_exhaustive_workspace_bytes(long nsel, long n) {
    return (((size_t)nsel * n * 4 + 255) & ~(size_t)255) + 256 + compact_bytes(nsel, n);
}

int ip_topk_exhaustive(const void* q, const int* qsel, long nsel, const void* c, long n, int d,
                       int k, int dtype, long id_offset, int metric, const float* lower_bound, float* D_out, long* I_out,
                       void* ws, size_t ws_bytes, hipStream_t st) {
    if (nsel <= 0 || n <= 0 || k <= 0 || d <= 0 || (dtype != DT_F32 && dtype != DT_BF16) ||
        d % (dtype == DT_F32 ? 4 : 8) || (metric != 0 && metric != 1)) {
        set_error("ip_topk_exhaustive: need nsel, n, k > 0, d %% 4 == 0 (f32) / d %% 8 == 0 (bf16), metric in {0,1}");
        return SSS_EINVAL;
    }
    if (n >= (1L << 31) || nsel > 65535 || k > RS_MAX_K) { set_error("ip_topk_exhaustive: n < 2^31, nsel <= 65535, k <= 1024"); return SSS_EINVAL; }
    if (ws_bytes < ip_topk_exhaustive_workspace_bytes(nsel, n)) {
        set_error("ip_topk_exhaustive: workspace too small");
        return SSS_EWORKSPACE;
    }
    float* scores = reinterpret_cast<float*>(ws);
    long gx = (n + EX_ROWS - 1) / EX_ROWS;
    if (gx > 8192) gx = 8192;
    const unsigned rb = (unsigned)((n + 255) / 256);
#define SSS_ROWS(D_, DT_) hipLaunchKernelGGL((k_exact_scores_rows<D_, DT_>), dim3(rb), dim3(256), 0, st, q, qsel, (int)nsel, c, n, scores, lower_bound)
    if (metric == 0 && dtype == DT_F32 && d == 64) SSS_ROWS(64, DT_F32);
    else if (metric == 0 && dtype == DT_F32 && d == 128) SSS_ROWS(128, DT_F32);
    else if (metric == 0 && dtype == DT_F32 && d == 256) SSS_ROWS(256, DT_F32);
    else if (metric == 0 && dtype == DT_BF16 && d == 128) SSS_ROWS(128, DT_BF16);
    else if (metric == 0 && dtype == DT_BF16 && d == 256) SSS_ROWS(256, DT_BF16);
    else if (dtype == DT_F32)
        hipLaunchKernelGGL(k_exact_scores<DT_F32>, dim3((unsigned)gx, (unsigned)nsel), dim3(64), 0, st, q, qsel, c, n, d, metric, scores);
    else
        hipLaunchKernelGGL(k_exact_scores<DT_BF16>, dim3((unsigned)gx, (unsigned)nsel), dim3(64), 0, st, q, qsel, c, n, d, metric, scores);
#undef SSS_ROWS
    int rc = check_launch("k_exact_scores");
    if (rc) return rc;
    const float* cs = nullptr; const int* cids = nullptr; const unsigned* ctotal = nullptr;
    if (n >= CP_MIN_N) {
        const int nslabs = (int)((n + CP_SLAB - 1) / CP_SLAB);
        char* p = reinterpret_cast<char*>(ws) + (((size_t)nsel * n * 4 + 255) & ~(size_t)255);
        float* cs_w = reinterpret_cast<float*>(p);                       p += (size_t)nsel * CP_CAP * 4;
        int* cids_w = reinterpret_cast<int*>(p);                         p += (size_t)nsel * CP_CAP * 4;
        unsigned* cnt_gt = reinterpret_cast<unsigned*>(p);               p += (size_t)nsel * nslabs * 4;
        unsigned* cnt_eq = reinterpret_cast<unsigned*>(p);               p += (size_t)nsel * nslabs * 4;
        unsigned* T0 = reinterpret_cast<unsigned*>(p);                   p += (size_t)nsel * 4;
        unsigned* total = reinterpret_cast<unsigned*>(p);
        static bool head_attr[MAX_DEVICES] = {};
        const int hdev = current_device();
        if (!head_attr[hdev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head_threshold), hipFuncAttributeMaxDynamicSharedMemorySize, CP_HEAD * 4);
            head_attr[hdev] = true;
        }
        hipLaunchKernelGGL(k_head_threshold, dim3((unsigned)nsel), dim3(RS_THREADS), (size_t)CP_HEAD * 4, st, scores, n, k, metric, T0);
        hipLaunchKernelGGL(k_count_ge, dim3((unsigned)nslabs, (unsigned)nsel), dim3(256), 0, st, scores, n, metric, T0, nslabs, cnt_gt, cnt_eq);
        hipLaunchKernelGGL(k_scan_slabs, dim3((unsigned)nsel), dim3(64), 0, st, cnt_gt, cnt_eq, nslabs, n, k, total);
        hipLaunchKernelGGL(k_compact_ge, dim3((unsigned)nslabs, (unsigned)nsel), dim3(RS_THREADS), 0, st, scores, n, k, metric, T0, nslabs,
                           cnt_gt, cnt_eq, total, CP_CAP, cs_w, cids_w);
        rc = check_launch("k_compact_ge");
        if (rc) return rc;
        cs = cs_w; cids = cids_w; ctotal = total;
    }
    hipLaunchKernelGGL(k_topk_radix, dim3((unsigned)nsel), dim3(RS_THREADS), 0, st, scores, qsel, n, k, id_offset, metric, D_out, I_out,
                       cs, cids, ctotal, CP_CAP);
    return check_launch("k_topk_radix");
}

}  // namespace sss

// Candidate selection + canonical float64 re-score + proof of exactness, and the k-way merge of
// per-shard results (gfx950).  Second half of what the reference asks of faiss at
// test_amazon_filterd.py:578 (the per-query heap inside IndexFlatIP.search).
//
// Input: the compact candidate keys k_scan (scan.hip) appended per query.  Per query:
//   select the best K2 = k + slack candidates by (scan score desc, id asc), re-score them in
//   float64 from the stored rows in the canonical sequential order (== the oracle's score), order by
//   (score desc, id asc), write the first k, and decide
//   status[q] = 0  proven exact: every row that was NOT re-scored has a scan score at or below
//                  the selection edge (it lost to a full list's tail <= edge, or to the admission
//                  threshold < edge, or it is a candidate ranked below the edge), and
//                  edge + B + one float32 ulp < k-th re-scored score, B bounding the error of the
//                  scan that produced the candidates (err_bound; DT_F16 scores are first divided by
//                  the query's and the corpus' power-of-two scales);
//            != 0  not proven (bit 0: a full list's tail outranks the edge, bit 1: the admission
//                  threshold does, bit 2: near-tie window) -> the caller re-runs the query through
//                  the exhaustive path.
//   k_select_fast  : one wave per query, K2 <= 16, candidates <= FS_CAP (the common case: the
//                    shared threshold leaves a few hundred candidates per query); a query that fails
//                    ONLY the near-tie window gets a second chance with up to 32 candidates
//   k_select_sort  : one workgroup per query, bitonic sort in LDS, any K2 <= SEL_MAX_K2
#include "scan.h"

namespace sss {

constexpr int FS_CAP = 2048;       // candidates a wave stages in LDS (more -> unproven)
constexpr int FS_K2 = 32;          // candidates the wave-per-query kernel can re-score (its second chance widens K2 <= 16 up to this)
constexpr int FS_COL = 16;         // candidate keys a lane keeps in registers (64 x 16 = 1024 candidates; more -> columns in LDS)
constexpr int SEL_MAX_K2 = 512;
#ifndef SSS_SA_ROWS_SMALL
#define SSS_SA_ROWS_SMALL 256
#endif
constexpr int SA_ROWS_SMALL = SSS_SA_ROWS_SMALL;   // k_select_all: survivors re-scored per group (one thread each): the first launch (<= 2048 kept rows) --
                                                   //   k + a few dozen survivors are ONE group up to k = 200-odd (a second group doubles the workgroup's time)
constexpr int SA_ROWS_FULL = 128;                  //   ... the full-capacity launch (128 KB of keys leave room for no more)
constexpr int SA_BYTES = 128;      //               bytes of every row staged through LDS per step
constexpr int SORT_THREADS = 256;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Wave-wide maximum of a u32 on the DPP network (no LDS round trips): row_shr 1/2/4/8 leave each
// 16-lane row's maximum in its last lane, row_bcast15 / row_bcast31 carry it across rows, lane 63
// holds the result.  bound_ctrl = true feeds 0 (the identity of umax) to lanes without a source.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));   // row_shr:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));   // row_shr:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));   // row_shr:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));   // row_shr:8
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));   // row_bcast:15 -> rows 1, 3
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));   // row_bcast:31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit keys: maximum of the high words, then of the low words among the lanes that hold it.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned mh = wave_max_u32(hi);
    const unsigned long long own = __builtin_amdgcn_ballot_w64(hi == mh);
    unsigned ml;
    if ((own & (own - 1)) == 0)                      // one lane holds the best score (the usual case): its low word
        ml = (unsigned)__builtin_amdgcn_readlane((int)lo, __builtin_ctzll(own));
    else
        ml = wave_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}

// Cross-lane hand-off through LDS inside ONE wave: the hardware runs a wave's LDS instructions in
// order; this only stops the compiler from moving memory operations across the point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Canonical float64 score of candidate rows, FOUR lanes per row: part p of a row's 16-byte chunks is
// loaded by lane 4c + p (every load of the row in flight at once: one memory round trip instead of
// four), and the strictly sequential float64 chain runs part after part, handed from lane to lane --
// the same additions in the same order as one lane walking the row.  Candidates [c0, c0 + 16) of `sel`.
__device__ __forceinline__ double dot_chunk(double acc, const char* qrow, int v, f32x4 c, int dtype);
__device__ __forceinline__ void rescore16(const unsigned long long* sel, double* resc, int c0, int c1, const void* C, int rb,
                                          const char* qrow, int dtype, int lane) {
    const int c = c0 + (lane >> 2), p = lane & 3;
    const int per = rb / 64;                                       // chunks per part: 4 / 8 / 16 / 32 (rows of 256 .. 2048 bytes)
    const unsigned long long key = c < c1 ? sel[c] : 0ull;
    const bool live = key != 0 && key_id(key) >= 0;
    constexpr int MAXP = 32;
    f32x4 ch[MAXP];
    const char* row = reinterpret_cast<const char*>(C) + (size_t)(live ? key_id(key) : 0) * rb + (size_t)p * per * 16;
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
        if (live && i < per) ch[i] = *reinterpret_cast<const f32x4*>(row + i * 16);
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (live && p == s) {
#pragma unroll
            for (int i = 0; i < MAXP; ++i)
                if (i < per) acc = dot_chunk(acc, qrow, s * per + i, ch[i], dtype);
        }
        if (s < 3) {                                               // hand the chain to the next part's lane
            const double up = __shfl_up(acc, 1);
            if (p == s + 1) acc = up;
        }
    }
    if (c < c1 && p == 3) resc[c] = live ? acc : 0.0;
}

// acc += sum over the elements of one 16-byte chunk (4 f32 or 8 bf16), sequential in k
__device__ __forceinline__ double dot_chunk(double acc, const char* qrow, int v, f32x4 c, int dtype) {
    if (dtype == DT_F32) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qrow + v * 16);
        acc += (double)qv.x * (double)c.x;
        acc += (double)qv.y * (double)c.y;
        acc += (double)qv.z * (double)c.z;
        acc += (double)qv.w * (double)c.w;
        return acc;
    }
    const u32x4 cu = __builtin_bit_cast(u32x4, c);
    const u32x4 qu = *reinterpret_cast<const u32x4*>(qrow + v * 16);
#define SSS_BF2(w)                                                                                              \
    acc += (double)__builtin_bit_cast(float, qu.w << 16) * (double)__builtin_bit_cast(float, cu.w << 16);      \
    acc += (double)__builtin_bit_cast(float, qu.w & 0xFFFF0000u) * (double)__builtin_bit_cast(float, cu.w & 0xFFFF0000u);
    SSS_BF2(x) SSS_BF2(y) SSS_BF2(z) SSS_BF2(w)
#undef SSS_BF2
    return acc;
}

// Canonical float64 score of ONE stored row by one thread: the row's 16-byte chunks are fetched sixteen at a time
// (sixteen loads in flight, one memory round trip per 256 bytes instead of one per chunk) and folded into the
// strictly sequential chain in k order.
__device__ __forceinline__ double rescore_row(const char* qrow, const char* row, int nchunks, int dtype) {
    double acc = 0.0;
    int v0 = 0;
    // full batches: sixteen UNCONDITIONAL loads (a guarded load sits in a basic block of its own and hipcc then drains
    // vmcnt before every one of them -- measured on 1600-wide rows: one load in flight, 1.4 us per 16-byte chunk)
    for (; v0 + 16 <= nchunks; v0 += 16) {
        f32x4 c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = *reinterpret_cast<const f32x4*>(row + (size_t)(v0 + i) * 16);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = dot_chunk(acc, qrow, v0 + i, c[i], dtype);
    }
    if (v0 < nchunks) {                             // tail: the same loads clamped to the row's last chunk, their results unused
        f32x4 c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = *reinterpret_cast<const f32x4*>(row + (size_t)min(v0 + i, nchunks - 1) * 16);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (v0 + i < nchunks) acc = dot_chunk(acc, qrow, v0 + i, c[i], dtype);
    }
    return acc;
}

__device__ __forceinline__ float elem_to_f32(const void* row, int kk, int dtype) {
    if (dtype == DT_F32) return reinterpret_cast<const float*>(row)[kk];
    const unsigned short b = reinterpret_cast<const unsigned short*>(row)[kk];
    return __builtin_bit_cast(float, (unsigned)b << 16);          // bf16 -> f32 is exact
}

// B = rounding-error bound of the scan's score of any row, by what the scan computed:
//   DT_F32   k-ordered f32 fma chain:                               d * 2^-24 * |q| |c|
//   DT_BF16  exact bf16 products summed in f32 with unspecified internal order / truncation:
//                                                                  d * 2^-23 * |q| |c|
//   DT_SPLIT x = xh + xl + xr with |x - xh| <= 2^-8 |x|, |xr| <= 2^-16 |x| (two roundings to 8
//            significant bits), likewise y; the scan sums xh*yh + xh*yl + xl*yh, so per element it
//            misses xl*yl + xr*y + (xh + xl)*yr <= 3.03 * 2^-16 |x||y|, and sum |x_k||y_k| <= |q||c|;
//            the 3d exact products (sum of magnitudes <= 1.016 |q||c|) are accumulated in f32 like
//            the bf16 case:                        (3.03 * 2^-16 + 3d * 2^-23 * 1.016) * |q| |c|
//   DT_F16   corpus and query each scaled by a power of two (exact) and rounded to 11 significant
//            bits; with c^ = c + rc, q^ = q + rq the scan sums c^ . q^ = c.q + rc.q + c^.rq, so by
//            Cauchy-Schwarz the rounding costs at most Rc |q| + (|c| + Rc) Rq, where Rc = the largest
//            row residual norm |c^ - c| over the corpus (measured when the image is built, passed in;
//            worst case 2^-11 |c|) and Rq = this query's residual norm (measured here).  Elements below
//            the f16 normal range (2^-14 in the scaled domain = 2^-26 of the largest element) add at
//            most 2^-25 sqrt(d) |q||c| even if the matrix unit flushed them; products of two f16 are
//            exact in f32 and accumulate like the bf16 case:
//                         Rc |q| + (|c| + Rc) Rq + (2^-25 sqrt(d) + d * 2^-23) |q| |c|
// (each with 2 % headroom; |c| <= the corpus' largest row norm).
__device__ __forceinline__ double err_bound(int d, int scan_dtype, double qnorm, double cmax, double c_resid, double q_resid) {
    double b;
    if (scan_dtype == DT_F32) b = (double)d * 5.9604644775390625e-08 * qnorm * cmax;
    else if (scan_dtype == DT_BF16) b = (double)d * 1.1920928955078125e-07 * qnorm * cmax;
    else if (scan_dtype == DT_SPLIT) b = (3.03 * 1.52587890625e-05 + 3.0 * (double)d * 1.1920928955078125e-07 * 1.016) * qnorm * cmax;
    else b = c_resid * qnorm + (cmax + c_resid) * q_resid +
             (2.98023223876953125e-08 * sqrt((double)d) + (double)d * 1.1920928955078125e-07) * qnorm * cmax;
    return b * 1.02;
}

// Hand the query's state words back zeroed (scan.h: the contract that replaces a per-call memset).
__device__ __forceinline__ void clear_state(const SelectArgs& A, int q, int t, int nthreads) {
    for (int j = t; j < A.J; j += nthreads) A.slots[(size_t)q * SLOT_STRIDE + j] = 0u;
    if (t == 0) { A.cnt[q] = 0u; A.maxlast[q] = 0ull; }
}

// The scan's final threshold word of query q from its J slots (0: not enough classes ever published): the min, or --
// rank-selected scans (cert == 1, J == 16: scan_dev.h tau_select16) -- the exact (skip + 1)-th smallest of the 16
// (the scan's own, cheaper pick lies at or below it).
// Called by a whole wave.
__device__ __forceinline__ unsigned final_tau_ord(const unsigned* slots, int J, int lane, int skip) {
    if (skip > 0 && J == 16) {
        const unsigned mine = slots[lane & 15];
        int rank = 0;                                             // values ordered by (word, slot index): ranks 0 .. 15
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned o = (unsigned)__shfl((int)mine, j);
            rank += (o < mine || (o == mine && j < (lane & 15))) ? 1 : 0;
        }
        const unsigned long long own = __builtin_amdgcn_ballot_w64(lane < 16 && rank == skip);
        return (unsigned)__builtin_amdgcn_readlane((int)mine, __builtin_ctzll(own));
    }
    unsigned m = 0xFFFFFFFFu;
    for (int j = lane; j < J; j += 64) m = min(m, slots[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, o));
    return m;
}

struct Verdict { int st; };

// Shared tail: rank the K2 re-scored candidates, write results, decide the status.  Called by
// the threads [0, nthreads) of one query with sel/resc in LDS; lane0 writes the status.
template <int NT>
__device__ __forceinline__ void rank_and_write(const unsigned long long* sel, const double* resc, int K2, int k,
                                               long id_offset, float* Dq, long* Iq, int t, int* s_nvalid,
                                               double* s_kth) {
    for (int c = t; c < K2; c += NT) {
        const unsigned long long key = sel[c];
        const int id = key_id(key);
        if (key == 0 || id < 0) continue;
        const float sc = (float)resc[c];
        int rank = 0;
        for (int j = 0; j < K2; ++j) {
            const int idj = key_id(sel[j]);
            if (j == c || sel[j] == 0 || idj < 0) continue;
            const float sj = (float)resc[j];
            if (sj > sc || (sj == sc && idj < id)) ++rank;
        }
        atomicAdd(s_nvalid, 1);
        if (rank < k) { Dq[rank] = sc; Iq[rank] = (long)id + id_offset; }
        if (rank == k - 1) *s_kth = resc[c];
    }
}

// unscale: what a scan score must be multiplied by to be a score (1 except for DT_F16).  A row outside
// the candidates has scan score <= the edge's, so its exact score is <= edge * unscale + B; it can
// neither enter the top k nor tie with the k-th result AFTER the rounding to float32 if that stays
// below kth by more than one float32 ulp of kth.
__device__ __forceinline__ int decide_status(unsigned long long edge, unsigned long long maxlast, unsigned tau_o,
                                             int J, int nvalid, int k, double kth, double B, double unscale) {
    int st = 0;
    const bool edge_real = edge != 0 && key_id(edge) >= 0;
    if (maxlast > edge) st |= 1;                                  // a full list may hide a contender
    if (J > 0 && tau_o > ORD_NEG_INF) {                           // rows were rejected at or below ord2f(tau_o - 1)
        if (!(edge_real && f2ord(key_score(edge)) >= tau_o)) st |= 2;
    }
    if (edge_real && nvalid >= k) {
        const double reach = (double)key_score(edge) * unscale + B + 2.4e-7 * fabs(kth) + 1e-44;
        if (!(reach < kth)) st |= 4;                              // the error window reaches the k-th result
    }
    return st;
}

// squared rounding residual of query element v under the DT_F16 scan's scaling + rounding (scan.hip)
__device__ __forceinline__ double f16_resid2(float v, int sh) {
    const float back = ldexpf((float)(_Float16)ldexpf(v, sh), -sh);
    const double r = (double)back - (double)v;
    return r * r;
}

// 2^-(corpus shift + query shift) of a DT_F16 scan (scan.h), from the query row's largest magnitude
__device__ __forceinline__ double scan_unscale(const SelectArgs& A, float q_amax) {
    if (A.scan_dtype != DT_F16) return 1.0;
    return ldexp(1.0, -(A.corpus_shift + f16_shift(q_amax)));
}

// ------------------------------------------------------------------------------------------
// One wave per query.  LDS per wave: keys[FS_CAP] | sel[FS_K2] | resc[FS_K2] | counters | qrow[rb]
__global__ __launch_bounds__(256) void k_select_fast(const SelectArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wv;
    if (q >= A.nq) return;                                        // whole wave; no block-level sync below
    const int rb = A.d * (A.dtype == DT_F32 ? 4 : 2);
    const size_t per_wave = (size_t)FS_CAP * 8 + FS_K2 * 16 + 16 + rb;
    char* base = smem + wv * ((per_wave + 15) & ~(size_t)15);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(base);
    unsigned long long* sel = keys + FS_CAP;
    double* resc = reinterpret_cast<double*>(sel + FS_K2);
    int* s_nvalid = reinterpret_cast<int*>(resc + FS_K2);
    double* s_kth = reinterpret_cast<double*>(s_nvalid + 2);
    char* qrow = reinterpret_cast<char*>(s_kth + 1);

    const int K2 = A.K2, k = A.k;
    float* Dq = A.D_out + (size_t)q * k;
    long* Iq = A.I_out + (size_t)q * k;
    const int M = (int)min(A.cnt[q], (unsigned)A.cap);            // (the append form of k_scan counts what it could not store, too)
    if (M > FS_CAP) {                                             // adversarial input: let the exhaustive path decide
        for (int j = lane; j < k; j += 64) { Dq[j] = -3.4028234663852886e38f; Iq[j] = -1; }   // (no k-th score known)
        if (lane == 0) { A.status[q] = 1; if (A.unproven_count) atomicAdd(A.unproven_count, 1); }
        clear_state(A, q, lane, 64);
        return;
    }
    // ---- the query row, and this lane's COLUMN of the candidate keys (keys lane, lane + 64, ...).  Up to 1024
    // candidates (the usual few hundred) a column is 16 keys in registers, sorted once by a bitonic network: a
    // selection round is then one wave-wide maximum of the column heads and a register shift in the owner lane.
    // More candidates: columns stay in LDS and the owner rescans its column after every round (as in round 2,
    // when this was the only form and the 16 rounds took 40 % of the kernel).
    const unsigned long long* ck = A.cand + (size_t)q * A.cap;
    const bool in_regs = M <= 64 * FS_COL;                        // wave-uniform
    unsigned long long col[FS_COL];
    unsigned long long best = 0; int bidx = -1;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < FS_COL; ++j) col[j] = lane + 64 * j < M ? ck[lane + 64 * j] : 0ull;
#pragma unroll
        for (int kk = 2; kk <= FS_COL; kk <<= 1)
#pragma unroll
            for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
                for (int x = 0; x < FS_COL; ++x) {
                    const int y = x ^ jj;
                    if (y > x) {
                        const unsigned long long a = col[x], b = col[y];
                        const bool sw = ((x & kk) == 0) ? a < b : a > b;      // descending overall
                        col[x] = sw ? b : a; col[y] = sw ? a : b;
                    }
                }
        best = col[0];
    } else {
        for (int i = lane; i < M; i += 64) {
            const unsigned long long v = ck[i];
            keys[i] = v;
            if (v > best) { best = v; bidx = i; }
        }
    }
    // the owner of the round's maximum retires it: its next-best key becomes its column head
    auto retire = [&]() __attribute__((always_inline)) {
        if (in_regs) {
#pragma unroll
            for (int j = 0; j + 1 < FS_COL; ++j) col[j] = col[j + 1];
            col[FS_COL - 1] = 0ull;
            best = col[0];
        } else {
            keys[bidx] = 0;
            best = 0; bidx = -1;
            for (int i = lane; i < M; i += 64) {
                const unsigned long long v = keys[i];
                if (v > best) { best = v; bidx = i; }
            }
        }
    };
    for (int i = lane; i < rb / 16; i += 64)
        reinterpret_cast<f32x4*>(qrow)[i] = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(A.Q) + (size_t)q * rb)[i];
    if (lane == 0) { *s_nvalid = 0; *s_kth = 0.0; }
    wave_sync();
    // ---- K2 rounds: wave-wide arg-max, the owner lane retires its key
    for (int it = 0; it < K2; ++it) {
        const unsigned long long w = wave_max_u64(best);
        if (lane == 0) sel[it] = w;
        if (w != 0 && best == w) retire();                        // keys of real candidates are unique
    }
    wave_sync();
    // ---- float64 re-score: one lane per candidate walks its corpus row (16-byte loads straight from
    // L2 / HBM, several in flight) sequentially in k -- the canonical order
    for (int c0 = 0; c0 < K2; c0 += 16) rescore16(sel, resc, c0, K2, A.C, rb, qrow, A.dtype, lane);
    double qn2 = 0.0;
    float q_amax = 0.f;
    for (int kk = lane; kk < A.d; kk += 64) {
        const float v = elem_to_f32(qrow, kk, A.dtype);
        qn2 += (double)v * (double)v;
        q_amax = fmaxf(q_amax, fabsf(v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                              // the norm only feeds the error BOUND: order-free
        qn2 += __shfl_xor(qn2, o);
        q_amax = fmaxf(q_amax, __shfl_xor(q_amax, o));
    }
    double rq2 = 0.0;
    if (A.scan_dtype == DT_F16) {
        const int sh = f16_shift(q_amax);
        for (int kk = lane; kk < A.d; kk += 64) rq2 += f16_resid2(elem_to_f32(qrow, kk, A.dtype), sh);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rq2 += __shfl_xor(rq2, o);
    }
    wave_sync();
    rank_and_write<64>(sel, resc, K2, k, A.id_offset, Dq, Iq, lane, s_nvalid, s_kth);
    wave_sync();
    const int nvalid = *s_nvalid;
    for (int j = nvalid + lane; j < k; j += 64) {                 // faiss pads missing results
        Dq[j] = -3.4028234663852886e38f;
        Iq[j] = -1;
    }
    const unsigned tau_o = A.J > 0 ? final_tau_ord(A.slots + (size_t)q * SLOT_STRIDE, A.J, lane, A.tau_skip) : 0u;
    const unsigned long long maxlast = A.maxlast[q];
    wave_sync();                                                  // every lane has read the state words
    clear_state(A, q, lane, 64);
    const double B = err_bound(A.d, A.scan_dtype, sqrt(qn2), (double)A.corpus_max_norm, (double)A.corpus_resid, sqrt(rq2));
    const double unscale = scan_unscale(A, q_amax);
    int st = decide_status(sel[K2 - 1], maxlast, tau_o, A.J, nvalid, k, *s_kth, B, unscale);
    // ---- second chance for a query whose ONLY problem is the near-tie window (status 4): the pool
    // usually holds more candidates than the K2 the threshold certifies, and it is complete down to
    // max(threshold, largest tail of a full list).  Take candidates from that region -- until the next one
    // already lies below the k-th result by more than the error window (the k-th result can only rise when
    // candidates are added), up to FS_K2 in all -- re-score the new ones, rank again; what stays outside
    // is bounded by the first key not taken (or by the region's floor).
    if (st == 4 && K2 < FS_K2) {                                  // wave-uniform (all lanes computed st)
        const bool has_tau = A.J > 0 && tau_o > ORD_NEG_INF;
        int K2x = K2;
        float edge_score = -INFINITY;
        bool open_end = false;                                    // stopped by the budget: edge = last key taken
        const double kth0 = *s_kth;                               // lower bound of the final k-th result
        for (; K2x < FS_K2; ++K2x) {
            const unsigned long long w = wave_max_u64(best);
            if (w != 0 && key_id(w) >= 0 && (double)key_score(w) * unscale + B + 2.4e-7 * fabs(kth0) + 1e-44 < kth0) {
                edge_score = key_score(w);                        // far enough below: nothing from here on can matter
                break;
            }
            const bool inside = w != 0 && key_id(w) >= 0 && w >= maxlast && (!has_tau || f2ord(key_score(w)) >= tau_o);
            if (!inside) {
                if (w != 0 && key_id(w) >= 0) edge_score = key_score(w);
                break;
            }
            if (lane == 0) sel[K2x] = w;
            if (best == w) retire();
        }
        if (K2x == FS_K2) open_end = true;
        wave_sync();
        if (K2x > K2) {
            for (int c0 = K2; c0 < K2x; c0 += 16) rescore16(sel, resc, c0, K2x, A.C, rb, qrow, A.dtype, lane);
            if (lane == 0) { *s_nvalid = 0; *s_kth = 0.0; }
            wave_sync();
            rank_and_write<64>(sel, resc, K2x, k, A.id_offset, Dq, Iq, lane, s_nvalid, s_kth);
            wave_sync();
        }
        // (K2x == K2: the very next candidate already lies far below -- phase 1 only failed because it
        //  measures the window from the last INCLUDED key)
        if (open_end) {
            edge_score = key_score(sel[K2x - 1]);
        } else {                                                  // everything else lies under the region's floor
            if (has_tau) edge_score = fmaxf(edge_score, ord2f(tau_o - 1));
            if (maxlast != 0 && key_id(maxlast) >= 0) edge_score = fmaxf(edge_score, key_score(maxlast));
        }
        const double kth = *s_kth;
        const double reach = (double)edge_score * unscale + B + 2.4e-7 * fabs(kth) + 1e-44;
        st = (*s_nvalid >= k && reach < kth) ? 0 : 4;
    }
    if (lane == 0) {
        A.status[q] = st;
        if (st && A.unproven_count) atomicAdd(A.unproven_count, 1);
    }
}

// The k-th largest score ordinal (high word of the keys) among keys[0 .. M), k <= M, by the whole workgroup of
// SORT_THREADS = 256 threads: a radix descent, eight bits a pass -- a 256-bin histogram (LDS atomics) of the keys
// that still match the prefix, then the bin holding the k-th from the top (one wave: four bins a lane, a suffix
// sum by shuffles) -- four passes over the keys instead of a sort of up to 8192 of them (round 3; it was 32 one-bit
// passes, 3 barriers each).  s_hist: 260 shared words; every thread returns the same value.
template <typename OrdAt>
__device__ __forceinline__ unsigned kth_largest_of(OrdAt ord_at, int M, int k, int tid, unsigned* s_hist) {
    unsigned prefix = 0u, mask = 0u;
    unsigned kk = (unsigned)k;                      // rank, from the top, inside the bucket that matches the prefix
    for (int shift = 24; shift >= 0; shift -= 8) {
        s_hist[tid] = 0u;
        __syncthreads();
        for (int x = tid; x < M; x += SORT_THREADS) {
            const unsigned o = ord_at(x);
            if ((o & mask) == prefix) atomicAdd(&s_hist[(o >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            const unsigned h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
            const unsigned mine = h0 + h1 + h2 + h3;
            unsigned suf = mine;                    // sum over this lane and every higher one
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = (unsigned)__shfl_down((int)suf, o);
                if (tid + o < 64) suf += v;
            }
            const unsigned above = suf - mine;
            if (above < kk && suf >= kk) {          // the k-th from the top falls into this lane's four bins (exactly one lane)
                unsigned cum = above;
                int b = 3;
                if (cum + h3 < kk) { cum += h3; b = 2; if (cum + h2 < kk) { cum += h2; b = 1; if (cum + h1 < kk) { cum += h1; b = 0; } } }
                s_hist[256] = (unsigned)(4 * tid + b);
                s_hist[257] = kk - cum;
            }
        }
        __syncthreads();
        prefix |= s_hist[256] << shift;
        mask |= 255u << shift;
        kk = s_hist[257];
        __syncthreads();                            // (the two words are rewritten in the next pass)
    }
    return prefix;
}

__device__ __forceinline__ unsigned kth_largest_ord(const unsigned long long* keys, int M, int k, int tid, unsigned* s_hist) {
    return kth_largest_of([&](int x) { return (unsigned)(keys[x] >> 32); }, M, k, tid, s_hist);
}

// The k-th largest 64-bit KEY among keys[0 .. M) (keys are unique: score ordinal << 32 | ~row id), k <= M: the k-th largest
// high word, then -- inside its tie group -- the low word that completes the count.  Exactly k keys lie at or above the
// result.  Whole workgroup; s_cnt: one shared word.
__device__ __forceinline__ unsigned long long kth_largest_key(const unsigned long long* keys, int M, int k, int tid, unsigned* s_hist,
                                                              unsigned* s_cnt) {
    const unsigned sk = kth_largest_of([&](int x) { return (unsigned)(keys[x] >> 32); }, M, k, tid, s_hist);
    if (tid == 0) *s_cnt = 0u;
    __syncthreads();
    unsigned gt = 0u;
    for (int x = tid; x < M; x += SORT_THREADS) gt += (unsigned)(keys[x] >> 32) > sk ? 1u : 0u;
    if (gt) atomicAdd(s_cnt, gt);
    __syncthreads();
    const int need_eq = k - (int)*s_cnt;                                // >= 1: the k-th itself has ordinal sk
    __syncthreads();
    const unsigned lowk = kth_largest_of([&](int x) { const unsigned long long kx = keys[x]; return (unsigned)(kx >> 32) == sk ? (unsigned)kx : 0u; },
                                         M, need_eq, tid, s_hist);
    return ((unsigned long long)sk << 32) | lowk;
}

// ------------------------------------------------------------------------------------------
// One workgroup per query: bitonic sort (descending) of the candidate keys in LDS.
__global__ __launch_bounds__(SORT_THREADS) void k_select_sort(const SelectArgs A, int cap_pow2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);             // [cap_pow2]
    double* resc = reinterpret_cast<double*>(keys + cap_pow2);                           // [SEL_MAX_K2]
    char* qrow = reinterpret_cast<char*>(resc + SEL_MAX_K2);
    __shared__ int s_nvalid;
    __shared__ double s_kth;
    __shared__ double s_q2[SORT_THREADS / 64];
    __shared__ float s_amax[SORT_THREADS / 64];
    __shared__ unsigned s_hist[260];
    __shared__ unsigned s_cnt;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int rb = A.d * (A.dtype == DT_F32 ? 4 : 2);
    const int K2 = A.K2, k = A.k;
    float* Dq = A.D_out + (size_t)q * k;
    long* Iq = A.I_out + (size_t)q * k;
    const int M = (int)min(A.cnt[q], (unsigned)A.cap);            // (the append form of k_scan counts what it could not store, too)
    int M2 = 64;
    while (M2 < M || M2 < K2) M2 <<= 1;                            // <= cap_pow2 by construction
    const unsigned long long* ck = A.cand + (size_t)q * A.cap;
    for (int i = tid; i < M2; i += SORT_THREADS) keys[i] = i < M ? ck[i] : 0ull;
    for (int i = tid; i < rb / 16; i += SORT_THREADS)
        reinterpret_cast<f32x4*>(qrow)[i] = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(A.Q) + (size_t)q * rb)[i];
    if (tid == 0) { s_nvalid = 0; s_kth = 0.0; }
    __syncthreads();
    // Only the K2 best candidates are needed, in order: with many more than that (the lists of a k = 500 search hold up to
    // 2048), they are SELECTED first (kth_largest_key) and only they are sorted -- 45 barrier stages over 512 keys instead of
    // 66 over 2048.  (The re-score slots serve as the compaction buffer: nothing has been re-scored yet.)
    int Ms = M2;
    {
        int K2p = 64;
        while (K2p < K2) K2p <<= 1;
        if (M > 2 * K2p) {
            const unsigned long long T = kth_largest_key(keys, M, K2, tid, s_hist, &s_cnt);
            unsigned long long* tmp = reinterpret_cast<unsigned long long*>(resc);
            __syncthreads();
            if (tid == 0) s_cnt = 0u;
            __syncthreads();
            for (int x = tid; x < M; x += SORT_THREADS) {
                const unsigned long long kx = keys[x];
                if (kx >= T) tmp[atomicAdd(&s_cnt, 1u)] = kx;
            }
            __syncthreads();
            for (int x = tid; x < K2p; x += SORT_THREADS) keys[x] = x < K2 ? tmp[x] : 0ull;
            __syncthreads();
            Ms = K2p;
        }
    }
    for (int kk = 2; kk <= Ms; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < Ms; i += SORT_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool desc = (i & kk) == 0;
                    if (desc ? a < b : a > b) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    // ---- float64 re-score straight from global (one thread per candidate, sequential in k)
    double q2 = 0.0;
    float q_amax = 0.f;
    for (int kx = tid; kx < A.d; kx += SORT_THREADS) {
        const float v = elem_to_f32(qrow, kx, A.dtype);
        q2 += (double)v * (double)v;
        q_amax = fmaxf(q_amax, fabsf(v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { q2 += __shfl_xor(q2, o); q_amax = fmaxf(q_amax, __shfl_xor(q_amax, o)); }
    if (lane == 0) { s_q2[tid >> 6] = q2; s_amax[tid >> 6] = q_amax; }
    for (int c = tid; c < K2; c += SORT_THREADS) {
        const unsigned long long key = keys[c];
        const int id = key_id(key);
        double acc = 0.0;
        if (key != 0 && id >= 0) acc = rescore_row(qrow, reinterpret_cast<const char*>(A.C) + (size_t)id * rb, rb / 16, A.dtype);
        resc[c] = acc;
    }
    __syncthreads();
    rank_and_write<SORT_THREADS>(keys, resc, K2, k, A.id_offset, Dq, Iq, tid, &s_nvalid, &s_kth);
    __syncthreads();
    const int nvalid = s_nvalid;
    for (int j = nvalid + tid; j < k; j += SORT_THREADS) {
        Dq[j] = -3.4028234663852886e38f;
        Iq[j] = -1;
    }
    if (tid < 64) {                                               // wave 0 alone touches the state from here on
        const unsigned tau_o = A.J > 0 ? final_tau_ord(A.slots + (size_t)q * SLOT_STRIDE, A.J, lane, A.tau_skip) : 0u;
        const unsigned long long maxlast = A.maxlast[q];
        wave_sync();
        clear_state(A, q, lane, 64);
        if (tid == 0) {
            double qq = 0.0;
            float am = 0.f;
            for (int w = 0; w < SORT_THREADS / 64; ++w) { qq += s_q2[w]; am = fmaxf(am, s_amax[w]); }
            double rq2 = 0.0;
            if (A.scan_dtype == DT_F16) {
                const int sh = f16_shift(am);
                for (int kx = 0; kx < A.d; ++kx) rq2 += f16_resid2(elem_to_f32(qrow, kx, A.dtype), sh);
            }
            const double B = err_bound(A.d, A.scan_dtype, sqrt(qq), (double)A.corpus_max_norm, (double)A.corpus_resid, sqrt(rq2));
            const int st = decide_status(keys[K2 - 1], maxlast, tau_o, A.J, nvalid, k, s_kth, B, scan_unscale(A, am));
            A.status[q] = st;
            if (st && A.unproven_count) atomicAdd(A.unproven_count, 1);
        }
    }
}

// ------------------------------------------------------------------------------------------
// THRESHOLD RUNG (between the fused search and the exhaustive kernels).  A query the fused search left
// unproven still has a valid LOWER BOUND of its k-th best score: lb = its k-th re-scored candidate.  A row
// whose exact score could reach lb (or tie with it after the rounding to float32) has a scan score above
//     thr = (lb - B - one float32 ulp of lb) / unscale          (B, unscale: as in decide_status)
// so ONE more scan of the corpus for just those queries (k_scan<..., THR>) that keeps EVERY row above thr,
// followed by the canonical re-score of all of them, is exact whatever the reason the proof failed -- near
// ties inside the scan's error window and exact ties (duplicate rows) alike -- as long as the rows above thr
// fit the candidate capacity; otherwise the query stays unproven and goes to the exhaustive kernels.
//
// k_thr_prepare: one wave per selected query: its threshold in the scan's domain, counter zeroed.
// B (scan error bound) and unscale (scan score -> score factor) of query row q, computed by ONE wave (all 64 lanes
// call it; every lane returns the same values).
__device__ __forceinline__ void query_bound(const ThrArgs& A, int i, int q, int lane, double& B, double& unscale) {
    if (A.qb != nullptr && A.qb_ready) { B = A.qb[2 * (size_t)i]; unscale = A.qb[2 * (size_t)i + 1]; return; }   // (selected query i = row q)
    const int rb = A.d * (A.dtype == DT_F32 ? 4 : 2);
    const char* qrow = reinterpret_cast<const char*>(A.Q) + (size_t)q * rb;
    double qn2 = 0.0;
    float q_amax = 0.f;
    for (int kk = lane; kk < A.d; kk += 64) {
        const float v = elem_to_f32(qrow, kk, A.dtype);
        qn2 += (double)v * (double)v;
        q_amax = fmaxf(q_amax, fabsf(v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { qn2 += __shfl_xor(qn2, o); q_amax = fmaxf(q_amax, __shfl_xor(q_amax, o)); }
    double rq2 = 0.0;
    if (A.scan_dtype == DT_F16) {
        const int sh = f16_shift(q_amax);
        for (int kk = lane; kk < A.d; kk += 64) rq2 += f16_resid2(elem_to_f32(qrow, kk, A.dtype), sh);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rq2 += __shfl_xor(rq2, o);
    }
    B = err_bound(A.d, A.scan_dtype, sqrt(qn2), (double)A.corpus_max_norm, (double)A.corpus_resid, sqrt(rq2));
    unscale = A.scan_dtype == DT_F16 ? ldexp(1.0, -(A.corpus_shift + f16_shift(q_amax))) : 1.0;
    if (A.qb != nullptr && lane == 0) { A.qb[2 * (size_t)i] = B; A.qb[2 * (size_t)i + 1] = unscale; }
}

// The scan threshold that a known lower bound `lb` of the query's k-th score allows: rows the scan does NOT keep have
// scan score <= thr, hence exact score <= thr * unscale + B < lb - ulp32(lb).  -inf when no bound is known (-FLT_MAX).
__device__ __forceinline__ float thr_from_bound(double lb, double B, double unscale) {
    const double t = (lb - B - 2.4e-7 * fabs(lb) - 1e-44) / unscale;
    float thr = (float)t;                                                   // round to nearest, then step below
    if ((double)thr >= t) thr = nextafterf(thr, -INFINITY);
    if (!(lb > -3.0e38)) thr = -INFINITY;
    return thr;
}

// keep mode (one wave): the rows kept so far were kept under an OLDER, lower threshold over tiles the next scan will not
// visit again; those that pass the new one stay (compacted in place, in order: a lane writes at or below the index it
// read, and the whole wave has read a chunk before any of it is written).  An overflowed array stays overflowed.
__device__ __forceinline__ void prune_kept(const ThrArgs& A, int i, float thr, int lane) {
    const unsigned M = A.cnt[i];
    if (M > (unsigned)A.cap) return;
    unsigned long long* ck = const_cast<unsigned long long*>(A.cand) + (size_t)i * A.cap;
    unsigned out = 0u;
    for (unsigned c0 = 0; c0 < M; c0 += 64) {
        const unsigned c = c0 + lane;
        const unsigned long long key = c < M ? ck[c] : 0ull;
        const bool kp = c < M && key_score(key) > thr;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(kp);
        const unsigned pos = out + (unsigned)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (kp) ck[pos] = key;
        out += (unsigned)__builtin_popcountll(mask);
    }
    if (lane == 0) A.cnt[i] = out;
}

__global__ __launch_bounds__(256) void k_thr_prepare(const ThrArgs A) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= A.nsel) return;
    const int q = A.qsel[i];
    double B, unscale;
    query_bound(A, i, q, lane, B, unscale);                                 // (every lane ends up with the same B / unscale)
    const float thr = thr_from_bound((double)A.D_out[(size_t)q * A.k + A.k - 1], B, unscale);    // -FLT_MAX when no k-th score is known
    if (lane == 0) A.thr[i] = thr;
    if (!A.keep) {
        if (lane == 0) A.cnt[i] = 0u;
        return;
    }
    prune_kept(A, i, thr, lane);
}

// sss_ip_topk_long, before the first scan: one wave per query (scan.h: launch_long_setup).
__global__ __launch_bounds__(256) void k_long_setup(const ThrArgs A, int* __restrict__ qsel, _Float16* __restrict__ qimg) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= A.nsel) return;
    if (qimg != nullptr) {                          // f32 queries -> f16 image scaled by the query's own power of two (scan.h f16_shift)
        const float* row = reinterpret_cast<const float*>(A.Q) + (size_t)i * A.d;
        float amax = 0.f;
        for (int kk = lane; kk < A.d; kk += 64) amax = fmaxf(amax, fabsf(row[kk]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        const int sh = f16_shift(amax);
        for (int kk = lane; kk < A.d; kk += 64) qimg[(size_t)i * A.d + kk] = (_Float16)ldexpf(row[kk], sh);
    }
    for (int j = lane; j < A.k; j += 64) A.D_out[(size_t)i * A.k + j] = -3.4028234663852886e38f;   // "no bound known"
    double B, unscale;
    query_bound(A, i, i, lane, B, unscale);         // fills the cache (A.qb_ready == 0 here)
    if (lane == 0) { qsel[i] = i; A.thr[i] = -INFINITY; A.cnt[i] = 0u; A.status[i] = 1; }
}

// descending bitonic sort of keys[0 .. M2) (M2 a power of two) by the whole workgroup
__device__ __forceinline__ void sort_desc(unsigned long long* keys, int M2, int tid) {
    for (int kk = 2; kk <= M2; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int x = tid; x < M2; x += SORT_THREADS) {
                const int ixj = x ^ j;
                if (ixj > x) {
                    const unsigned long long a = keys[x], b = keys[ixj];
                    const bool desc = (x & kk) == 0;
                    if (desc ? a < b : a > b) { keys[x] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// k_bound_prepare (sss_ip_topk_long, between two levels; scan.h: launch_bound_prepare): no row is read.  At least k of
// the kept rows have a scan score >= the k-th largest kept scan score s_k, so at least k rows have an exact score
// >= s_k * unscale - B: a valid LOWER BOUND of the query's true k-th score, written to column k-1 of its row of D_out
// (left unchanged when the kept rows overflowed the capacity or are fewer than k) -- and the next level's threshold
// straight from it (what k_thr_prepare would compute in a launch of its own).  The four radix passes run over LDS: read
// from the array in global memory they moved 4 x 64 KB per query -- 270 MB for the first level of a 1024-query search
// (every one of its 8192 sampled rows is kept), 70 us.  One workgroup per query; the selection by all of it, the rest
// by its first wave.
__global__ __launch_bounds__(SORT_THREADS) void k_bound_prepare(const ThrArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* ords = reinterpret_cast<unsigned*>(smem);                 // [cap] score ordinals of the kept rows
    __shared__ unsigned s_hist[260];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int q = A.qsel[i];
    const int k = A.k;
    const unsigned M = A.cnt[i];
    const bool have = M <= (unsigned)A.cap && (int)M >= k;              // (workgroup-uniform)
    unsigned sk = 0u;
    if (have) {
        const unsigned long long* ck = A.cand + (size_t)i * A.cap;
        for (int x = tid; x < (int)M; x += SORT_THREADS) ords[x] = (unsigned)(ck[x] >> 32);
        __syncthreads();
        sk = kth_largest_of([&](int x) { return ords[x]; }, (int)M, k, tid, s_hist);
    }
    if (tid >= 64) return;
    double B, unscale;
    query_bound(A, i, q, tid, B, unscale);
    float lbf = A.D_out[(size_t)q * k + k - 1];
    if (have) {
        const double lb = (double)ord2f(sk) * unscale - B;
        float f = (float)lb;
        if ((double)f > lb) f = nextafterf(f, -INFINITY);               // round DOWN: stays a lower bound
        if (f == f && f > lbf) { lbf = f; if (tid == 0) A.D_out[(size_t)q * k + k - 1] = f; }
    }
    const float thr = thr_from_bound((double)lbf, B, unscale);
    if (tid == 0) A.thr[i] = thr;
    if (!A.keep) {
        if (tid == 0) A.cnt[i] = 0u;
        return;
    }
    prune_kept(A, i, thr, tid);
}

// k_select_all: one workgroup per selected query.  The kept rows are first pruned by SCAN score, before any row is
// read: with s_k the k-th largest kept scan score, at least k rows have an exact score >= s_k * unscale - B, and a
// row whose scan score lies more than 2 B (+ one float32 ulp) below s_k cannot reach that -- the survivors are a
// superset of every possible result, typically k + a few.  They are re-scored canonically (float64, sequential in
// k, from the stored rows), bitonic-sorted by (score desc, id asc), the first k written.  status[q] = 0 when the
// kept rows fit the capacity (and there are at least min(k, n) of them); untouched otherwise.
// Launched twice: first with a SMALL LDS footprint (`cap_lds` = 2048 keys: several workgroups per CU -- the common
// case of a few hundred kept rows), then with the full capacity for the queries the first launch had to skip
// (`second`: resolved queries return at once).
template <int SA_ROWS>
__global__ __launch_bounds__(SORT_THREADS) void k_select_all(const ThrArgs A, int cap_pow2, int second) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);             // [cap_pow2] kept rows (scan keys)
    unsigned long long* surv = keys + cap_pow2;                                          // [cap_pow2] survivors, then exact keys
    char* qrow = reinterpret_cast<char*>(surv + cap_pow2);
    __shared__ unsigned s_hist[260];
    __shared__ float s_cut;
    __shared__ unsigned s_keep;
    const int i = blockIdx.x, tid = threadIdx.x;
    const int q = A.qsel[i];
    const int rb = A.d * (A.dtype == DT_F32 ? 4 : 2);
    const int k = A.k;
    const unsigned M = A.cnt[i];
    const long need = (long)k < (long)A.n ? k : A.n;
    if (M > (unsigned)A.cap || (long)M < need) return;                  // overflow (or NaNs): stays unproven
    if (M > (unsigned)cap_pow2 || (second && A.status[q] == 0)) return; // the other launch's share
    for (int v = tid; v < rb / 16; v += SORT_THREADS)
        reinterpret_cast<f32x4*>(qrow)[v] = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(A.Q) + (size_t)q * rb)[v];
    const unsigned long long* ck = A.cand + (size_t)i * A.cap;
    for (int c = tid; c < (int)M; c += SORT_THREADS) keys[c] = ck[c];
    if (tid == 0) { s_cut = -INFINITY; s_keep = 0u; }
    __syncthreads();
    if ((int)M > 2 * k + 64) {                                          // (worth a selection only when there is much to prune)
        const unsigned sk_o = kth_largest_ord(keys, (int)M, k, tid, s_hist);
        if (tid < 64) {
            double B, unscale;
            query_bound(A, i, q, tid, B, unscale);
            if (tid == 0) {
                const double sk = (double)ord2f(sk_o);
                const double c = sk - (2.0 * B + 2.4e-7 * fabs(sk * unscale) + 1e-44) / unscale;
                float f = (float)c;
                if ((double)f > c) f = nextafterf(f, -INFINITY);
                s_cut = f == f ? f : -INFINITY;                         // (NaN bound: keep everything)
            }
        }
        __syncthreads();
    }
    const float cut = s_cut;
    for (int c = tid; c < (int)M; c += SORT_THREADS) {
        const unsigned long long key = keys[c];
        if (key_score(key) >= cut || !(cut > -INFINITY)) surv[atomicAdd(&s_keep, 1u)] = key;
    }
    __syncthreads();
    const int keep = (int)s_keep;                                       // >= k: the k-th largest itself passes the cut
    int K2 = 64;
    while (K2 < keep) K2 <<= 1;
    // Canonical re-score of the survivors, SA_ROWS of them at a time, one thread per row for the strictly sequential
    // float64 chain -- but the rows come in through LDS: the workgroup fetches 128 contiguous bytes of each of the
    // group's rows per step (coalesced: eight lanes a row) and every thread then reads its own row's chunks from the
    // staging tile.  (A thread walking its own 6400-byte row 16 bytes at a time -- round 3's first form -- turned
    // every load into 64 separate line requests per wave: 0.53 ms of a 5.7 ms search at D = 1600, K = 100.)
    char* stage = qrow + ((rb + 15) & ~15);                             // [SA_ROWS][SA_BYTES + 16]
    const int nchunks = rb / 16;
    if (rb < 1024) {                                                    // short rows (a few lines each): a thread per row, all 256 busy
        for (int c = tid; c < K2; c += SORT_THREADS) {
            unsigned long long key = 0ull;
            if (c < keep) {
                const int id = key_id(surv[c]);
                key = make_key((float)rescore_row(qrow, reinterpret_cast<const char*>(A.C) + (size_t)id * rb, nchunks, A.dtype), id);
            }
            keys[c] = key;
        }
    } else
    for (int c0 = 0; c0 < K2; c0 += SA_ROWS) {
        constexpr int PER = SA_ROWS * (SA_BYTES / 16) / SORT_THREADS;   // 16-byte pieces a thread fetches per step
        f32x4 pre[PER];
        auto fetch = [&](int b) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int idx = tid + SORT_THREADS * j, r = idx / (SA_BYTES / 16), ch = idx % (SA_BYTES / 16);
                const int cs = min(c0 + r, keep - 1), vs = min(b + ch, nchunks - 1);      // (clamped: unused copies of valid bytes)
                pre[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(A.C) + (size_t)key_id(surv[cs]) * rb + (size_t)vs * 16);
            }
        };
        double acc = 0.0;
        fetch(0);
        for (int b = 0; b < nchunks; b += SA_BYTES / 16) {
            __syncthreads();                                            // the previous step's tile has been consumed
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int idx = tid + SORT_THREADS * j, r = idx / (SA_BYTES / 16), ch = idx % (SA_BYTES / 16);
                *reinterpret_cast<f32x4*>(stage + r * (SA_BYTES + 16) + ch * 16) = pre[j];
            }
            __syncthreads();
            if (b + SA_BYTES / 16 < nchunks) fetch(b + SA_BYTES / 16);  // in flight under this step's chain
            if (tid < SA_ROWS && c0 + tid < keep) {
#pragma unroll
                for (int i = 0; i < SA_BYTES / 16; ++i)
                    if (b + i < nchunks)
                        acc = dot_chunk(acc, qrow, b + i, *reinterpret_cast<const f32x4*>(stage + tid * (SA_BYTES + 16) + i * 16), A.dtype);
            }
        }
        __syncthreads();                                                // (surv is read by every fetch; keys below aliases nothing)
        if (tid < SA_ROWS && c0 + tid < K2) {
            unsigned long long key = 0ull;
            if (c0 + tid < keep) key = make_key((float)acc, key_id(surv[c0 + tid]));
            keys[c0 + tid] = key;                                       // (the scan keys are no longer needed)
        }
    }
    __syncthreads();
    // Many more survivors than results (a query whose k-th neighbour sits in a group of thousands of identical rows: config
    // C3's one-click prefix sessions): a bitonic sort of all K2 exact keys -- 91 barrier stages at K2 = 8192 -- was 0.3 of
    // the 0.5 ms such a workgroup took.  The k best are SELECTED first (two radix descents: the k-th largest score ordinal,
    // then, inside its tie group, the id word that completes the count -- keys are unique, so exactly k lie at or above the
    // resulting key) and only they are sorted.
    int Ks = 64;
    while (Ks < k) Ks <<= 1;
    const unsigned long long* outk = keys;
    if (keep > 2 * Ks) {
        const unsigned long long T = kth_largest_key(keys, keep, k, tid, s_hist, &s_keep);
        __syncthreads();
        if (tid == 0) s_keep = 0u;
        __syncthreads();
        for (int x = tid; x < keep; x += SORT_THREADS) {                // (surv: the survivors' scan keys are no longer needed)
            const unsigned long long kx = keys[x];
            if (kx >= T) surv[atomicAdd(&s_keep, 1u)] = kx;
        }
        __syncthreads();
        for (int x = (int)s_keep + tid; x < Ks; x += SORT_THREADS) surv[x] = 0ull;
        __syncthreads();
        sort_desc(surv, Ks, tid);
        outk = surv;
    } else {
        sort_desc(keys, K2, tid);
    }
    float* Dq = A.D_out + (size_t)q * k;
    long* Iq = A.I_out + (size_t)q * k;
    for (int j = tid; j < k; j += SORT_THREADS) {
        if (j < keep) { Dq[j] = key_score(outk[j]); Iq[j] = (long)key_id(outk[j]) + A.id_offset; }
        else { Dq[j] = -3.4028234663852886e38f; Iq[j] = -1; }
    }
    if (tid == 0) A.status[q] = 0;
}

// ------------------------------------------------------------------------------------------
// k-way merge of per-shard results (after the RCCL all-gather): [shards][nq][k] -> [nq][k] by
// (score desc, id asc); ids < 0 are padding.  One thread per query (k*shards is tiny).
__global__ void k_topk_merge(const float* __restrict__ D_in, long d_stride, const long* __restrict__ I_in,
                             long i_stride, int shards, int nq, int k, float* __restrict__ D_out,
                             long* __restrict__ I_out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    int pos[64];
    for (int s = 0; s < shards; ++s) pos[s] = 0;
    for (int o = 0; o < k; ++o) {
        int bs = -1; float bd = 0.f; long bi = 0;
        for (int s = 0; s < shards; ++s) {
            if (pos[s] >= k) continue;
            const size_t a = (size_t)q * k + pos[s];
            const long id = I_in[(size_t)s * i_stride + a];
            if (id < 0) { pos[s] = k; continue; }
            const float dd = D_in[(size_t)s * d_stride + a];
            if (bs < 0 || dd > bd || (dd == bd && id < bi)) { bs = s; bd = dd; bi = id; }
        }
        if (bs < 0) { D_out[(size_t)q * k + o] = -3.4028234663852886e38f; I_out[(size_t)q * k + o] = -1; }
        else { D_out[(size_t)q * k + o] = bd; I_out[(size_t)q * k + o] = bi; ++pos[bs]; }
    }
}

// ------------------------------------------------------------------------------ host side
int launch_select(const SelectArgs& a, hipStream_t st) {
    const int rb = a.d * elem_bytes(a.dtype);
    const int dev = current_device();
    // the wave-per-query kernel serves the K2 <= 16 regime (class maxima + bootstrap: a few hundred
    // candidates per query); beyond it the lists run without the bootstrap and fill up (thousands of
    // candidates, more than a wave stages), which is the sort kernel's job
    if (a.K2 <= KP) {
        const size_t per_wave = (((size_t)FS_CAP * 8 + FS_K2 * 16 + 16 + rb) + 15) & ~(size_t)15;
        const size_t lds = 4 * per_wave;
        static bool done[MAX_DEVICES] = {};
        if (!done[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_fast),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            done[dev] = true;
        }
        hipLaunchKernelGGL(k_select_fast, dim3((unsigned)((a.nq + 3) / 4)), dim3(256), lds, st, a);
        return check_launch("k_select_fast");
    }
    if (a.K2 > SEL_MAX_K2) { set_error("select: K2 %d > %d", a.K2, SEL_MAX_K2); return SSS_EINVAL; }
    int cap_pow2 = 64;
    while (cap_pow2 < a.cap || cap_pow2 < a.K2) cap_pow2 <<= 1;
    const size_t lds = (size_t)cap_pow2 * 8 + SEL_MAX_K2 * 8 + rb;
    if (lds > 150 * 1024) { set_error("select: candidate capacity %d too large", a.cap); return SSS_EINVAL; }
    static bool done2[MAX_DEVICES] = {};
    if (!done2[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_sort),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);     // (its static words take ~1.1 KB)
        done2[dev] = true;
    }
    hipLaunchKernelGGL(k_select_sort, dim3((unsigned)a.nq), dim3(SORT_THREADS), lds, st, a, cap_pow2);
    return check_launch("k_select_sort");
}

int launch_thr_prepare(const ThrArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_thr_prepare, dim3((unsigned)((a.nsel + 3) / 4)), dim3(256), 0, st, a);
    return check_launch("k_thr_prepare");
}

int launch_long_setup(const ThrArgs& a, int* qsel, void* qimg, hipStream_t st) {
    hipLaunchKernelGGL(k_long_setup, dim3((unsigned)((a.nsel + 3) / 4)), dim3(256), 0, st, a, qsel, reinterpret_cast<_Float16*>(qimg));
    return check_launch("k_long_setup");
}

int launch_bound_prepare(const ThrArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_bound_prepare, dim3((unsigned)a.nsel), dim3(SORT_THREADS), (size_t)a.cap * 4, st, a);     // (cap <= 8192: 32 KB)
    return check_launch("k_bound_prepare");
}

int launch_select_all(const ThrArgs& a, hipStream_t st) {
    const int rb = a.d * elem_bytes(a.dtype);
    int cap_pow2 = 64;
    while (cap_pow2 < a.cap) cap_pow2 <<= 1;
    // the re-score's staging tile (rows of 1024 bytes and more; shorter rows are walked by a thread each)
    const size_t stage = rb < 1024 ? 0 : (size_t)SA_ROWS_FULL * (SA_BYTES + 16);
    const size_t stage_small = rb < 1024 ? 0 : (size_t)SA_ROWS_SMALL * (SA_BYTES + 16);
    const size_t lds = 2 * (size_t)cap_pow2 * 8 + ((rb + 15) & ~15) + stage;   // kept keys + survivors + the query row + the tile
    if (lds > 156 * 1024) { set_error("select_all: candidate capacity %d / row of %d bytes too large", a.cap, rb); return SSS_EINVAL; }
    static bool done[MAX_DEVICES] = {};
    const int dev = current_device();
    if (!done[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_all<SA_ROWS_SMALL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_all<SA_ROWS_FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        done[dev] = true;
    }
    const int small = cap_pow2 < 2048 ? cap_pow2 : 2048;
    // (k <= 64: k + a few dozen survivors fit one 128-row group, and a 256-row group would fetch twice the clamped copies)
    if (a.k > 64) hipLaunchKernelGGL(k_select_all<SA_ROWS_SMALL>, dim3((unsigned)a.nsel), dim3(SORT_THREADS), 2 * (size_t)small * 8 + ((rb + 15) & ~15) + stage_small, st, a, small, 0);
    else hipLaunchKernelGGL(k_select_all<SA_ROWS_FULL>, dim3((unsigned)a.nsel), dim3(SORT_THREADS), 2 * (size_t)small * 8 + ((rb + 15) & ~15) + stage, st, a, small, 0);
    if (small < cap_pow2) hipLaunchKernelGGL(k_select_all<SA_ROWS_FULL>, dim3((unsigned)a.nsel), dim3(SORT_THREADS), lds, st, a, cap_pow2, 1);
    return check_launch("k_select_all");
}

int topk_merge(const float* D_in, long d_stride, const long* I_in, long i_stride, int shards, long nq, int k,
               float* D_out, long* I_out, hipStream_t st) {
    if (shards < 1 || shards > 64 || nq <= 0 || k <= 0 || d_stride < nq * k || i_stride < nq * k) {
        set_error("topk_merge: bad arguments");
        return SSS_EINVAL;
    }
    hipLaunchKernelGGL(k_topk_merge, dim3((unsigned)((nq + 127) / 128)), dim3(128), 0, st, D_in, d_stride, I_in,
                       i_stride, shards, (int)nq, k, D_out, I_out);
    return check_launch("k_topk_merge");
}

}  // namespace sss

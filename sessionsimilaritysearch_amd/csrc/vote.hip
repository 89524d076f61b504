// Neighbour-weighted item vote (gfx950) -- the step right after the search in the reference's
// `get_prediction_by_knn` (test_amazon_filterd.py:59-78; SURVEY.md section 8(f) row 2, BASELINE
// config C3 "aggregated top-10"):
//     D, I = index.search(emb, sample_size)
//     every item of neighbour session I[j] gets weight D[j]; weights are summed per item (in
//     float64, in neighbour order: an int64 array times a float32 scalar is float64, added to a
//     Python accumulator); items are ranked by weight, ties keep first-seen order (stable sort);
//     the K heaviest are returned.
// One workgroup per query.  The neighbours' item lists (CSR session -> distinct items, the
// `product.x` of each indexed graph) are expanded into LDS as 64-bit keys
//     item << 32 | position-in-concatenation << 16 | neighbour j
// bitonic-sorted, so equal items become contiguous IN NEIGHBOUR ORDER; a segment's weight is the
// sequential float64 sum over it -- bit-identical to the reference's accumulation.  The top K
// segments by (weight desc, first position asc) come out of K rounds of block-wide arg-max.
// HBM-bound integer/byte work: no matrix cores.
#include "sss_common.h"

namespace sss {

constexpr int VT = 256;                 // threads per query
constexpr int VOTE_MAX_ENTRIES = 16384; // expanded (neighbour, item) pairs per query (128 KiB of LDS)
constexpr int VOTE_SMALL_ENTRIES = 4096; // ... the first launch's share (32 KiB: four workgroups per CU)
constexpr unsigned long long TAKEN = 1ull << 15;

struct Best { double w; int seq; int pos; };
__device__ __forceinline__ bool better(const Best& a, const Best& b) {   // a strictly ahead of b
    if (a.pos < 0) return false;
    if (b.pos < 0) return true;
    return a.w > b.w || (a.w == b.w && a.seq < b.seq);
}

__global__ __launch_bounds__(VT) void k_item_vote(const float* __restrict__ D, const long* __restrict__ I, int S,
                                                  const long* __restrict__ items_ptr, const int* __restrict__ items,
                                                  long id_offset, long n_sessions, int K, int cap,
                                                  long* __restrict__ out_items, double* __restrict__ out_w,
                                                  int* __restrict__ status, int cap_full, int second) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];     // [cap]
    __shared__ int s_scan[VT];
    __shared__ int s_total;
    __shared__ double s_bw[VT / 64];
    __shared__ int s_bs[VT / 64], s_bp[VT / 64];
    __shared__ int s_winner_pos;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (second && status[q] != 2) return;                    // (whole workgroup) resolved by the first launch
    const float* Dq = D + (size_t)q * S;
    const long* Iq = I + (size_t)q * S;
    long* oi = out_items + (size_t)q * K;

    // ---- per-neighbour item counts -> exclusive offsets (block scan, S <= 2 * VT ... any S by chunks)
    int total = 0;
    // thread t owns neighbours [t * per, (t + 1) * per)
    const int per = (S + VT - 1) / VT;
    int mine = 0;
    for (int j = tid * per; j < min(S, (tid + 1) * per); ++j) {
        const long sid = Iq[j] - id_offset;
        if (Iq[j] >= 0 && sid >= 0 && sid < n_sessions) mine += (int)(items_ptr[sid + 1] - items_ptr[sid]);
    }
    s_scan[tid] = mine;
    __syncthreads();
    for (int o = 1; o < VT; o <<= 1) {                       // Hillis-Steele inclusive scan
        const int v = tid >= o ? s_scan[tid - o] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    if (tid == VT - 1) s_total = s_scan[tid];
    int off = s_scan[tid] - mine;
    __syncthreads();
    total = s_total;
    if (total > cap && total <= cap_full) {                  // first launch (small LDS footprint): the second one's share
        if (tid == 0) status[q] = 2;
        return;
    }
    if (total > cap) {                                       // does not fit the LDS budget: tell the caller
        if (tid == 0) status[q] = 1;
        for (int i = tid; i < K; i += VT) { oi[i] = -1; if (out_w) out_w[(size_t)q * K + i] = 0.0; }
        return;
    }
    int M2 = 64;
    while (M2 < total) M2 <<= 1;
    for (int i = tid; i < M2; i += VT) keys[i] = ~0ull;      // padding sorts last
    __syncthreads();
    for (int j = tid * per; j < min(S, (tid + 1) * per); ++j) {
        const long sid = Iq[j] - id_offset;
        if (!(Iq[j] >= 0 && sid >= 0 && sid < n_sessions)) continue;
        for (long e = items_ptr[sid]; e < items_ptr[sid + 1]; ++e, ++off)
            keys[off] = ((unsigned long long)(unsigned)items[e] << 32) | ((unsigned long long)off << 16) | (unsigned)j;
    }
    __syncthreads();
    // ---- bitonic sort, ascending: equal items end up contiguous in position (= neighbour) order
    for (int kk = 2; kk <= M2; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < M2; i += VT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool asc = (i & kk) == 0;
                    if (asc ? a > b : a < b) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    // ---- a thread owns the segment heads at positions tid, tid + VT, ...; weight = sequential f64 sum
    auto item_of = [&](int p) { return (unsigned)(keys[p] >> 32); };
    auto scan_mine = [&]() -> Best {
        Best best{0.0, 0, -1};
        for (int p = tid; p < total; p += VT) {
            const unsigned long long kp = keys[p];
            if (kp & TAKEN) continue;
            if (p > 0 && item_of(p - 1) == (unsigned)(kp >> 32)) continue;        // not a head
            double w = 0.0;
            for (int t = p; t < total && item_of(t) == (unsigned)(kp >> 32); ++t)
                w += (double)Dq[(int)(keys[t] & 0x7FFFu)];
            Best c{w, (int)((kp >> 16) & 0xFFFFu), p};
            if (better(c, best)) best = c;
        }
        return best;
    };
    Best mineb = scan_mine();
    for (int r = 0; r < K; ++r) {
        Best b = mineb;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Best other{__shfl_xor(b.w, o), __shfl_xor(b.seq, o), __shfl_xor(b.pos, o)};
            if (better(other, b)) b = other;
        }
        if (lane == 0) { s_bw[wv] = b.w; s_bs[wv] = b.seq; s_bp[wv] = b.pos; }
        __syncthreads();
        if (tid == 0) {
            Best g{s_bw[0], s_bs[0], s_bp[0]};
            for (int w = 1; w < VT / 64; ++w) {
                Best c{s_bw[w], s_bs[w], s_bp[w]};
                if (better(c, g)) g = c;
            }
            s_winner_pos = g.pos;
            if (g.pos >= 0) {
                oi[r] = (long)item_of(g.pos);
                if (out_w) out_w[(size_t)q * K + r] = g.w;
                keys[g.pos] |= TAKEN;
            } else {
                oi[r] = -1;
                if (out_w) out_w[(size_t)q * K + r] = 0.0;
            }
        }
        __syncthreads();
        const int wp = s_winner_pos;
        if (wp >= 0 && (wp % VT) == tid) mineb = scan_mine();     // only the owner's candidates changed
    }
    if (tid == 0) status[q] = 0;
}

size_t item_vote_lds_bytes(int cap) { return (size_t)cap * 8; }

int item_vote(const float* D, const long* I, long nq, int S, const long* items_ptr, const int* items, long id_offset,
              long n_sessions, int K, long* out_items, double* out_w, int* status, hipStream_t st) {
    if (nq <= 0 || S <= 0 || S > 32767 || K <= 0 || n_sessions < 0) {
        set_error("item_vote: need nq, K > 0 and 0 < sample_size < 32768");
        return SSS_EINVAL;
    }
    // Launched twice, like k_select_all: first with a SMALL LDS footprint (4096 expanded pairs = 32 KB: four workgroups
    // per CU -- the usual few thousand pairs of a query; with the full 128 KB every workgroup had a CU to itself and
    // 1024 queries ran in four rounds: 1.0 ms of config C3's step), then with the full capacity for the queries the first
    // launch had to leave (status 2; the others return at once).
    const int cap = VOTE_MAX_ENTRIES, cap_small = VOTE_SMALL_ENTRIES;
    static bool done[64] = {};
    int dev = 0; (void)hipGetDevice(&dev); if (dev < 0 || dev >= 64) dev = 0;
    if (!done[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_item_vote), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)item_vote_lds_bytes(cap));
        done[dev] = true;
    }
    hipLaunchKernelGGL(k_item_vote, dim3((unsigned)nq), dim3(VT), item_vote_lds_bytes(cap_small), st, D, I, S, items_ptr, items,
                       id_offset, n_sessions, K, cap_small, out_items, out_w, status, cap, 0);
    hipLaunchKernelGGL(k_item_vote, dim3((unsigned)nq), dim3(VT), item_vote_lds_bytes(cap), st, D, I, S, items_ptr, items,
                       id_offset, n_sessions, K, cap, out_items, out_w, status, cap, 1);
    return check_launch("k_item_vote");
}

}  // namespace sss

// Action table -> batched session graphs in CSR-by-target form, on device (gfx950).
//
// Structural part of the reference's `sequence_to_graph` (util_amazon_filtered.py:98-230: query
// nodes :7-22, distinct items :128-142, click edges :180-195, de-duplicated transitions with
// counts :199-218, pos ids :77-85) + PyG `Batch.from_data_list` (test_amazon_filterd.py:485-488;
// SURVEY.md Appendix A.6), producing directly what the encoder kernels read: three CSR-by-target
// adjacencies, the expanded-node index arrays of the pooling and the per-graph segment pointers
// (SURVEY.md section 8(f) row 1).  Integer work only; bit-exact against oracle/graph_ref.py.
//
// A session has at most 64 actions (the reference's positional table holds 20,
// config.py:5): ONE WAVE PER SESSION, lane t = action t; every "have I seen this item / pair
// before" question is a loop of v_readlane broadcasts over the session, ranks are popcounts of
// ballots.  Two sweeps: k_session_counts (5 counts per session) -> exclusive scans (k_scan_*)
// -> k_session_fill (recomputes the lane state and writes every array at its final offset).
// Distinct items keep first-occurrence order (the reference's `list(set(...))` is hash order;
// documented deviation, sessions.py).
#include "sss_common.h"
#include "kargs.h"

namespace sss {

constexpr int NC = 5;            // per-session counts: query nodes, product nodes, expanded product rows, click edges, unique transitions
constexpr int SCAN_BLOCK = 1024;

struct LaneState {
    int n;                       // actions in the session
    bool srch, clk;
    int item;
    unsigned long long sm, cm, fm, um;   // ballots: searches, clicks, first occurrences, unique transitions
    int firstlane, earlier_same, cnt_same, pidx;
    bool isfirst, haspair, isuniq;
    int a, b, paircount;
    unsigned long long lt;       // lanes below this one
};

__device__ __forceinline__ LaneState lane_state(const long* __restrict__ sess_ptr, const unsigned char* __restrict__ is_search,
                                                const long* __restrict__ item_id, long s, int t) {
    LaneState L;
    const long a0 = sess_ptr[s];
    L.n = (int)(sess_ptr[s + 1] - a0);
    const bool valid = t < L.n;
    L.srch = valid && is_search[a0 + (valid ? t : 0)] != 0;
    L.clk = valid && !L.srch;
    L.item = L.clk ? (int)item_id[a0 + t] : -1;
    L.sm = __builtin_amdgcn_ballot_w64(L.srch);
    L.cm = __builtin_amdgcn_ballot_w64(L.clk);
    L.lt = t == 0 ? 0ull : (~0ull >> (64 - t));
    L.firstlane = t; L.earlier_same = 0; L.cnt_same = 0;
    for (int u = 0; u < L.n; ++u) {                              // uniform loop: broadcast action u
        const int iu = __builtin_amdgcn_readlane(L.item, u);
        const bool same = L.clk && iu == L.item;                 // iu == -1 for searches: never equals a click's item
        if (same) { ++L.cnt_same; if (u < t) { ++L.earlier_same; if (u < L.firstlane) L.firstlane = u; } }
    }
    L.isfirst = L.clk && L.firstlane == t;
    L.fm = __builtin_amdgcn_ballot_w64(L.isfirst);
    L.pidx = L.clk ? __builtin_popcountll(L.fm & (L.firstlane == 0 ? 0ull : (~0ull >> (64 - L.firstlane)))) : -1;
    // transition to the next click of the session
    const unsigned long long after = t >= 63 ? 0ull : (L.cm & ~((2ull << t) - 1ull));
    L.haspair = L.clk && after != 0;
    const int nt = L.haspair ? __builtin_ctzll(after) : t;
    L.a = L.pidx;
    L.b = __shfl(L.pidx, nt);
    if (!L.haspair) { L.a = -1; L.b = -2; }
    L.paircount = 0;
    bool seen_before = false;
    for (int u = 0; u < L.n; ++u) {
        const int au = __builtin_amdgcn_readlane(L.a, u), bu = __builtin_amdgcn_readlane(L.b, u);
        const bool same = L.haspair && au == L.a && bu == L.b;
        if (same) { ++L.paircount; if (u < t) seen_before = true; }
    }
    L.isuniq = L.haspair && !seen_before;
    L.um = __builtin_amdgcn_ballot_w64(L.isuniq);
    return L;
}

__global__ __launch_bounds__(256) void k_session_counts(const long* __restrict__ sess_ptr,
                                                        const unsigned char* __restrict__ is_search,
                                                        const long* __restrict__ item_id, long S,
                                                        int* __restrict__ counts, int* __restrict__ err) {
    const long s = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int t = threadIdx.x & 63;
    if (s >= S) return;
    const long n = sess_ptr[s + 1] - sess_ptr[s];
    if (n > 64 || n < 0) {                                        // whole wave
        if (t == 0) { atomicOr(err, 1); for (int c = 0; c < NC; ++c) counts[(long)c * (S + 1) + s] = 0; }
        return;
    }
    const LaneState L = lane_state(sess_ptr, is_search, item_id, s, t);
    if (t == 0) {
        const int nclk = __builtin_popcountll(L.cm), nd = __builtin_popcountll(L.fm);
        counts[0 * (S + 1) + s] = 1 + __builtin_popcountll(L.sm);
        counts[1 * (S + 1) + s] = nd > 0 ? nd : 1;                // a session without clicks gets the "unknown item" node
        counts[2 * (S + 1) + s] = nclk > 0 ? nclk : 1;            // ... with one expanded row (pos id 0)
        counts[3 * (S + 1) + s] = nclk;
        counts[4 * (S + 1) + s] = __builtin_popcountll(L.um);
    }
}

// ---- exclusive scan of NC arrays [S] -> same storage, totals at [S]; three tiny kernels
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_local(int* __restrict__ data, long S, int* __restrict__ bsum, int nblk) {
    __shared__ int sh[SCAN_BLOCK];
    int* d = data + (long)blockIdx.y * (S + 1);
    const long i = (long)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const int v = i < S ? d[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < SCAN_BLOCK; o <<= 1) {
        const int add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    if (i < S) d[i] = sh[threadIdx.x] - v;                        // exclusive, block-local
    if (threadIdx.x == SCAN_BLOCK - 1) bsum[blockIdx.y * nblk + blockIdx.x] = sh[threadIdx.x];
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_tops(int* __restrict__ bsum, int nblk, int* __restrict__ data, long S) {
    __shared__ int sh[SCAN_BLOCK];
    __shared__ int carry;
    int* b = bsum + blockIdx.x * nblk;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += SCAN_BLOCK) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? b[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < SCAN_BLOCK; o <<= 1) {
            const int add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nblk) b[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == SCAN_BLOCK - 1) carry += sh[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) data[(long)blockIdx.x * (S + 1) + S] = carry;     // grand total
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_add(int* __restrict__ data, long S, const int* __restrict__ bsum, int nblk) {
    const long i = (long)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    if (i < S) data[(long)blockIdx.y * (S + 1) + i] += bsum[blockIdx.y * nblk + blockIdx.x];
}


__global__ __launch_bounds__(256) void k_session_fill(const long* __restrict__ sess_ptr,
                                                      const unsigned char* __restrict__ is_search,
                                                      const long* __restrict__ item_id, const long* __restrict__ query_tok,
                                                      long S, const int* __restrict__ bases, const GraphOut O) {
    const long s = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int t = threadIdx.x & 63;
    if (s >= S) return;
    const long nn = sess_ptr[s + 1] - sess_ptr[s];
    if (nn > 64 || nn < 0) return;                               // flagged by k_session_counts
    const LaneState L = lane_state(sess_ptr, is_search, item_id, s, t);
    const int qb = bases[0 * (S + 1) + s], pb = bases[1 * (S + 1) + s], xb = bases[2 * (S + 1) + s];
    const int eb = bases[3 * (S + 1) + s], ppb = bases[4 * (S + 1) + s];
    const int Xp = bases[2 * (S + 1) + S], Nq = bases[0 * (S + 1) + S];   // expanded product rows, query nodes in the batch
    const int n = L.n;
    const long a0 = sess_ptr[s];
    // ---- query nodes: the root, then one per search (util_amazon_filtered.py:7-22)
    const int lq = __builtin_popcountll(L.sm & L.lt);            // searches before this action
    if (t == 0) {
        O.q_x[qb] = 0; O.q_pos[qb] = n; O.q_batch[qb] = s;
        O.rowptr_pq[qb] = eb;
        O.src_row[Xp + qb] = qb; O.pos_id[Xp + qb] = n;
    }
    if (L.srch) {
        const int k = qb + lq + 1;
        O.q_x[k] = query_tok[a0 + t]; O.q_pos[k] = n - (t + 1); O.q_batch[k] = s;
        O.rowptr_pq[k] = eb + __builtin_popcountll(L.cm & L.lt);
        O.src_row[Xp + k] = k; O.pos_id[Xp + k] = n - (t + 1);
    }
    // ---- product nodes (first occurrences), their edge / row offsets
    int node_off = 0, pp_before = 0;
    for (int u = 0; u < n; ++u) {
        const bool fu = (L.fm >> u) & 1;
        const int pu = __builtin_amdgcn_readlane(L.pidx, u), cu = __builtin_amdgcn_readlane(L.cnt_same, u);
        if (fu && L.clk && pu < L.pidx) node_off += cu;          // clicks of the nodes ordered before mine
        const bool uu = (L.um >> u) & 1;
        const int bu = __builtin_amdgcn_readlane(L.b, u);
        if (uu && L.clk && bu < L.pidx) ++pp_before;             // unique transitions into earlier nodes
    }
    const int nd = __builtin_popcountll(L.fm);
    if (L.isfirst) {
        const int node = pb + L.pidx;
        O.p_x[node] = L.item; O.p_cnt[node] = L.cnt_same; O.p_batch[node] = s;
        O.rowptr_qp[node] = eb + node_off;
        O.rowptr_pp[node] = ppb + pp_before;
    }
    if (nd == 0 && t == 0) {                                     // no click at all: the "unknown item" node (:132-135)
        O.p_x[pb] = 0; O.p_cnt[pb] = 1; O.p_batch[pb] = s;
        O.rowptr_qp[pb] = eb; O.rowptr_pp[pb] = ppb;
        O.src_row[xb] = pb; O.pos_id[xb] = 0;
    }
    // ---- clicks: grouped by product node in time order (expanded rows, CSR qp), in action order (CSR pq)
    if (L.clk) {
        const int g = node_off + L.earlier_same;
        O.src_row[xb + g] = pb + L.pidx;
        O.pos_id[xb + g] = n - t;                                // len(seq) - j  (:82)
        O.col_qp[eb + g] = qb + lq;                              // most recent query node (:183-191)
        O.col_pq[eb + __builtin_popcountll(L.cm & L.lt)] = pb + L.pidx;
    }
    // ---- unique transitions a -> b, grouped by target b in first-occurrence order (:199-218)
    int pos = 0;
    for (int u = 0; u < n; ++u) {
        const bool uu = (L.um >> u) & 1;
        const int bu = __builtin_amdgcn_readlane(L.b, u);
        if (uu && (bu < L.b || (bu == L.b && u < t))) ++pos;
    }
    if (L.isuniq) {
        O.col_pp[ppb + pos] = pb + L.a;
        O.w_pp[ppb + pos] = (float)L.paircount;
    }
    if (s == S - 1 && t == 0) {                                  // closing entries of the three row pointers
        const int Np = bases[1 * (S + 1) + S], E = bases[3 * (S + 1) + S], Epp = bases[4 * (S + 1) + S];
        O.rowptr_qp[Np] = E; O.rowptr_pq[Nq] = E; O.rowptr_pp[Np] = Epp;
    }
}

// ------------------------------------------------------------------------------ host launchers
size_t graph_scratch_ints(long S) { return (size_t)NC * ((S + SCAN_BLOCK - 1) / SCAN_BLOCK + 1); }

// bases: int32 [NC][S+1] (out: exclusive scans, totals at [S]); scratch: graph_scratch_ints(S) ints; err: int32 [1]
int graph_counts(const long* sess_ptr, const unsigned char* is_search, const long* item_id, long S, int* bases, int* scratch,
                 int* err, hipStream_t st) {
    if (S <= 0) { set_error("graph_counts: need at least one session"); return SSS_EINVAL; }
    if (hipMemsetAsync(err, 0, sizeof(int), st) != hipSuccess) { set_error("graph_counts: memset failed"); return SSS_EHIP; }
    const unsigned nb = (unsigned)((S * 64 + 255) / 256);
    hipLaunchKernelGGL(k_session_counts, dim3(nb), dim3(256), 0, st, sess_ptr, is_search, item_id, S, bases, err);
    int rc = check_launch("k_session_counts");
    if (rc) return rc;
    const int nblk = (int)((S + SCAN_BLOCK - 1) / SCAN_BLOCK);
    hipLaunchKernelGGL(k_scan_local, dim3(nblk, NC), dim3(SCAN_BLOCK), 0, st, bases, S, scratch, nblk);
    hipLaunchKernelGGL(k_scan_tops, dim3(NC), dim3(SCAN_BLOCK), 0, st, scratch, nblk, bases, S);
    hipLaunchKernelGGL(k_scan_add, dim3(nblk, NC), dim3(SCAN_BLOCK), 0, st, bases, S, scratch, nblk);
    return check_launch("k_scan");
}

int graph_fill(const long* sess_ptr, const unsigned char* is_search, const long* item_id, const long* query_tok, long S,
               const int* bases, const GraphOut& out, hipStream_t st) {
    if (S <= 0) { set_error("graph_fill: need at least one session"); return SSS_EINVAL; }
    const unsigned nb = (unsigned)((S * 64 + 255) / 256);
    hipLaunchKernelGGL(k_session_fill, dim3(nb), dim3(256), 0, st, sess_ptr, is_search, item_id, query_tok, S, bases, out);
    return check_launch("k_session_fill");
}

}  // namespace sss

// Query x corpus inner-product scoring with fused running top-k (gfx950 / CDNA4).
//
// Replaces what the reference asks of faiss at test_amazon_filterd.py:578
// (`D, I = index.search(normalize(emb), K)`, faiss.IndexFlatIP).
//
// k_ip_topk_f32<D>  -- the dominant kernel (MFMA-bound, DESIGN.md "scoring kernel"):
//   * one workgroup = 8 waves (2 per SIMD) = 256 queries x one contiguous corpus split;
//   * each wave keeps its 32 queries resident in D/2 VGPRs as the B operand of
//     v_mfma_f32_32x32x2_f32 (exact f32 fma chain), so the query tile is read from HBM once;
//   * corpus rows stream HBM -> LDS in 64-row tiles with global_load_lds_dwordx4 (no VGPR
//     staging), double buffered, 16-byte chunks XOR-swizzled on the SOURCE address so the
//     ds_read_b128 fragment reads are bank-conflict free;
//   * the score matrix is never written: each lane owns one query column of the 32x32
//     accumulator and keeps a sorted top-KP list (scores + row ids) in registers; a score
//     enters only if it beats the lane's current KP-th best (one v_max3 tree + one compare per
//     tile in the steady state).
// k_select_rescore -- per query: merge the per-(split, half-wave) lists, take the best
//   k+slack by float32 score, re-score those in float64 in the canonical sequential order and
//   emit (score desc, id asc); also emits the per-query "proven exact" status.
#include "sss_common.h"

namespace sss {

constexpr int KP = 16;          // per-lane list length (register resident)
constexpr int TILE_ROWS = 64;   // corpus rows per LDS tile
constexpr int WG_QUERIES = 256; // queries per workgroup (8 waves x 32)

typedef const float __attribute__((address_space(1)))* gptr_f32;
typedef float __attribute__((address_space(3)))* lptr_f32;
typedef char __attribute__((address_space(3)))* lptr_c;

// Sorted (descending) insert of (x, id) into a register list; lanes whose x does not beat
// their list tail fall through untouched.  Strict '>' keeps equal scores in arrival (= id)
// order because every lane sees its rows in ascending id order.
template <int N>
__device__ __forceinline__ void list_insert(float (&ls)[N], int (&li)[N], float x, int id) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool c = x > ls[i];
        const float ns = c ? x : ls[i];
        const int ni = c ? id : li[i];
        x = c ? ls[i] : x;
        id = c ? li[i] : id;
        ls[i] = ns;
        li[i] = ni;
    }
}

template <int D>
__global__ __launch_bounds__(512, 2) void k_ip_topk_f32(
    const float* __restrict__ Q, int nq, const float* __restrict__ C, int n, int rows_per_split,
    int S, int G, float* __restrict__ cand_s, int* __restrict__ cand_i) {
    constexpr int CH = D / 4;                       // 16-byte chunks per row
    constexpr int TILE_BYTES = TILE_ROWS * D * 4;
    constexpr int LOADS_PER_WAVE = CH / 8;          // glds wave-instructions per wave per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware remap: blocks b and b+8 share an XCD (and its L2); the G query groups that
    // stream the same corpus split are given consecutive slots of ONE XCD so the split is
    // fetched from HBM once and re-read from that L2.  Speed only, never correctness.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int split = xcd * (S >> 3) + slot / G;
    const int g = slot % G;

    // ---- resident queries: lane (r, h) holds Q[q][8u + 4h + i] in qreg[4u + i]
    const int q_local = wave * 32 + r;
    const int q_glob = g * WG_QUERIES + q_local;
    const int q_ld = q_glob < nq ? q_glob : nq - 1;
    float qreg[D / 2];
    {
        const float4* qp = reinterpret_cast<const float4*>(Q + (size_t)q_ld * D) + h;
#pragma unroll
        for (int u = 0; u < D / 8; ++u) {
            const float4 v = qp[2 * u];
            qreg[4 * u + 0] = v.x; qreg[4 * u + 1] = v.y;
            qreg[4 * u + 2] = v.z; qreg[4 * u + 3] = v.w;
        }
        // retire the query loads HERE: otherwise hipcc sinks their counted vmcnt waits into the
        // tile loop, where they would also wait on the (uncounted) LDS-DMA of the next tile.
#pragma unroll
        for (int t = 0; t < D / 2; ++t) asm volatile("" : "+v"(qreg[t]));
    }

    float ls[KP];
    int li[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) { ls[i] = -INFINITY; li[i] = -1; }
    float pend_s = -INFINITY;   // one parked candidate per lane (see the epilogue)
    int pend_i = -1;

    const long row_lo = (long)split * rows_per_split;
    long row_hi = row_lo + rows_per_split;
    if (row_hi > n) row_hi = n;
    const int ntiles = row_lo < row_hi ? (int)((row_hi - row_lo + TILE_ROWS - 1) / TILE_ROWS) : 0;

    // LDS-DMA staging (global_load_lds_dwordx4, 1 KiB per wave-instruction).  Written as inline
    // asm so hipcc neither counts it nor drains vmcnt(0) at the next ds_read: the next tile
    // stays in flight under this tile's MFMAs and is retired by the explicit vmcnt(0) that
    // precedes the barrier at the end of the iteration (cdna_hip_programming.md section 5.7).
    const unsigned lds_base = (unsigned)(unsigned long)(lptr_c)smem;
    auto stage = [&](int buf, long row0) {
#pragma unroll
        for (int i = 0; i < LOADS_PER_WAVE; ++i) {
            const int instr = wave * LOADS_PER_WAVE + i;      // wave-uniform
            const int p = instr * 64 + lane;                   // 16-byte slot inside the tile
            const int tr = p / CH, sc = p % CH;
            const int c = sc ^ (tr & 15);                      // source-side swizzle
            long grow = row0 + tr;
            if (grow > (long)n - 1) grow = (long)n - 1;        // clamp; masked in the epilogue
            const float* src = C + (size_t)grow * D + c * 4;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + buf * TILE_BYTES + instr * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };

    // per-lane LDS read offset (bytes) of chunk (2u + h) of row r, before the constant part
    const int x = h ^ (r & 15);

    if (ntiles > 0) stage(0, row_lo);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) stage(buf ^ 1, row_lo + (long)(t + 1) * TILE_ROWS);

        const char* tile = smem + buf * TILE_BYTES;
        f32x16 acc0 = {0}, acc1 = {0};
        auto lda = [&](int u, int mb) -> float4 {
            const int c = (2 * u) ^ x;                          // == (2u + h) ^ (r & 15)
            return *reinterpret_cast<const float4*>(tile + ((r + 32 * mb) * CH + c) * 16);
        };
        float4 a0 = lda(0, 0), a1 = lda(0, 1);
#pragma unroll
        for (int u = 0; u < D / 8; ++u) {
            float4 n0 = a0, n1 = a1;
            if (u + 1 < D / 8) { n0 = lda(u + 1, 0); n1 = lda(u + 1, 1); }   // one k-group ahead
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE this group's MFMAs
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, qreg[4 * u + 0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, qreg[4 * u + 0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, qreg[4 * u + 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, qreg[4 * u + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, qreg[4 * u + 2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, qreg[4 * u + 2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, qreg[4 * u + 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, qreg[4 * u + 3], acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0; a1 = n1;
        }

        // ---- fused top-k epilogue.  acc[j] is (corpus row base + (j&3) + 8*(j>>2) + 4h, query r)
        const long tile_row0 = row_lo + (long)t * TILE_ROWS;
        const bool ragged = tile_row0 + TILE_ROWS > row_hi;     // wave-uniform, last tile only
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            f32x16 a = mb ? acc1 : acc0;
            const int base = (int)tile_row0 + mb * 32 + 4 * h;
            if (ragged) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (base + (j & 3) + 8 * (j >> 2) >= (int)row_hi) a[j] = -INFINITY;
            }
            float m = fmaxf(fmaxf(a[0], a[1]), a[2]);
#pragma unroll
            for (int j = 3; j < 15; j += 2) m = fmaxf(fmaxf(m, a[j]), a[j + 1]);
            m = fmaxf(m, a[15]);
            if (__builtin_amdgcn_ballot_w64(m > ls[KP - 1]) != 0) {
                // A passing score parks in the lane's one pending slot; the 80-instruction
                // sorted insert runs only when some lane needs its slot again (then every
                // lane's pending entry goes in with that same pass).  ls[KP-1] may therefore
                // lag behind -- it only admits extra candidates, never drops one.
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const bool pass = a[j] > ls[KP - 1];
                    if (__builtin_amdgcn_ballot_w64(pass) != 0) {
                        if (__builtin_amdgcn_ballot_w64(pass && pend_i >= 0) != 0) {
                            list_insert<KP>(ls, li, pend_s, pend_i);
                            pend_s = -INFINITY; pend_i = -1;
                        }
                        const bool still = a[j] > ls[KP - 1];
                        pend_s = still ? a[j] : pend_s;
                        pend_i = still ? base + (j & 3) + 8 * (j >> 2) : pend_i;
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next tile landed
        __syncthreads();                                   // ... everyone's did, and this buffer is free
    }

    list_insert<KP>(ls, li, pend_s, pend_i);   // no-op for lanes with an empty slot (-inf)
    // ---- spill the lane lists: cand[q][split*2 + h][KP]
    if (q_glob < nq) {
        const size_t o = ((size_t)q_glob * (2 * S) + (size_t)split * 2 + h) * KP;
#pragma unroll
        for (int i = 0; i < KP; i += 4) {
            *reinterpret_cast<float4*>(cand_s + o + i) = make_float4(ls[i], ls[i + 1], ls[i + 2], ls[i + 3]);
            *reinterpret_cast<int4*>(cand_i + o + i) = make_int4(li[i], li[i + 1], li[i + 2], li[i + 3]);
        }
    }
}

// --------------------------------------------------------------------------------------------
// Per query: select the best K2 = k + slack candidates by (float32 score desc, id asc) out of
// L lists x KP entries, re-score them in float64 (sequential k order == the oracle's canonical
// score), order by (score desc, id asc) and write the first k.
//   status[q] = 0  result proven exact (every excluded row scores strictly below the k-th)
//             = 1  not proven (a full list may have dropped a contender, or the float32
//                  near-tie window reached the selection edge) -> caller re-runs the query
//                  through the exhaustive path.
constexpr int SEL_THREADS = 256;
constexpr int SEL_MAX_K2 = 128;

__global__ __launch_bounds__(SEL_THREADS) void k_select_rescore(
    const float* __restrict__ Q, const float* __restrict__ C, int d, int L,
    const float* __restrict__ cand_s, const int* __restrict__ cand_i, int k, int K2,
    long id_offset, float corpus_max_norm, float* __restrict__ D_out, long* __restrict__ I_out,
    int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = L * KP;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);          // [M]
    unsigned long long* sel = keys + M;                                              // [SEL_MAX_K2]
    double* resc = reinterpret_cast<double*>(sel + SEL_MAX_K2);                      // [SEL_MAX_K2]
    float* qrow = reinterpret_cast<float*>(resc + SEL_MAX_K2);                       // [d]
    __shared__ unsigned long long wmax[SEL_THREADS / 64];
    __shared__ unsigned long long s_maxlast;
    __shared__ float s_qnorm2;

    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* cs = cand_s + (size_t)q * M;
    const int* ci = cand_i + (size_t)q * M;
    unsigned long long lastmax = 0;
    for (int i = tid; i < M; i += SEL_THREADS) {
        const unsigned long long key = make_key(cs[i], ci[i]);
        keys[i] = key;
        if ((i % KP) == KP - 1 && ci[i] >= 0 && key > lastmax) lastmax = key;  // tail of a FULL list
    }
    float qs = 0.f;
    for (int i = tid; i < d; i += SEL_THREADS) {
        const float v = Q[(size_t)q * d + i];
        qrow[i] = v;
        qs += v * v;
    }
    // block reductions (max of list tails, |q|^2)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(lastmax, o);
        lastmax = other > lastmax ? other : lastmax;
        qs += __shfl_xor(qs, o);
    }
    __shared__ float wq[SEL_THREADS / 64];
    if (lane == 0) { wmax[wv] = lastmax; wq[wv] = qs; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long mm = 0; float qq = 0.f;
        for (int w = 0; w < SEL_THREADS / 64; ++w) { mm = wmax[w] > mm ? wmax[w] : mm; qq += wq[w]; }
        s_maxlast = mm; s_qnorm2 = qq;
    }
    __syncthreads();

    // ---- K2 rounds of block-wide arg-max extraction (keys are unique per real candidate)
    for (int it = 0; it < K2; ++it) {
        unsigned long long best = 0; int bidx = -1;
        for (int i = tid; i < M; i += SEL_THREADS) {
            const unsigned long long v = keys[i];
            if (v > best) { best = v; bidx = i; }
        }
        unsigned long long wbest = best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(wbest, o);
            wbest = other > wbest ? other : wbest;
        }
        if (lane == 0) wmax[wv] = wbest;
        __syncthreads();
        unsigned long long gbest = 0;
#pragma unroll
        for (int w = 0; w < SEL_THREADS / 64; ++w) gbest = wmax[w] > gbest ? wmax[w] : gbest;
        if (best == gbest && bidx >= 0) keys[bidx] = 0;   // unique owner (empty slots: any, harmless)
        if (tid == 0) sel[it] = gbest;
        __syncthreads();
    }

    // ---- float64 re-score of the selected rows, one thread per candidate, sequential in k
    if (tid < K2) {
        const int id = key_id(sel[tid]);
        double acc = 0.0;
        if (id >= 0 && sel[tid] != 0) {
            const float4* row = reinterpret_cast<const float4*>(C + (size_t)id * d);
            for (int kk = 0; kk < d / 4; ++kk) {
                const float4 v = row[kk];
                acc += (double)qrow[4 * kk + 0] * (double)v.x;
                acc += (double)qrow[4 * kk + 1] * (double)v.y;
                acc += (double)qrow[4 * kk + 2] * (double)v.z;
                acc += (double)qrow[4 * kk + 3] * (double)v.w;
            }
        }
        resc[tid] = acc;
    }
    __syncthreads();
    // ---- rank by (float32(score64) desc, id asc); write the first k
    __shared__ double s_kth;
    __shared__ int s_nvalid;
    if (tid == 0) { s_nvalid = 0; s_kth = 0.0; }
    __syncthreads();
    if (tid < K2) {
        const int id = key_id(sel[tid]);
        const bool valid = sel[tid] != 0 && id >= 0;
        if (valid) {
            const float s = (float)resc[tid];
            int rank = 0;
            for (int j = 0; j < K2; ++j) {
                const int idj = key_id(sel[j]);
                if (j == tid || sel[j] == 0 || idj < 0) continue;
                const float sj = (float)resc[j];
                if (sj > s || (sj == s && idj < id)) ++rank;
            }
            atomicAdd(&s_nvalid, 1);
            if (rank < k) {
                D_out[(size_t)q * k + rank] = s;
                I_out[(size_t)q * k + rank] = (long)id + id_offset;
            }
            if (rank == k - 1) s_kth = resc[tid];
        }
    }
    __syncthreads();
    for (int j = s_nvalid + tid; j < k; j += SEL_THREADS) {     // faiss pads missing results
        D_out[(size_t)q * k + j] = -3.4028234663852886e38f;
        I_out[(size_t)q * k + j] = -1;
    }
    if (tid == 0) {
        // Everything NOT selected has float32 key below sel[K2-1] (or below the tail of a full
        // list).  Proven exact when (a) no full list's tail outranks the selection edge and
        // (b) edge score + 2*B < k-th re-scored score, B = d * 2^-24 * |q| * max|c| bounding the
        // float32 fma-chain error of any row.
        int st = 0;
        const unsigned long long edge = sel[K2 - 1];
        if (s_maxlast > edge) st = 1;
        if (edge != 0 && key_id(edge) >= 0 && s_nvalid >= k) {
            const double B = (double)d * 5.9604644775390625e-08 * sqrt((double)s_qnorm2) *
                             (double)corpus_max_norm * 1.02;
            if ((double)key_score(edge) + 2.0 * B >= s_kth) st = 1;
        }
        status[q] = st;
    }
}

// --------------------------------------------------------------------------------------------
// k-way merge of per-shard results (after the RCCL all-gather): [shards][nq][k] -> [nq][k] by
// (score desc, id asc); ids < 0 are padding.  One thread per query (k*shards is tiny).
__global__ void k_topk_merge(const float* __restrict__ D_in, const long* __restrict__ I_in,
                             int shards, int nq, int k, float* __restrict__ D_out,
                             long* __restrict__ I_out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    int pos[64];
    for (int s = 0; s < shards; ++s) pos[s] = 0;
    for (int o = 0; o < k; ++o) {
        int bs = -1; float bd = 0.f; long bi = 0;
        for (int s = 0; s < shards; ++s) {
            if (pos[s] >= k) continue;
            const size_t a = ((size_t)s * nq + q) * k + pos[s];
            const long id = I_in[a];
            if (id < 0) { pos[s] = k; continue; }
            const float dd = D_in[a];
            if (bs < 0 || dd > bd || (dd == bd && id < bi)) { bs = s; bd = dd; bi = id; }
        }
        if (bs < 0) { D_out[(size_t)q * k + o] = -3.4028234663852886e38f; I_out[(size_t)q * k + o] = -1; }
        else { D_out[(size_t)q * k + o] = bd; I_out[(size_t)q * k + o] = bi; ++pos[bs]; }
    }
}

// ------------------------------------------------------------------------------ host launchers
static int pick_splits(long n, int G) {
    // S*G workgroups, one per CU (256 CUs); S a multiple of 8 (XCD remap); >= 64 rows a split.
    int S = (256 / G) & ~7;
    if (S < 8) S = 8;
    while (S > 8 && (long)S * TILE_ROWS > n) S -= 8;
    return S;
}

size_t ip_topk_workspace_bytes(long nq, long n, int d, int k) {
    const int G = (int)((nq + WG_QUERIES - 1) / WG_QUERIES);
    const int S = pick_splits(n, G);
    return (size_t)nq * 2 * S * KP * 8 + 256;
}

template <int D>
static int launch_scan(const float* q, int nq, const float* c, int n, int S, int G, int rps,
                       float* cs, int* ci, hipStream_t st) {
    const size_t lds = 2 * TILE_ROWS * D * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_topk_f32<D>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(k_ip_topk_f32<D>, dim3(S * G), dim3(512), lds, st, q, nq, c, n, rps, S, G, cs, ci);
    return check_launch("k_ip_topk_f32");
}

int ip_topk_f32(const float* q, long nq, const float* c, long n, int d, int k, long id_offset,
                float corpus_max_norm, float* D_out, long* I_out, int* status, void* ws,
                size_t ws_bytes, hipStream_t st) {
    if (nq <= 0 || n <= 0 || k <= 0) { set_error("ip_topk: nq, n, k must be positive"); return SSS_EINVAL; }
    if (d != 64 && d != 128 && d != 256) { set_error("ip_topk: d must be 64, 128 or 256 (got %d)", d); return SSS_EINVAL; }
    if (n >= (1L << 31) || nq >= (1L << 31)) { set_error("ip_topk: n and nq must be < 2^31 per shard"); return SSS_EINVAL; }
    int K2 = k + (k <= 12 ? KP - k : 12);
    if (K2 > SEL_MAX_K2) { set_error("ip_topk: k too large (max %d)", SEL_MAX_K2 - 12); return SSS_EINVAL; }
    const int G = (int)((nq + WG_QUERIES - 1) / WG_QUERIES);
    const int S = pick_splits(n, G);
    if (2 * S * KP < K2) K2 = 2 * S * KP;
    const size_t need = ip_topk_workspace_bytes(nq, n, d, k);
    if (ws_bytes < need) { set_error("ip_topk: workspace %zu < %zu", ws_bytes, need); return SSS_EWORKSPACE; }
    long rps = (n + S - 1) / S;
    rps = (rps + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
    float* cs = reinterpret_cast<float*>(ws);
    int* ci = reinterpret_cast<int*>(cs + (size_t)nq * 2 * S * KP);
    int rc;
    if (d == 64) rc = launch_scan<64>(q, (int)nq, c, (int)n, S, G, (int)rps, cs, ci, st);
    else if (d == 128) rc = launch_scan<128>(q, (int)nq, c, (int)n, S, G, (int)rps, cs, ci, st);
    else rc = launch_scan<256>(q, (int)nq, c, (int)n, S, G, (int)rps, cs, ci, st);
    if (rc) return rc;
    const int L = 2 * S;
    const size_t lds = (size_t)L * KP * 8 + SEL_MAX_K2 * 16 + (size_t)d * 4;
    hipLaunchKernelGGL(k_select_rescore, dim3((unsigned)nq), dim3(SEL_THREADS), lds, st, q, c, d, L, cs, ci,
                       k, K2, id_offset, corpus_max_norm, D_out, I_out, status);
    return check_launch("k_select_rescore");
}

int topk_merge(const float* D_in, const long* I_in, int shards, long nq, int k, float* D_out,
               long* I_out, hipStream_t st) {
    if (shards < 1 || shards > 64 || nq <= 0 || k <= 0) { set_error("topk_merge: bad arguments"); return SSS_EINVAL; }
    hipLaunchKernelGGL(k_topk_merge, dim3((unsigned)((nq + 127) / 128)), dim3(128), 0, st, D_in, I_in, shards,
                       (int)nq, k, D_out, I_out);
    return check_launch("k_topk_merge");
}

}  // namespace sss

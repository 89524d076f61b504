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

// PRE = true is the sampled pre-pass: same MFMA stream, but the epilogue only keeps each lane's
// running maximum; the K2-th largest of a query's 2*S lane maxima (k_tau) is a score that at
// least K2 distinct corpus rows reach, i.e. a valid admission threshold for the main pass.
//
// A tile is TR = 64*H corpus rows (one barrier per tile); a wave walks it in H sub-steps of 64
// rows (two 32x32 accumulators) with the top-k epilogue after every sub-step.
template <int D> struct ScanCfg { static constexpr int TR = D <= 128 ? 128 : 64; };

template <int D, bool PRE>
__global__ __launch_bounds__(512, 2) void k_ip_topk_f32(
    const float* __restrict__ Q, int nq, const float* __restrict__ C, int n, int tiles_per_split,
    int total_tiles, int tile_step_rows, int S, int G, const float* __restrict__ tau0,
    float* __restrict__ cand_s, int* __restrict__ cand_i) {
    constexpr int TR = ScanCfg<D>::TR;
    constexpr int H = TR / 64;
    constexpr int CH = D / 4;                       // 16-byte chunks per row
    constexpr int TILE_BYTES = TR * D * 4;
    constexpr int LOADS_PER_WAVE = TR * CH / 64 / 8;  // glds wave-instructions per wave per tile
    constexpr bool PRECOMP = D <= 128;               // keep the DMA lane offsets in VGPRs (register budget)
    static_assert(CH <= 64, "row longer than one LDS-DMA instruction");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // (Deferring the epilogue of waves 4-7 by one sub-step -- the "stagger" of the MI355X guide --
    // was measured here at 1M and 10M rows: no gain over running every wave in phase, so the
    // simpler in-phase form is kept.)

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
    float t0 = -INFINITY;
    {
        const float4* qp = reinterpret_cast<const float4*>(Q + (size_t)q_ld * D) + h;
#pragma unroll
        for (int u = 0; u < D / 8; ++u) {
            const float4 v = qp[2 * u];
            qreg[4 * u + 0] = v.x; qreg[4 * u + 1] = v.y;
            qreg[4 * u + 2] = v.z; qreg[4 * u + 3] = v.w;
        }
        if (tau0) t0 = tau0[q_ld];
        // retire the query loads HERE: otherwise hipcc sinks their counted vmcnt waits into the
        // tile loop, where they would also wait on the (uncounted) LDS-DMA of the next tile.
#pragma unroll
        for (int t = 0; t < D / 2; ++t) asm volatile("" : "+v"(qreg[t]));
        asm volatile("" : "+v"(t0));
    }

    // Lane list, sorted descending.  Empty slots are (t0, -1): t0 is the admission threshold
    // handed in by the sampled pre-pass (-inf without one), so "beats the list tail" is the only
    // test the hot path needs.
    float ls[KP];
    int li[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) { ls[i] = t0; li[i] = -1; }
    float pend_s = -INFINITY;   // one parked candidate per lane (see the epilogue)
    int pend_i = -1;

    int tile_lo = split * tiles_per_split;
    int tile_hi = tile_lo + tiles_per_split;
    if (tile_hi > total_tiles) tile_hi = total_tiles;
    const int ntiles = tile_lo < tile_hi ? tile_hi - tile_lo : 0;

    // LDS-DMA staging (global_load_lds_dwordx4, 1 KiB per wave-instruction).  Written as inline
    // asm so hipcc neither counts it nor drains vmcnt(0) at the next ds_read: the next tile
    // stays in flight under this tile's MFMAs and is retired by the explicit vmcnt(0) that
    // precedes the barrier at the end of the iteration (cdna_hip_programming.md section 5.7).
    // Slot p of the tile (16 B each) holds chunk (p % CH) ^ (row & 15) of row p / CH: the
    // swizzle is on the SOURCE address, the LDS image is lane-linear.
    const unsigned lds_base = (unsigned)(unsigned long)(lptr_c)smem;
    auto slot_row = [&](int i) { return ((wave * LOADS_PER_WAVE + i) * 64 + lane) / CH; };
    auto slot_off = [&](int i) {                    // byte offset of this lane's chunk inside the tile
        const int p = (wave * LOADS_PER_WAVE + i) * 64 + lane;
        const int tr = p / CH, sc = p % CH;
        return (unsigned)(tr * D * 4 + ((sc ^ (tr & 15)) * 16));
    };
    unsigned lane_off[PRECOMP ? LOADS_PER_WAVE : 1];
    if (PRECOMP) {
#pragma unroll
        for (int i = 0; i < LOADS_PER_WAVE; ++i) lane_off[i] = slot_off(i);
    }
    // One LDS-DMA wave-instruction (piece i of this wave's share of a tile).
    auto stage_piece = [&](int buf, int tile_idx, int i) {
        const long row0 = (long)tile_idx * tile_step_rows;
        const bool inside = row0 + TR <= (long)n;          // wave-uniform
        const float* tile_src = C + (size_t)row0 * D;       // wave-uniform -> SGPR pair
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            lds_base + buf * TILE_BYTES + (wave * LOADS_PER_WAVE + i) * 1024);
        unsigned off = PRECOMP ? lane_off[PRECOMP ? i : 0] : slot_off(i);
        if (!inside) {                                       // ragged last tile: clamp the row
            const int lr = slot_row(i);
            long grow = row0 + lr;
            if (grow > (long)n - 1) grow = (long)n - 1;
            off = (unsigned)((grow - row0) * D * 4) + (off - (unsigned)(lr * D * 4));
        }
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(dst), "s"(tile_src) : "memory");
    };
    auto stage = [&](int buf, int tile_idx) {
#pragma unroll
        for (int i = 0; i < LOADS_PER_WAVE; ++i) stage_piece(buf, tile_idx, i);
    };

    // per-lane LDS read offset (bytes) of chunk (2u + h) of row r, before the constant part
    const int x = h ^ (r & 15);
    f32x16 acc0 = {0}, acc1 = {0};

    auto mfma_sub = [&](int buf, int sub, int next_tile) {
        const char* tile = smem + buf * TILE_BYTES + sub * (64 * D * 4);
        auto lda = [&](int u, int mb) -> float4 {
            const int c = (2 * u) ^ x;                          // == (2u + h) ^ (r & 15)
            return *reinterpret_cast<const float4*>(tile + ((r + 32 * mb) * CH + c) * 16);
        };
        float4 a0 = lda(0, 0), a1 = lda(0, 1);
        const f32x16 zero = {0};
        acc0 = zero; acc1 = zero;
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
#ifndef SSS_EXP_NO_STAGE
            // one DMA piece per k-group: each issue hides under the MFMAs this wave just queued
            if (u >= 1 && u - 1 < LOADS_PER_WAVE && next_tile >= 0) {
                stage_piece(buf ^ 1, next_tile, u - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
    };

    // Top-k epilogue of one 32x32 accumulator: a[j] is (corpus row base + (j&3) + 8*(j>>2), query r).
    // Hot path: quarter maxima (rows 8g..8g+3 of this lane's 16) + one compare.  When some lane's
    // maximum beats its list tail, only the quarters that hold a passing score are walked.
    auto epilogue_block = [&](const f32x16& a, int base) {
        const float q0 = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
        const float q1 = fmaxf(fmaxf(a[4], a[5]), fmaxf(a[6], a[7]));
        const float q2 = fmaxf(fmaxf(a[8], a[9]), fmaxf(a[10], a[11]));
        const float q3 = fmaxf(fmaxf(a[12], a[13]), fmaxf(a[14], a[15]));
        const float m = fmaxf(fmaxf(q0, q1), fmaxf(q2, q3));
        if (PRE) { pend_s = fmaxf(pend_s, m); return; }
        if (__builtin_amdgcn_ballot_w64(m > ls[KP - 1]) == 0) return;
        // A passing score parks in the lane's one pending slot; the 80-instruction sorted insert
        // runs only when some lane needs its slot again (then every lane's pending entry goes in
        // with that same pass).  ls[KP-1] may therefore lag behind -- it only admits extra
        // candidates, never drops one.
        auto walk = [&](float qm, int j0) {
            if (__builtin_amdgcn_ballot_w64(qm > ls[KP - 1]) == 0) return;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j0 + jj;
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
        };
        walk(q0, 0); walk(q1, 4); walk(q2, 8); walk(q3, 12);     // ascending row order per lane
    };

    if (ntiles > 0) stage(0, tile_lo);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        for (int sub = 0; sub < H; ++sub) {
            const int next_tile = (sub == 0 && t + 1 < ntiles) ? tile_lo + t + 1 : -1;
            mfma_sub(buf, sub, next_tile);

            // ---- fused top-k epilogue of this sub-step.
            // acc[j] is (corpus row base + (j&3) + 8*(j>>2) + 4h, query r).
#ifdef SSS_EXP_NO_EPILOGUE
            asm volatile("" :: "v"(acc0), "v"(acc1));
            if (false) {
#else
            {
#endif
                const long sub_row0 = (long)(tile_lo + t) * tile_step_rows + sub * 64;
                if (sub_row0 + 64 > n) {                        // wave-uniform, last tile only
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int rr = (int)sub_row0 + 4 * h + (j & 3) + 8 * (j >> 2);
                        if (rr >= n) acc0[j] = -INFINITY;
                        if (rr + 32 >= n) acc1[j] = -INFINITY;
                    }
                }
                epilogue_block(acc0, (int)sub_row0 + 4 * h);
                epilogue_block(acc1, (int)sub_row0 + 32 + 4 * h);
            }
        }
#ifndef SSS_EXP_NO_STAGE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next tile landed
#endif
#if !defined(SSS_EXP_NO_STAGE) || defined(SSS_EXP_KEEP_BARRIER)
#ifndef SSS_EXP_NO_BARRIER
        __syncthreads();                                   // ... everyone's did, and this buffer is free
#endif
#endif
    }

    if (PRE) {      // lane maximum -> cand_s[q][split*2 + h]
        if (q_glob < nq) cand_s[(size_t)q_glob * (2 * S) + (size_t)split * 2 + h] = pend_s;
        return;
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
constexpr unsigned long long EMPTY_KEY = 0x007FFFFF00000000ull;   // make_key(-inf, -1)

// Load the L*KP candidates of query q as keys into LDS; returns (via wave 0 lane 0 in *maxlast)
// the largest tail key over FULL lists.  All threads of the block must call.
__device__ __forceinline__ unsigned long long load_keys(const float* cs, const int* ci, int M,
                                                        unsigned long long* keys, int tid) {
    unsigned long long lastmax = 0;
    for (int i = tid; i < M; i += SEL_THREADS) {
        const int id = ci[i];
        const unsigned long long key = id >= 0 ? make_key(cs[i], id) : EMPTY_KEY;
        keys[i] = key;
        if ((i % KP) == KP - 1 && id >= 0 && key > lastmax) lastmax = key;   // tail of a FULL list
    }
    return lastmax;
}

// K2 rounds of block-wide arg-max extraction (keys of real candidates are unique).  sel[it]
// receives the it-th best key (0 once the candidates are exhausted).
__device__ __forceinline__ void extract_top(unsigned long long* keys, int M, int K2,
                                            unsigned long long* sel, unsigned long long* wmax,
                                            int tid) {
    const int lane = tid & 63, wv = tid >> 6;
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
}

// Sampled pre-pass -> admission threshold of the main pass: tau[q] = the float just below the
// K2-th largest of the L lane maxima of query q.  Each maximum is the score of a distinct corpus
// row, so at least K2 rows score above tau[q] and no row at or below it can be among the best
// K2.  One wave per query; L <= 512.
__global__ __launch_bounds__(64) void k_tau(const float* __restrict__ pre_max, int L, int K2,
                                            float* __restrict__ tau) {
    const int q = blockIdx.x, lane = threadIdx.x;
    unsigned v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = lane + 64 * i;
        v[i] = idx < L ? f2ord(pre_max[(size_t)q * L + idx]) : 0u;
    }
    unsigned cur = 0;
    for (int it = 0; it < K2; ++it) {      // K2 rounds: wave-wide max, remove one instance
        unsigned best = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) best = v[i] > best ? v[i] : best;
        unsigned w = best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned other = __shfl_xor(w, o); w = other > w ? other : w; }
        cur = w;
        const unsigned long long owners = __builtin_amdgcn_ballot_w64(best == w);
        const int first = __builtin_ctzll(owners);
        if (lane == first) {
            bool done = false;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (!done && v[i] == w) { v[i] = 0u; done = true; }
        }
    }
    if (lane == 0) tau[q] = (cur > f2ord(-INFINITY)) ? ord2f(cur - 1) : -INFINITY;
}

__global__ __launch_bounds__(SEL_THREADS) void k_select_rescore(
    const float* __restrict__ Q, const float* __restrict__ C, int d, int L,
    const float* __restrict__ cand_s, const int* __restrict__ cand_i, const float* __restrict__ tau0,
    int k, int K2, long id_offset, float corpus_max_norm, float* __restrict__ D_out,
    long* __restrict__ I_out, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = L * KP;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);          // [M]
    unsigned long long* sel = keys + M;                                              // [SEL_MAX_K2]
    double* resc = reinterpret_cast<double*>(sel + SEL_MAX_K2);                      // [SEL_MAX_K2]
    float* qrow = reinterpret_cast<float*>(resc + SEL_MAX_K2);                       // [d]
    float* rows = qrow + d;                                                          // [K2][d + 4]
    __shared__ unsigned long long wmax[SEL_THREADS / 64];
    __shared__ float wq[SEL_THREADS / 64];
    __shared__ unsigned long long s_maxlast;
    __shared__ float s_qnorm2;
    __shared__ double s_kth;
    __shared__ int s_nvalid;

    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned long long lastmax = load_keys(cand_s + (size_t)q * M, cand_i + (size_t)q * M, M, keys, tid);
    float qs = 0.f;
    for (int i = tid; i < d; i += SEL_THREADS) {
        const float v = Q[(size_t)q * d + i];
        qrow[i] = v;
        qs += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(lastmax, o);
        lastmax = other > lastmax ? other : lastmax;
        qs += __shfl_xor(qs, o);
    }
    if (lane == 0) { wmax[wv] = lastmax; wq[wv] = qs; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long mm = 0; float qq = 0.f;
        for (int w = 0; w < SEL_THREADS / 64; ++w) { mm = wmax[w] > mm ? wmax[w] : mm; qq += wq[w]; }
        s_maxlast = mm; s_qnorm2 = qq; s_nvalid = 0; s_kth = 0.0;
    }
    __syncthreads();

    extract_top(keys, M, K2, sel, wmax, tid);

    // ---- stage the K2 selected corpus rows in LDS (coalesced 16-byte loads, all threads)
    const int ldr = d + 4;
    const int nv = d / 4;
    for (int i = tid; i < K2 * nv; i += SEL_THREADS) {
        const int c = i / nv, v = i % nv;
        const unsigned long long key = sel[c];
        const int id = key_id(key);
        if (key != 0 && id >= 0)
            *reinterpret_cast<float4*>(rows + c * ldr + v * 4) =
                *reinterpret_cast<const float4*>(C + (size_t)id * d + v * 4);
    }
    __syncthreads();
    // ---- float64 re-score, one thread per candidate, sequential in k (the canonical order)
    if (tid < K2) {
        const int id = key_id(sel[tid]);
        double acc = 0.0;
        if (id >= 0 && sel[tid] != 0) {
            const float* row = rows + tid * ldr;
            for (int kk = 0; kk < d; ++kk) acc += (double)qrow[kk] * (double)row[kk];
        }
        resc[tid] = acc;
    }
    __syncthreads();
    // ---- rank by (float32(score64) desc, id asc); write the first k
    if (tid < K2) {
        const int id = key_id(sel[tid]);
        const bool valid = sel[tid] != 0 && id >= 0;
        if (valid) {
            const float sc = (float)resc[tid];
            int rank = 0;
            for (int j = 0; j < K2; ++j) {
                const int idj = key_id(sel[j]);
                if (j == tid || sel[j] == 0 || idj < 0) continue;
                const float sj = (float)resc[j];
                if (sj > sc || (sj == sc && idj < id)) ++rank;
            }
            atomicAdd(&s_nvalid, 1);
            if (rank < k) {
                D_out[(size_t)q * k + rank] = sc;
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
        // A row that was NOT selected (a) lost to a full list's tail, (b) scored at or below the
        // pre-pass threshold tau0, or (c) is a candidate ranked below sel[K2-1].  Proven exact
        // when no full list's tail and no tau0 outranks the selection edge, and
        // edge score + 2*B < k-th re-scored score, with B = d * 2^-24 * |q| * max|c| bounding the
        // float32 fma-chain error of any row.
        int st = 0;
        const unsigned long long edge = sel[K2 - 1];
        const bool edge_real = edge != 0 && key_id(edge) >= 0;
        if (s_maxlast > edge) st = 1;
        if (tau0 != nullptr && tau0[q] > -INFINITY && !(edge_real && key_score(edge) > tau0[q])) st = 1;
        if (edge_real && s_nvalid >= k) {
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

// ------------------------------------------------------------------------------ host launchers
// Optional timing of the dominant kernel (bench.py roofline leg): when enabled, every main-pass
// scan launch is bracketed by a hipEvent pair on ITS stream; profile_read() drains the ring.
namespace {
constexpr int PROF_RING = 512;
struct Prof {
    bool on = false;
    int n = 0;
    hipEvent_t ev[2 * PROF_RING];
    bool made = false;
} g_prof;
}  // namespace

int profile_enable(int on) {
    if (on && !g_prof.made) {
        for (int i = 0; i < 2 * PROF_RING; ++i)
            if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) { set_error("profile_enable: hipEventCreate failed"); return SSS_EHIP; }
        g_prof.made = true;
    }
    g_prof.on = on != 0;
    g_prof.n = 0;
    return SSS_OK;
}

int profile_read(double* total_ms, int* launches) {
    double sum = 0.0;
    for (int i = 0; i < g_prof.n; ++i) {
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) { set_error("profile_read: sync failed"); return SSS_EHIP; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) { set_error("profile_read: elapsed failed"); return SSS_EHIP; }
        sum += ms;
    }
    *total_ms = sum;
    *launches = g_prof.n;
    g_prof.n = 0;
    return SSS_OK;
}

static int tile_rows_for(int d) { return d <= 128 ? 128 : 64; }   // == ScanCfg<D>::TR

static int pick_splits(long n, int G, int tr) {
    // S*G workgroups, one per CU (256 CUs); S a multiple of 8 (XCD remap); >= one tile a split.
    int S = (256 / G) & ~7;
    if (S < 8) S = 8;
    while (S > 8 && (long)S * tr > n) S -= 8;
    return S;
}

struct ScanPlan {
    int G, S, L, K2, tile_rows;
    int total_tiles, tiles_per_split;       // main pass
    int pre_tiles, pre_tiles_per_split, pre_step_rows;   // sampled pre-pass (pre_tiles == 0: none)
};

static ScanPlan make_plan(long nq, long n, int d, int k) {
    ScanPlan p;
    const int TILE_ROWS = tile_rows_for(d);
    p.tile_rows = TILE_ROWS;
    p.G = (int)((nq + WG_QUERIES - 1) / WG_QUERIES);
    p.S = pick_splits(n, p.G, TILE_ROWS);
    p.L = 2 * p.S;
    p.K2 = k + (k <= 12 ? KP - k : 12);
    if (p.L * KP < p.K2) p.K2 = p.L * KP;
    p.total_tiles = (int)((n + TILE_ROWS - 1) / TILE_ROWS);
    p.tiles_per_split = (p.total_tiles + p.S - 1) / p.S;
    // Pre-pass sample: about 2n/L rows (so a lane list of the main pass admits ~KP rows),
    // at least 4 tiles a split, evenly spaced tiles; skipped unless it is < 1/8 of the corpus.
    long want_rows = 2 * n / p.L;
    if (want_rows < 256L * p.S) want_rows = 256L * p.S;
    int ptps = (int)((want_rows / TILE_ROWS + p.S - 1) / p.S);
    p.pre_tiles = 0; p.pre_tiles_per_split = 0; p.pre_step_rows = 0;
    if ((long)ptps * p.S * 8 <= p.total_tiles && p.L >= p.K2 && p.L <= 512) {
        p.pre_tiles_per_split = ptps;
        p.pre_tiles = ptps * p.S;
        p.pre_step_rows = (p.total_tiles / p.pre_tiles) * TILE_ROWS;
    }
    return p;
}

size_t ip_topk_workspace_bytes(long nq, long n, int d, int k) {
    const ScanPlan p = make_plan(nq, n, d, k);
    return (size_t)nq * p.L * KP * 8 + (size_t)nq * 4 + 256;
}

template <int D>
static void set_lds_attr() {
    const int lds = 2 * ScanCfg<D>::TR * D * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_topk_f32<D, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ip_topk_f32<D, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

template <int D>
static int launch_scan(const float* q, int nq, const float* c, int n, const ScanPlan& p, bool pre,
                       const float* tau, float* cs, int* ci, hipStream_t st) {
    const size_t lds = 2 * ScanCfg<D>::TR * D * 4;
    static bool attr_done = false;
    if (!attr_done) { set_lds_attr<D>(); attr_done = true; }
    if (pre)
        hipLaunchKernelGGL((k_ip_topk_f32<D, true>), dim3(p.S * p.G), dim3(512), lds, st, q, nq, c, n,
                           p.pre_tiles_per_split, p.pre_tiles, p.pre_step_rows, p.S, p.G,
                           (const float*)nullptr, cs, ci);
    else
        hipLaunchKernelGGL((k_ip_topk_f32<D, false>), dim3(p.S * p.G), dim3(512), lds, st, q, nq, c, n,
                           p.tiles_per_split, p.total_tiles, p.tile_rows, p.S, p.G, tau, cs, ci);
    return check_launch("k_ip_topk_f32");
}

static int scan_dispatch(int d, const float* q, int nq, const float* c, int n, const ScanPlan& p,
                         bool pre, const float* tau, float* cs, int* ci, hipStream_t st) {
    if (d == 64) return launch_scan<64>(q, nq, c, n, p, pre, tau, cs, ci, st);
    if (d == 128) return launch_scan<128>(q, nq, c, n, p, pre, tau, cs, ci, st);
    return launch_scan<256>(q, nq, c, n, p, pre, tau, cs, ci, st);
}

int ip_topk_f32(const float* q, long nq, const float* c, long n, int d, int k, long id_offset,
                float corpus_max_norm, float* D_out, long* I_out, int* status, void* ws,
                size_t ws_bytes, hipStream_t st) {
    if (nq <= 0 || n <= 0 || k <= 0) { set_error("ip_topk: nq, n, k must be positive"); return SSS_EINVAL; }
    if (d != 64 && d != 128 && d != 256) { set_error("ip_topk: d must be 64, 128 or 256 (got %d)", d); return SSS_EINVAL; }
    if (n >= (1L << 31) - 1024 || nq >= (1L << 31)) { set_error("ip_topk: n and nq must be < 2^31 per shard"); return SSS_EINVAL; }
    if (k + 12 > SEL_MAX_K2) { set_error("ip_topk: k too large (max %d)", SEL_MAX_K2 - 12); return SSS_EINVAL; }
    const ScanPlan p = make_plan(nq, n, d, k);
    const size_t need = ip_topk_workspace_bytes(nq, n, d, k);
    if (ws_bytes < need) { set_error("ip_topk: workspace %zu < %zu", ws_bytes, need); return SSS_EWORKSPACE; }
    float* cs = reinterpret_cast<float*>(ws);
    int* ci = reinterpret_cast<int*>(cs + (size_t)nq * p.L * KP);
    float* tau = reinterpret_cast<float*>(ci + (size_t)nq * p.L * KP);
    int rc;
    const size_t key_lds = (size_t)p.L * KP * 8;
    if (p.pre_tiles > 0) {
        rc = scan_dispatch(d, q, (int)nq, c, (int)n, p, true, nullptr, cs, ci, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_tau, dim3((unsigned)nq), dim3(64), 0, st, cs, p.L, p.K2, tau);
        rc = check_launch("k_tau");
        if (rc) return rc;
    }
    const float* tau_in = p.pre_tiles > 0 ? tau : nullptr;
    const bool prof = g_prof.on && g_prof.n < PROF_RING;
    if (prof) (void)hipEventRecord(g_prof.ev[2 * g_prof.n], st);
    rc = scan_dispatch(d, q, (int)nq, c, (int)n, p, false, tau_in, cs, ci, st);
    if (prof) { (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st); ++g_prof.n; }
    if (rc) return rc;
    const size_t lds = key_lds + SEL_MAX_K2 * 16 + (size_t)d * 4 + (size_t)p.K2 * (d + 4) * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_rescore),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        attr_done = true;
    }
    hipLaunchKernelGGL(k_select_rescore, dim3((unsigned)nq), dim3(SEL_THREADS), lds, st, q, c, d, p.L, cs, ci,
                       tau_in, k, p.K2, id_offset, corpus_max_norm, D_out, I_out, status);
    return check_launch("k_select_rescore");
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

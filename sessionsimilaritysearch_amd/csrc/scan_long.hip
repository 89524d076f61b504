// Scoring + top-k for LONG rows -- the reference's own session vectors are D = 1600 wide with K = 100
// (pretrain_filtered_amazon.py:281, test_amazon_filterd.py:459,578: `index.search(normalize(emb), K)`), far beyond
// the 1024-byte rows k_scan keeps resident in registers.  gfx950 / CDNA4.
//
// A row of 3200 bytes (f16 image of a float32 index) or more fits neither the registers nor LDS next to a query
// tile, so the scan is a K-TILED contraction: k_scan_long = 256 queries x 256 corpus rows per workgroup tile,
// both operands streamed through LDS in 128-byte K slabs (LDS-DMA, double buffered, XOR swizzle on the source
// address), v_mfma_f32_32x32x16_{f16,bf16} accumulating the full dot products in 8 accumulators per wave
// (a wave = 64 queries x 128 rows: two B operands x four row blocks; a lane owns one query column per operand).
//
// The top-k rides on the THRESHOLD machinery of the rung (select.hip: THRESHOLD RUNG) instead of running lists:
// the epilogue of a tile only compares its 128 scores per lane with the lane's fixed threshold and appends what
// passes.  The threshold comes from LEVELS of evenly spread row samples:
//   level 1   a sample of up to `cap` rows (>= 2k), threshold -inf: every sampled row is kept; at least k of them have a scan
//             score >= the k-th largest kept scan score s_k, hence an exact score >= s_k - (scan error bound): a
//             valid LOWER BOUND of the true k-th score (k_bound_prepare: a selection over scan scores, no row is read);
//   level i   a sample r <= `fmax` times larger scanned with threshold = (bound - error bound - one ulp): about
//             k * fmax rows pass per query -> a tighter bound the same way;
//   last      every row.  What passes is everything that can still reach the k-th score already known -- near ties
//             and duplicate rows included -- so the canonical float64 re-score of all of it (k_select_all) IS the
//             exact answer (status 0); a query with more than 8192 such rows keeps status 1 and goes to the
//             exhaustive kernels.  The last sample is 1/r of the corpus: ~r k rows per query reach the re-score.
//   With three levels and more the last sample and the final level are DISJOINT (host side, "DISJOINT LEVELS"): the
//   sample takes every R-th tile, the final level the tiles in between, and the rows the sample kept stay (pruned in place
//   to the final threshold) -- no tile is scanned twice.
// Two to three passes, the last one dominant; no per-lane list, no shared threshold slots, no bootstrap.
#include "scan.h"
#include <cmath>
#include <cstdlib>
#include "scan_dev.h"

namespace sss {

constexpr int LT_ROWS = 256;        // corpus rows per workgroup tile
constexpr int LT_Q = 256;           // queries per workgroup (8 waves x 32)
constexpr int LT_BK = 128;          // bytes of K per slab (64 16-bit elements = 4 MFMA k-groups)
constexpr int LT_HALF = LT_ROWS * LT_BK;               // 32 KB: one slab of ONE operand (LT_ROWS == LT_Q)
// LDS: the corpus slabs in a ring of THREE, the query slabs in a ring of two -- all 160 KB.  The corpus slab (streamed
// from HBM: the long latency) is fetched two slabs ahead, the query slab (L2-resident) one: half the bytes that must
// land within one slab's matrix time, and the far operand gets the deep look-ahead (counters, round 4: the waves were
// parked 52 % of their cycles waiting for the single 64 KB slab in flight; L2 hit rate 71 %, fetch from HBM ~1.2x
// algorithmic, no LDS stall or conflict -- the bound was the latency of a one-slab look-ahead, not a rate).
constexpr int LT_LDS_BYTES = 5 * LT_HALF;
static_assert(LT_ROWS == LT_Q, "the two operand slabs have one size");

struct LongArgs {
    const void* Qimg;               // [nq][d] 16-bit queries (f16: scaled image, bf16: the queries themselves)
    const void* C;                  // [n][d] 16-bit rows (f16 image / bf16 rows)
    int nq, n, d, G, S;
    int gpx;                        // query groups that share an XCD (block map below); divides G, G / gpx divides 8
    int dense;                      // sample level (many rows kept per lane and tile): aggregated appends
    int tile_count;                 // tiles this level scans: tile j of the level = corpus tile j * total_tiles / tile_count
    int skip_R;                     // > 0: the level scans the tile_count corpus tiles that are NOT multiples of skip_R, in order
                                    //      (the complement of a level that took every skip_R-th tile: DISJOINT LEVELS, host side)
    int total_tiles, tiles_per_split, cap;
    const float* thr;               // [nq] per-query threshold in the scan's domain
    unsigned* cnt;                  // [nq] rows kept so far
    unsigned long long* cand;       // [nq][cap]
};

// (the f16 image of f32 queries -- scaled by the query's own power of two, scan.h: f16_shift -- is written by
//  select.hip: k_long_setup; the select side, err_bound / k_thr_prepare, re-derives the same shift)

#ifndef SSS_LT_LOADERS
#define SSS_LT_LOADERS 4
#endif
constexpr int LT_LOADERS = SSS_LT_LOADERS;             // loader waves per workgroup (one per SIMD), behind the 8 matrix waves
constexpr int LT_NP = 32 / LT_LOADERS;                 // 1 KiB pieces of an operand slab per loader
constexpr int LT_THREADS = (8 + LT_LOADERS) * 64;

template <int DT>
__global__ __launch_bounds__(LT_THREADS, (LT_THREADS + 255) / 256) void k_scan_long(const LongArgs A) {
    static_assert(DT == DT_F16 || DT == DT_BF16, "16-bit rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nq = A.nq, n = A.n, S = A.S, G = A.G;
    const int rb = A.d * 2;                             // bytes per row (queries and corpus alike)
    const int nslab = rb / LT_BK;
    const char* __restrict__ Qb = reinterpret_cast<const char*>(A.Qimg);
    const char* __restrict__ Cb = reinterpret_cast<const char*>(A.C);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware block map: the G query groups that stream one corpus split sit on ONE XCD (blocks b and b + 8 share an
    // XCD and its L2: the split is fetched from HBM once and re-read from that L2 by the other groups; speed only, never
    // correctness).  gpx < G -- an XCD serving fewer groups and more splits, so that its query images (800 KB a group
    // at d = 1600) stay L2-resident -- was measured in round 4: 1-4 % SLOWER (L2 hit rate 71 % as it is; the queries
    // are not what misses), so the plan keeps gpx = G.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int gpx = A.gpx, nteam = G / gpx;                       // XCD teams: team t serves groups [t gpx, (t + 1) gpx)
    const int team = xcd % nteam, xi = xcd / nteam;               // this XCD's team and its index inside the team
    const int splits_per_xcd = (S * G >> 3) / gpx;
    const int split = xi * splits_per_xcd + slot / gpx;
    const int g = team * gpx + slot % gpx;
    int j_lo = split * A.tiles_per_split, j_hi = j_lo + A.tiles_per_split;
    if (j_hi > A.tile_count) j_hi = A.tile_count;
    if (j_lo >= j_hi) return;                           // whole workgroup: no barrier below is skipped by part of it

    const int total_steps = (j_hi - j_lo) * nslab;             // slab steps of this split
    // tile j of the level = corpus tile floor(j * total_tiles / tile_count), advanced incrementally (one division here
    // instead of a 64-bit scalar division sequence per tile)
    // (skip_R > 0: the j-th tile that is not a multiple of R is (j / (R - 1)) R + 1 + j % (R - 1); `tile_frac` then holds
    //  tile % R and a step that lands on a multiple of R moves one further)
    const int skip_R = A.skip_R;
    const int t_quo = A.total_tiles / A.tile_count, t_rem = A.total_tiles % A.tile_count;
    int tile = skip_R > 0 ? (j_lo / (skip_R - 1)) * skip_R + 1 + j_lo % (skip_R - 1) : (int)((long)j_lo * A.total_tiles / A.tile_count);
    int tile_frac = skip_R > 0 ? 1 + j_lo % (skip_R - 1) : (int)((long)j_lo * A.total_tiles % A.tile_count);
    auto next_tile = [&](int& tl, int& fr) __attribute__((always_inline)) {
        if (skip_R > 0) {
            ++tl;
            if (++fr == skip_R) { fr = 1; ++tl; }
            return;
        }
        tl += t_quo; fr += t_rem;
        if (fr >= A.tile_count) { fr -= A.tile_count; ++tl; }
    };

    // ---- LDS: a slab is [256 rows][128 B] of one operand in 1 KiB pieces (8 rows x 8 chunks); chunk c of row t sits
    // at chunk slot c ^ ((t >> 1) & 7): with 128-byte rows two rows share a 256-byte bank row, and this key makes every
    // 16-lane group of a ds_read_b128 (16 consecutive rows, one chunk) cover all 64 banks exactly once.  Corpus slabs
    // in a ring of three slots, query slabs in a ring of two behind them.
    // Steps run over (tile, slab) pairs; step t reads corpus slot t % 3 and query slot t % 2; ONE barrier per step.
    //
    // LOADER WAVES (round 4).  Twelve waves: eight matrix waves and four loaders, one per SIMD.  A DMA instruction costs
    // its wave 100-200 cycles of issue in this loop (MI355X_MICROARCH.md, LDS-DMA piece issue cost) -- with every wave
    // issuing its eight pieces of each slab, ~1.2 k cycles per wave and slab in which it issued no matrix instruction,
    // against 1 k of its own matrix work: the pipe was 43 % busy with the waves parked half of their time.  Here the
    // matrix waves issue NO vector-memory instruction inside the K loop (their counter only sees the tile epilogue's
    // stores), the loaders nothing else: sixteen pieces per loader and step -- the query pieces of step t + 1, then the
    // corpus pieces of step t + 2 -- and `vmcnt(8)` at the top of a step leaves exactly the youngest eight (corpus,
    // t + 2) in flight while everything step t + 1 reads is known to have landed before the barrier releases it.
    const unsigned lds_base = (unsigned)(unsigned long)(lptr_c)smem;
    if (wave >= 8) {
        const int lw = wave - 8;
        unsigned off_q[LT_NP], off_c[LT_NP];
        const char* q_base = Qb + (size_t)g * LT_Q * rb;
        const char* c_base = Cb;
#pragma unroll
        for (int i = 0; i < LT_NP; ++i) {
            const int t = ((lw + LT_LOADERS * i) << 3) + (lane >> 3);    // query row of the tile: piece lw + 4 i
            int tq = t;
            if (g * LT_Q + tq > nq - 1) tq = nq - 1 - g * LT_Q;          // short query batch: clamp (the group holds >= 1 query)
            off_q[i] = (unsigned)tq * (unsigned)rb + (unsigned)(((lane & 7) ^ ((t >> 1) & 7)) * 16);
        }
        auto set_tile = [&](int tl) __attribute__((always_inline)) {
            c_base = Cb + (size_t)tl * LT_ROWS * rb;
#pragma unroll
            for (int i = 0; i < LT_NP; ++i) {
                const int t = ((lw + LT_LOADERS * i) << 3) + (lane >> 3);   // corpus row of the tile
                int tc = t;
                if ((long)tl * LT_ROWS + tc > (long)n - 1) tc = (int)((long)n - 1 - (long)tl * LT_ROWS);   // ragged last tile: clamp
                off_c[i] = (unsigned)tc * (unsigned)rb + (unsigned)(((lane & 7) ^ ((t >> 1) & 7)) * 16);
            }
        };
        // one operand slab: a uniform 64-bit base (first row + slab offset: scalar adds) and per-lane 32-bit offsets
        auto stage = [&](const char* base, const unsigned (&off)[LT_NP], int slot_idx, int slab) __attribute__((always_inline)) {
            const unsigned long bv = (unsigned long)(base + (size_t)slab * LT_BK);
            // (readfirstlane returns int: without the unsigned casts a low word with bit 31 set sign-extends into the high word)
            const unsigned long bs = ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bv >> 32)) << 32) | (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)bv);
#pragma unroll
            for (int i = 0; i < LT_NP; ++i) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot_idx * LT_HALF + (lw + LT_LOADERS * i) * 1024);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
                             : : "v"(off[i]), "s"(dst), "s"(bs) : "memory");
            }
        };
        int itile = tile, ifrac = tile_frac, islab = 0, ibuf = 0;   // next corpus slab to fetch and its ring slot
        auto issue_rows = [&]() __attribute__((always_inline)) {
            if (islab == 0) set_tile(itile);
            stage(c_base, off_c, ibuf, islab);
            ibuf = ibuf == 2 ? 0 : ibuf + 1;
            if (++islab == nslab) { islab = 0; next_tile(itile, ifrac); }
        };
        issue_rows();
        stage(q_base, off_q, 3, 0);
        if (total_steps > 1) issue_rows();
        int qslab = 1 % nslab;                                      // query slab of step + 1
        for (int step = 0; step < total_steps; ++step) {
            if (step + 2 > total_steps - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (LT_NP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (LT_NP == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __syncthreads();                                        // step's slabs are there; the slots of step - 1 are free
            if (step + 1 < total_steps) stage(q_base, off_q, 3 + ((step & 1) ^ 1), qslab);
            if (step + 2 < total_steps) issue_rows();               // into slot (step + 2) % 3, read last in step - 1
            qslab = qslab + 1 == nslab ? 0 : qslab + 1;
        }
        return;                                                     // (the matrix waves' epilogues hold no barrier)
    }

    // ---- matrix waves: 4 query groups of 64 (two MFMA B operands: queries r and 32 + r of the group) x 2 halves of
    // the row tile (128 rows = four 32-row blocks): every fragment read from LDS feeds two MFMAs (24 ds_read_b128 per
    // slab and wave for 32 MFMAs; 32 queries x 256 rows per wave would need 36)
    const int wq = wave & 3, wr = wave >> 2;
    const int q0 = g * LT_Q + wq * 64 + r, q1 = q0 + 32;
    const float thr0 = q0 < nq ? A.thr[q0] : INFINITY;          // padding lanes never keep anything
    const float thr1 = q1 < nq ? A.thr[q1] : INFINITY;
    const f32x16 zero = {0};
    f32x16 acc[4][2];
    const int keyq0 = ((wq * 64 + r) >> 1) & 7, keyq1 = ((wq * 64 + 32 + r) >> 1) & 7;
    int step = 0, bufa = 0;                                     // bufa = step % 3
    for (int j = j_lo; j < j_hi; ++j) {
#pragma unroll
        for (int b = 0; b < 4; ++b) { acc[b][0] = zero; acc[b][1] = zero; }
        for (int s = 0; s < nslab; ++s, ++step) {
            const int bufb = step & 1;
            __syncthreads();                                    // the loaders have seen this step's slabs land
            const char* rows = smem + bufa * LT_HALF;
            const char* qs = smem + (3 + bufb) * LT_HALF;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(qs + (wq * 64 + r) * LT_BK + ((2 * u + h) ^ keyq0) * 16);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(qs + (wq * 64 + 32 + r) * LT_BK + ((2 * u + h) ^ keyq1) * 16);
                f32x4 a[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int t = wr * 128 + b * 32 + r;
                    a[b] = *reinterpret_cast<const f32x4*>(rows + t * LT_BK + ((2 * u + h) ^ ((t >> 1) & 7)) * 16);
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if constexpr (DT == DT_F16) {
                        acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[b]), __builtin_bit_cast(f16x8, b0), acc[b][0], 0, 0, 0);
                        acc[b][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[b]), __builtin_bit_cast(f16x8, b1), acc[b][1], 0, 0, 0);
                    } else {
                        acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[b]), __builtin_bit_cast(bf16x8, b0), acc[b][0], 0, 0, 0);
                        acc[b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[b]), __builtin_bit_cast(bf16x8, b1), acc[b][1], 0, 0, 0);
                    }
                }
            }
            bufa = bufa == 2 ? 0 : bufa + 1;
        }
        // ---- tile epilogue: acc[b][s][jj] is (corpus row tile * 256 + 128 wr + 32 b + (jj & 3) + 8 (jj >> 2) + 4 h,
        // query r (s = 0) / 32 + r (s = 1) of the wave's 64)
        float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int jj = 0; jj < 16; jj += 2) {
                m0 = vmax3(m0, acc[b][0][jj], acc[b][0][jj + 1]);
                m1 = vmax3(m1, acc[b][1][jj], acc[b][1][jj + 1]);
            }
        }
        if (!A.dense) {
            // the last level: a lane keeps a row of a tile now and then -- one atomic add per kept row
            if (__builtin_amdgcn_ballot_w64(m0 > thr0 || m1 > thr1) != 0) {
                const int row_base = tile * LT_ROWS + wr * 128 + 4 * h;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) {
                        const int row = row_base + 32 * b + (jj & 3) + 8 * (jj >> 2);
                        if (acc[b][0][jj] > thr0 && row < n) {
                            const unsigned pos = atomicAdd(A.cnt + q0, 1u);
                            if (pos < (unsigned)A.cap) A.cand[(size_t)q0 * A.cap + pos] = make_key(acc[b][0][jj], row);
                        }
                        if (acc[b][1][jj] > thr1 && row < n) {
                            const unsigned pos = atomicAdd(A.cnt + q1, 1u);
                            if (pos < (unsigned)A.cap) A.cand[(size_t)q1 * A.cap + pos] = make_key(acc[b][1][jj], row);
                        }
                    }
                }
            }
        } else if (__builtin_amdgcn_ballot_w64(m0 > thr0 || m1 > thr1) != 0) {
            // The sample levels keep many rows per lane and tile (the first one EVERY row: threshold -inf) -- one atomic
            // add per lane and query for all of them (count, reserve, store); one per row made the first level cost
            // four times its matrix work.  (On the last level the counting pass costs more than it saves: +9 %.)
            const int row_base = tile * LT_ROWS + wr * 128 + 4 * h;
            unsigned n0 = 0u, n1 = 0u;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const bool in = row_base + 32 * b + (jj & 3) + 8 * (jj >> 2) < n;
                    n0 += (acc[b][0][jj] > thr0 && in) ? 1u : 0u;
                    n1 += (acc[b][1][jj] > thr1 && in) ? 1u : 0u;
                }
            }
            unsigned at0 = 0u, at1 = 0u;
            if (n0) at0 = atomicAdd(A.cnt + q0, n0);
            if (n1) at1 = atomicAdd(A.cnt + q1, n1);
            unsigned long long* c0 = A.cand + (size_t)q0 * A.cap;
            unsigned long long* c1 = A.cand + (size_t)q1 * A.cap;
            if (__builtin_amdgcn_ballot_w64(n0 != 0u) != 0) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) {
                        const int row = row_base + 32 * b + (jj & 3) + 8 * (jj >> 2);
                        if (acc[b][0][jj] > thr0 && row < n) {
                            if (at0 < (unsigned)A.cap) c0[at0] = make_key(acc[b][0][jj], row);
                            ++at0;
                        }
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(n1 != 0u) != 0) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) {
                        const int row = row_base + 32 * b + (jj & 3) + 8 * (jj >> 2);
                        if (acc[b][1][jj] > thr1 && row < n) {
                            if (at1 < (unsigned)A.cap) c1[at1] = make_key(acc[b][1][jj], row);
                            ++at1;
                        }
                    }
                }
            }
        }
        next_tile(tile, tile_frac);
    }
}

// ------------------------------------------------------------------------------ host side
constexpr int LONG_CAP = 8192;      // rows kept per query (k_select_all sorts them in 64 KB of LDS)
constexpr int LONG_MAX_K = 1024;    // what the exhaustive kernels -- the path of a query left at status 1 -- can resolve
// k_select_all keeps 2 x cap keys + the exact query row + its staging tile in LDS: rows beyond 10240 bytes
// (f32 d > 2560, bf16 d > 5120) leave room for half the capacity only.
static int long_cap(int d, int exact_dtype) { return d * elem_bytes(exact_dtype) > 10240 ? LONG_CAP / 2 : LONG_CAP; }
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

static bool long_shape_ok(int d, int exact_dtype, int scan_dtype) {
    if (d <= 0 || d % 64 || d * elem_bytes(exact_dtype) > 16384) return false;     // (k_select_all keeps the query row in LDS)
    return (exact_dtype == DT_F32 && scan_dtype == DT_F16) || (exact_dtype == DT_BF16 && scan_dtype == DT_BF16);
}

size_t ip_topk_long_workspace_bytes(long nq, long n, int d, int dtype) {
    const int scan = dtype == DT_F32 ? DT_F16 : DT_BF16;
    if (nq <= 0 || n <= 0 || !long_shape_ok(d, dtype, scan)) return 0;
    return al256((size_t)nq * d * 2) + al256((size_t)nq * 4) * 3 + al256((size_t)nq * 16) + (size_t)nq * LONG_CAP * 8;
}


int ip_topk_long(const void* q, long nq, const void* c_exact, int exact_dtype, const void* c_scan, int corpus_shift,
                 float corpus_resid, long n, int d, int k, long id_offset, float corpus_max_norm, float* D_out, long* I_out,
                 int* status, void* ws, size_t ws_bytes, hipStream_t st) {
    const int scan_dtype = exact_dtype == DT_F32 ? DT_F16 : DT_BF16;
    if (nq <= 0 || n <= 0 || k <= 0) { set_error("ip_topk_long: nq, n, k must be positive"); return SSS_EINVAL; }
    if (!long_shape_ok(d, exact_dtype, scan_dtype)) { set_error("ip_topk_long: need dtype 0 / 1, d %% 64 == 0 and rows of at most 16384 bytes (got dtype %d d %d)", exact_dtype, d); return SSS_EINVAL; }
    if (!c_scan || (reinterpret_cast<uintptr_t>(c_scan) & 15)) { set_error("ip_topk_long: scan image missing or not 16-byte aligned"); return SSS_EINVAL; }
    if (n >= (1L << 31) - 1024 || nq >= (1L << 31)) { set_error("ip_topk_long: n and nq must be < 2^31"); return SSS_EINVAL; }
    if (k > LONG_MAX_K) { set_error("ip_topk_long: k too large (max %d)", LONG_MAX_K); return SSS_EINVAL; }
    const int cap = long_cap(d, exact_dtype);
    if (reinterpret_cast<uintptr_t>(ws) & 255) { set_error("ip_topk_long: workspace must be 256-byte aligned"); return SSS_EINVAL; }
    const size_t need = ip_topk_long_workspace_bytes(nq, n, d, exact_dtype);
    if (ws_bytes < need) { set_error("ip_topk_long: workspace %zu < %zu", ws_bytes, need); return SSS_EWORKSPACE; }
    char* w = reinterpret_cast<char*>(ws);
    void* qimg = w;                              w += al256((size_t)nq * d * 2);
    int* qsel = reinterpret_cast<int*>(w);       w += al256((size_t)nq * 4);
    float* thr = reinterpret_cast<float*>(w);    w += al256((size_t)nq * 4);
    unsigned* cnt = reinterpret_cast<unsigned*>(w); w += al256((size_t)nq * 4);
    double* qb = reinterpret_cast<double*>(w);   w += al256((size_t)nq * 16);       // (error bound, unscale) per query: computed once
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(w);

    // (the f16 query image, the identity selection, D_out at "no bound known", thresholds, counters, status and the
    //  per-query bound cache are written by ONE launch below, once the ThrArgs are filled: launch_long_setup)
    const void* q_scan = scan_dtype == DT_F16 ? qimg : q;
    int rc = SSS_OK;

    // ---- levels (descending in level_tiles, run in reverse).  The first sample is as large as the capacity allows with
    // the threshold at -inf (cap rows: every sampled row is kept) -- a 32-tile level costs the time of ONE tile, the chip
    // being far from full, and its k-th best score is a much better bound than that of a 2k-row sample.  From there to
    // the whole corpus the tile counts grow by equal factors r <= fmax = cap / (4 k): about k * r rows pass a level's
    // threshold (2x margin below the capacity at r = fmax), the fewest levels that allows, and the factors balanced so
    // that the last sample is as small as it can be (1M x 1600, K = 100: 32, 353, 3907 tiles -- the samples add ~10 % to
    // the matrix work; it was 1, 24, 488, 3907).  Intermediate levels read no row at all (k_bound_prepare).
    const int total_tiles = (int)((n + LT_ROWS - 1) / LT_ROWS);
    int fmax = cap / (4 * k);
    if (fmax < 2) fmax = 2;
    int first = cap / LT_ROWS;
    if (first > total_tiles) first = total_tiles;
    int level_tiles[40];
    int levels = 0;
    level_tiles[levels++] = total_tiles;
    if (first < total_tiles) {
        const double span = (double)total_tiles / (double)first;
        int steps = 1;
        while (pow((double)fmax, (double)steps) < span && steps < 36) ++steps;      // growth steps from `first` to the corpus
        const double r = pow(span, 1.0 / (double)steps);
        for (int i = steps - 1; i >= 1; --i) {
            int tl = (int)((double)first * pow(r, (double)i) + 0.5);
            if (tl <= first) break;
            if (tl < level_tiles[levels - 1]) level_tiles[levels++] = tl;
        }
        level_tiles[levels++] = first;
    }
    ThrArgs t;
    t.Q = q; t.C = c_exact; t.qsel = qsel; t.nsel = (int)nq; t.d = d; t.dtype = exact_dtype; t.k = k; t.cap = cap;
    t.scan_dtype = scan_dtype; t.corpus_shift = corpus_shift; t.corpus_resid = corpus_resid; t.corpus_max_norm = corpus_max_norm;
    t.id_offset = id_offset; t.thr = thr; t.cnt = cnt; t.cand = cand; t.D_out = D_out; t.I_out = I_out; t.status = status;
    t.qb = qb; t.qb_ready = 0;
    rc = launch_long_setup(t, qsel, scan_dtype == DT_F16 ? qimg : nullptr, st);
    if (rc) return rc;
    t.qb_ready = 1;
    LongArgs a;
    a.Qimg = q_scan; a.C = c_scan; a.nq = (int)nq; a.n = (int)n; a.d = d; a.G = (int)((nq + LT_Q - 1) / LT_Q);
    a.total_tiles = total_tiles; a.cap = cap; a.thr = thr; a.cnt = cnt; a.cand = cand;
    // groups per XCD (block map of k_scan_long): all of them (fewer measured slower, see the kernel); the switch stays
    // for A/B runs
    a.gpx = a.G;
    if (a.G == 2 || a.G == 4 || a.G == 8) {
        static const int gpx_env = getenv("SSS_LONG_GPX") ? atoi(getenv("SSS_LONG_GPX")) : 0;
        if (gpx_env > 0 && a.G % gpx_env == 0 && 8 % (a.G / gpx_env) == 0) a.gpx = gpx_env;
    }
    static bool attr_done[MAX_DEVICES][2] = {};
    const int dev = current_device();
    // DISJOINT LEVELS (three levels and more).  The last sample used to be scanned twice: once as a sample, once more as
    // part of the whole corpus -- 9 % of the matrix work at 1M x 1600, K = 100.  Now the last sample takes EVERY R-th
    // tile (R = the planned ratio, rounded, >= 2) and the final level only the tiles in between: the rows the sample kept
    // stay in the query's array, pruned by k_thr_prepare to those that pass the final threshold (its `keep` mode; the
    // final threshold is the tighter one, so nothing above it was ever dropped), and the final level appends behind
    // them.  Together they are exactly the rows of the whole corpus above the final threshold -- what k_select_all's
    // proof needs -- and there are fewer of them (the sample's rows are filtered by the final threshold now).
    int R = 0;
    if (levels >= 3) {
        R = (int)((double)total_tiles / (double)level_tiles[1] + 0.5);
        if (R < 2) R = 0;                              // (ratio below 1.5: the planned sample is most of the corpus -- keep the plain form)
    }
    for (int lv = levels - 1; lv >= 0; --lv) {
        int tiles = level_tiles[lv];
        const bool last = lv == 0;
        a.total_tiles = total_tiles;
        a.skip_R = 0;
        if (R > 0 && lv == 1) {                        // every R-th tile: the proportional map with total = R * count
            tiles = (total_tiles + R - 1) / R;
            a.total_tiles = R * tiles;
        } else if (R > 0 && last) {                    // the tiles in between
            tiles = total_tiles - (total_tiles + R - 1) / R;
            a.skip_R = R;
        }
        a.tile_count = tiles;
        a.dense = last ? 0 : 1;
        int S = (256 / a.G) & ~7;
        if (S < 8) S = 8;
        while (S > 8 && S > tiles) S -= 8;
        a.S = S;
        a.tiles_per_split = (tiles + S - 1) / S;
        t.n = n;
        // (thresholds and counters of this level: written by launch_long_setup -- first level -- or by the previous level's
        //  launch_bound_prepare)
        const size_t lds = LT_LDS_BYTES;
        const int ti = scan_dtype == DT_F16 ? 0 : 1;
        if (!attr_done[dev][ti]) {
            if (ti == 0) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan_long<DT_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan_long<DT_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_done[dev][ti] = true;
        }
        if (ti == 0) hipLaunchKernelGGL(k_scan_long<DT_F16>, dim3(a.S * a.G), dim3(LT_THREADS), lds, st, a);
        else hipLaunchKernelGGL(k_scan_long<DT_BF16>, dim3(a.S * a.G), dim3(LT_THREADS), lds, st, a);
        rc = check_launch("k_scan_long");
        if (rc) return rc;
        if (last) rc = launch_select_all(t, st);
        else {
            // bound from this level's kept scan scores -> the next level's thresholds; the final level of a disjoint
            // schedule keeps this (the last sample's) rows that pass them
            t.keep = (R > 0 && lv == 1) ? 1 : 0;
            rc = launch_bound_prepare(t, st);
        }
        if (rc) return rc;
    }
    return SSS_OK;
}

}  // namespace sss

// Internal interface between the scan kernel (scan.hip), the select / re-score kernels
// (select.hip) and the host orchestration of sss_ip_topk (ip_topk.hip).  gfx950 only.
#pragma once
#include "sss_common.h"

namespace sss {

constexpr int DT_F32 = 0;       // element type codes of the C ABI (include/sss.h: dtype)
constexpr int DT_BF16 = 1;
// Scan-only element type (never crosses the C ABI as a corpus dtype): an f32 corpus row of d
// elements stored as [hi(d) | lo(d)] bfloat16, hi = rne_bf16(x), lo = rne_bf16(x - hi) -- 4 bytes
// per element like f32.  The scan scores hi*hi + hi*lo + lo*hi on the bf16 MFMA (three passes at
// 16x the f32 MFMA rate); queries are f32 and are split by the kernel.  Candidates are re-scored
// from the f32 rows, and the proof uses the split's own error bound (select.hip: err_bound).
constexpr int DT_SPLIT = 2;
// Scan-only element type: an f32 corpus stored as float16 after an exact power-of-two scaling,
// x * 2^shift with ONE shift for the whole corpus chosen so that the largest |element| lies in
// [2^12, 2^13) (f16_shift below: far from both ends of the f16 range) -- 2 bytes per element.  The
// kernel scales each f32 query by its own power of two the same way, so a scan score is the true
// score times 2^(corpus shift + query shift): thresholds and candidate selection work in that
// domain (per query it is a fixed positive factor), and the select kernel divides it out for the
// proof.  One f16 MFMA pass (1/3 of DT_SPLIT's matrix work, half its bytes) with a coarser bound
// (select.hip: err_bound ~ 2^-10 |q||c|), still proven per query and re-scored from the f32 rows.
constexpr int DT_F16 = 3;

// shift that maps a largest magnitude `amax` into [2^12, 2^13); 0 for an all-zero / non-finite row
__host__ __device__ inline int f16_shift(float amax) {
    if (!(amax > 0.f) || !(amax <= 3.4028234663852886e38f)) return 0;   // zero, NaN or inf
    int e;
    (void)frexpf(amax, &e);          // amax = m * 2^e, m in [0.5, 1)
    return 13 - e;
}

constexpr int KP = 16;          // per-lane candidate list length (register resident)
constexpr int WG_QUERIES = 256; // queries per scan workgroup (8 waves x 32)
constexpr int MAX_SLOTS = 128;  // admission-threshold slots per query (J <= MAX_SLOTS)
// A query's slot words start SLOT_STRIDE words apart whatever J is: the 1024 lines of a 1024-query batch then spread
// over 512 KB of address space -- and with it over the memory channels -- instead of sitting in 64 contiguous KB that
// every workgroup of the launch polls, fetches and hits with agent-scope atomics at the same moment (the bootstrap).
#ifndef SSS_SLOT_STRIDE
#define SSS_SLOT_STRIDE MAX_SLOTS
#endif
constexpr int SLOT_STRIDE = SSS_SLOT_STRIDE;
constexpr unsigned ORD_NEG_INF = 0x007FFFFFu;   // f2ord(-inf); slot value 0 = "never written"

static inline int elem_bytes(int dtype) { return (dtype == DT_BF16 || dtype == DT_F16) ? 2 : 4; }

// STATE words (caller-owned, zero before the first call; every call leaves them zero: the select
// kernel, their last reader, clears what the call used -- no per-call memset launch).  Because the
// whole buffer is zero between calls, each call may lay it out as it likes:
//   slots   u32 [nq][SLOT_STRIDE]   admission-threshold slots: the first J words of a query's 512 bytes (J = 16: one 64-byte line)
//   cnt     u32 [nq]      candidates written per query        (at word nq * MAX_SLOTS)
//   maxlast u64 [nq]      largest tail key over FULL lists    (8-byte aligned, after cnt)
static inline size_t state_off_cnt(long nq) { return (size_t)nq * MAX_SLOTS; }                       // in words
static inline size_t state_off_maxlast(long nq) { return (state_off_cnt(nq) + (size_t)nq + 1) & ~(size_t)1; }
static inline size_t state_words(long nq) { return state_off_maxlast(nq) + 2 * (size_t)nq; }

// Per-search plan (host).  Workspace: cand u64 [nq][cap] compacted candidate keys, cap = L * KP.
struct ScanPlan {
    int G, S, L, K2, J, Ju, cert, boot, append, tile_rows;      // J slots per query = Ju classes; boot: the first tile of a split is scanned twice (max-only first)
    int tau_skip;                                               // cert == 1: the threshold is the (tau_skip + 1)-th smallest slot (16 - K2)
    int total_tiles, tiles_per_split, cap;
    size_t total_bytes;
};

ScanPlan make_plan(long nq, long n, int d, int k, int dtype);

struct ScanArgs {
    const void* Q;
    const void* C;
    int nq, n, tiles_per_split, total_tiles, S, G, J, Ju, cert, boot, append, cap;
    int tau_skip = 0;
    unsigned* slots;                // the three arrays live in the caller's state buffer
    unsigned* cnt;
    unsigned long long* maxlast;
    unsigned long long* cand;
    // threshold form only (k_scan<..., THR = true>): the compact query list and its per-query thresholds
    const int* qsel = nullptr;
    const float* thr = nullptr;
};

ScanPlan make_thr_plan(long nsel, long n, int d, int scan_dtype, int cap);
int launch_scan(int dtype, int d, int tile_rows, const ScanArgs& a, hipStream_t st);
int scan_boot_expired(int reset);

struct SelectArgs {
    const void* Q;
    const void* C;
    int nq, d, dtype, k, K2, J, cap;
    int tau_skip = 0;               // the scan's rank-selected threshold: (tau_skip + 1)-th smallest of the 16 slots
    int scan_dtype;                 // what produced the candidates (DT_F32 / DT_BF16 / DT_SPLIT / DT_F16): picks the error bound
    int corpus_shift;               // DT_F16: the corpus image is corpus * 2^corpus_shift (else 0)
    float corpus_resid;             // DT_F16: largest row norm of (image * 2^-corpus_shift - corpus)
    const unsigned long long* cand;
    unsigned* slots;                // state arrays: read, then cleared
    unsigned* cnt;
    unsigned long long* maxlast;
    long id_offset;
    float corpus_max_norm;
    float* D_out;
    long* I_out;
    int* status;
    int* unproven_count;        // optional device counter: += 1 per query left unproven
};

int launch_select(const SelectArgs& a, hipStream_t st);

// Threshold rung (select.hip: k_thr_prepare, k_select_all; scan.hip: k_scan<..., THR = true>)
struct ThrArgs {
    const void* Q;                  // all queries [*, d] of the exact element type
    const void* C;                  // the stored rows (re-score)
    const int* qsel;                // [nsel] query rows to resolve
    int nsel, d, dtype, k, cap;
    long n;
    int scan_dtype, corpus_shift;
    float corpus_resid, corpus_max_norm;
    long id_offset;
    float* thr;                     // [nsel]   workspace
    unsigned* cnt;                  // [nsel]   workspace
    const unsigned long long* cand; // [nsel][cap] workspace
    float* D_out;                   // [nq, k]: row q's k-th entry is read (lower bound), rows of resolved queries are rewritten
    long* I_out;
    int* status;                    // [nq]: set to 0 for resolved queries
    double* qb = nullptr;           // optional [nsel][2] cache of the queries' (error bound, unscale): written by the first kernel
    int qb_ready = 0;               //   that computes them (qb_ready == 0), read by the later ones (sss_ip_topk_long: five kernels a search)
    int keep = 0;                   // k_thr_prepare: 1 = keep the rows already kept that pass the NEW threshold (compacted in place)
                                    //                instead of starting from an empty array (sss_ip_topk_long: disjoint levels)
};
int launch_thr_prepare(const ThrArgs& a, hipStream_t st);
int launch_select_all(const ThrArgs& a, hipStream_t st);
// sss_ip_topk_long's fused steps: everything a search needs before its first scan in ONE launch (f16 query image when
// `qimg` is given, identity selection, D_out rows at -FLT_MAX, thresholds -inf, counters 0, status 1, the per-query
// (error bound, unscale) cache), and the step between two levels in one launch (bound from the level just scanned ->
// column k-1 of D_out -> the next level's threshold; counters zeroed, or -- a.keep -- the kept rows pruned in place).
int launch_long_setup(const ThrArgs& a, int* qsel, void* qimg, hipStream_t st);
int launch_bound_prepare(const ThrArgs& a, hipStream_t st);

}  // namespace sss

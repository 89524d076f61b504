// Host orchestration of sss_ip_topk / sss_ip_topk_f16 / sss_ip_topk_split: plan -> k_scan (over the
// corpus or its f16 / split image) -> k_select_* (over the stored rows): two launches; the per-query
// state words are handed back zeroed by the select kernel, so there is no per-call memset.
// (reference call site: `D, I = index.search(normalize(emb), K)`, test_amazon_filterd.py:578.)
#include "scan.h"

namespace sss {

// Optional timing of the dominant kernel (bench.py roofline leg): when enabled, every k_scan
// launch is bracketed by a hipEvent pair on ITS stream; profile_read() drains the ring of the
// calling thread's current device.
namespace {
constexpr int PROF_RING = 512;
struct Prof {
    bool on = false;
    int n = 0;
    hipEvent_t ev[2 * PROF_RING];
    bool made = false;
};
Prof g_prof[MAX_DEVICES];
}  // namespace

int profile_enable(int on) {
    Prof& p = g_prof[current_device()];
    if (on && !p.made) {
        for (int i = 0; i < 2 * PROF_RING; ++i)
            if (hipEventCreate(&p.ev[i]) != hipSuccess) { set_error("profile_enable: hipEventCreate failed"); return SSS_EHIP; }
        p.made = true;
    }
    p.on = on != 0;
    p.n = 0;
    return SSS_OK;
}

int profile_read(double* total_ms, int* launches) {
    Prof& p = g_prof[current_device()];
    double sum = 0.0;
    for (int i = 0; i < p.n; ++i) {
        if (hipEventSynchronize(p.ev[2 * i + 1]) != hipSuccess) { set_error("profile_read: sync failed"); return SSS_EHIP; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]) != hipSuccess) { set_error("profile_read: elapsed failed"); return SSS_EHIP; }
        sum += ms;
    }
    *total_ms = sum;
    *launches = p.n;
    p.n = 0;
    return SSS_OK;
}

static bool fused_shape_ok(int d, int dtype) {
    const int rb = d * elem_bytes(dtype);
    return (dtype == DT_F32 || dtype == DT_BF16 || dtype == DT_SPLIT || dtype == DT_F16) && (rb == 256 || rb == 512 || rb == 1024);
}

size_t ip_topk_state_bytes(long nq) { return nq > 0 ? state_words(nq) * 4 : 0; }

size_t ip_topk_workspace_bytes(long nq, long n, int d, int k, int dtype) {      // dtype: the C ABI's (0 / 1)
    if (nq <= 0 || n <= 0 || k <= 0 || (dtype != DT_F32 && dtype != DT_BF16) || !fused_shape_ok(d, dtype)) return 0;
    return make_plan(nq, n, d, k, dtype).total_bytes;
}

size_t ip_topk_scan_workspace_bytes(long nq, long n, int d, int k, int scan_dtype) {   // scan.h codes (0..3)
    if (nq <= 0 || n <= 0 || k <= 0 || !fused_shape_ok(d, scan_dtype)) return 0;
    return make_plan(nq, n, d, k, scan_dtype).total_bytes;
}

// scan_dtype: what k_scan reads at c_scan (DT_F32 / DT_BF16: the corpus itself; DT_SPLIT: the
// [hi | lo] bf16 image of an f32 corpus; DT_F16: its scaled f16 image, corpus * 2^corpus_shift);
// c_exact / exact_dtype: the rows the candidates are re-scored from (and the element type of q).
static int ip_topk_impl(const void* q, long nq, const void* c_scan, int scan_dtype, int corpus_shift, float corpus_resid,
                        const void* c_exact,
                        int exact_dtype, long n, int d, int k, long id_offset, float corpus_max_norm, float* D_out, long* I_out,
                        int* status, int* unproven_count, void* state, size_t state_bytes, void* ws, size_t ws_bytes,
                        hipStream_t st) {
    if (nq <= 0 || n <= 0 || k <= 0) { set_error("ip_topk: nq, n, k must be positive"); return SSS_EINVAL; }
    if (!fused_shape_ok(d, scan_dtype)) {
        set_error("ip_topk: need dtype 0 (f32, d in {64,128,256}) or 1 (bf16, d in {128,256,512}); got dtype %d d %d", exact_dtype, d);
        return SSS_EINVAL;
    }
    if (n >= (1L << 31) - 1024 || nq >= (1L << 31)) { set_error("ip_topk: n and nq must be < 2^31 per shard"); return SSS_EINVAL; }
    if (k > 500) { set_error("ip_topk: k too large (max 500)"); return SSS_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(ws) & 255) || (reinterpret_cast<uintptr_t>(state) & 15)) {
        set_error("ip_topk: workspace must be 256-byte aligned, state 16-byte aligned");
        return SSS_EINVAL;
    }
    if (!state || state_bytes < ip_topk_state_bytes(nq)) { set_error("ip_topk: state %zu < %zu bytes", state_bytes, ip_topk_state_bytes(nq)); return SSS_EWORKSPACE; }
    const ScanPlan p = make_plan(nq, n, d, k, scan_dtype);
    if (ws_bytes < p.total_bytes) { set_error("ip_topk: workspace %zu < %zu", ws_bytes, p.total_bytes); return SSS_EWORKSPACE; }
    char* w = reinterpret_cast<char*>(ws);

    ScanArgs a;
    a.Q = q; a.C = c_scan; a.nq = (int)nq; a.n = (int)n;
    a.tiles_per_split = p.tiles_per_split; a.total_tiles = p.total_tiles;
    a.S = p.S; a.G = p.G; a.J = p.J; a.Ju = p.Ju; a.cert = p.cert; a.boot = p.boot; a.append = p.append; a.cap = p.cap;
    a.tau_skip = p.tau_skip;
    unsigned* sw = reinterpret_cast<unsigned*>(state);
    a.slots = sw;
    a.cnt = sw + state_off_cnt(nq);
    a.maxlast = reinterpret_cast<unsigned long long*>(sw + state_off_maxlast(nq));
    a.cand = reinterpret_cast<unsigned long long*>(w);
    Prof& pr = g_prof[current_device()];
    const bool prof = pr.on && pr.n < PROF_RING;
    if (prof) (void)hipEventRecord(pr.ev[2 * pr.n], st);
    int rc = launch_scan(scan_dtype, d, p.tile_rows, a, st);
    if (prof) { (void)hipEventRecord(pr.ev[2 * pr.n + 1], st); ++pr.n; }
    if (rc) return rc;                 // nothing ran: the state is still clean

    SelectArgs s;
    s.Q = q; s.C = c_exact; s.nq = (int)nq; s.d = d; s.dtype = exact_dtype; s.scan_dtype = scan_dtype; s.corpus_shift = corpus_shift; s.corpus_resid = corpus_resid;
    s.k = k; s.K2 = p.K2; s.J = p.J; s.cap = p.cap; s.tau_skip = p.tau_skip;
    s.cand = a.cand; s.slots = a.slots; s.cnt = a.cnt; s.maxlast = a.maxlast;
    s.id_offset = id_offset; s.corpus_max_norm = corpus_max_norm;
    s.D_out = D_out; s.I_out = I_out; s.status = status; s.unproven_count = unproven_count;
    rc = launch_select(s, st);
    if (rc) (void)hipMemsetAsync(state, 0, ip_topk_state_bytes(nq), st);   // the scan dirtied it and nobody will clear it
    return rc;
}

// Threshold rung for the queries `qsel` a fused search left unproven (select.hip: THRESHOLD RUNG).
// Workspace: thr f32 [nsel] | cnt u32 [nsel] | (256-byte aligned) cand u64 [nsel][cap].
constexpr int THR_CAP = 8192;       // rows kept per query (64 KB of keys in LDS for the sort)
static size_t thr_head_bytes(long nsel) { return ((size_t)nsel * 8 + 255) & ~(size_t)255; }

size_t ip_topk_threshold_workspace_bytes(long nsel, long n, int d, int scan_dtype) {
    if (nsel <= 0 || n <= 0 || !fused_shape_ok(d, scan_dtype)) return 0;
    return thr_head_bytes(nsel) + make_thr_plan(nsel, n, d, scan_dtype, THR_CAP).total_bytes;
}

int ip_topk_threshold(const void* q, const int* qsel, long nsel, const void* c_exact, int exact_dtype, const void* c_scan,
                      int scan_dtype, int corpus_shift, float corpus_resid, long n, int d, int k, long id_offset,
                      float corpus_max_norm, float* D_out, long* I_out, int* status, void* ws, size_t ws_bytes, hipStream_t st) {
    if (nsel <= 0 || n <= 0 || k <= 0 || !qsel) { set_error("ip_topk_threshold: nsel, n, k must be positive"); return SSS_EINVAL; }
    if (exact_dtype != DT_F32 && exact_dtype != DT_BF16) { set_error("ip_topk_threshold: dtype must be 0 (f32) or 1 (bf16)"); return SSS_EINVAL; }
    const bool native = scan_dtype == exact_dtype;
    if (!fused_shape_ok(d, scan_dtype) || (!native && (exact_dtype != DT_F32 || (scan_dtype != DT_SPLIT && scan_dtype != DT_F16)))) {
        set_error("ip_topk_threshold: no scan of type %d for dtype %d, d %d", scan_dtype, exact_dtype, d);
        return SSS_EINVAL;
    }
    if (!c_scan || (reinterpret_cast<uintptr_t>(c_scan) & 15)) { set_error("ip_topk_threshold: scan image missing or not 16-byte aligned"); return SSS_EINVAL; }
    if (n >= (1L << 31) - 1024 || nsel >= (1L << 31)) { set_error("ip_topk_threshold: n and nsel must be < 2^31"); return SSS_EINVAL; }
    if (k > THR_CAP) { set_error("ip_topk_threshold: k too large (max %d)", THR_CAP); return SSS_EINVAL; }
    if (reinterpret_cast<uintptr_t>(ws) & 255) { set_error("ip_topk_threshold: workspace must be 256-byte aligned"); return SSS_EINVAL; }
    const ScanPlan p = make_thr_plan(nsel, n, d, scan_dtype, THR_CAP);
    if (ws_bytes < thr_head_bytes(nsel) + p.total_bytes) { set_error("ip_topk_threshold: workspace %zu < %zu", ws_bytes, thr_head_bytes(nsel) + p.total_bytes); return SSS_EWORKSPACE; }
    char* w = reinterpret_cast<char*>(ws);
    ThrArgs t;
    t.Q = q; t.C = c_exact; t.qsel = qsel; t.nsel = (int)nsel; t.d = d; t.dtype = exact_dtype; t.k = k; t.cap = p.cap; t.n = n;
    t.scan_dtype = scan_dtype; t.corpus_shift = corpus_shift; t.corpus_resid = corpus_resid; t.corpus_max_norm = corpus_max_norm;
    t.id_offset = id_offset;
    t.thr = reinterpret_cast<float*>(w);
    t.cnt = reinterpret_cast<unsigned*>(w + (size_t)nsel * 4);
    t.cand = reinterpret_cast<unsigned long long*>(w + thr_head_bytes(nsel));
    t.D_out = D_out; t.I_out = I_out; t.status = status;
    int rc = launch_thr_prepare(t, st);
    if (rc) return rc;
    ScanArgs a = {};
    a.Q = q; a.C = c_scan; a.nq = (int)nsel; a.n = (int)n;
    a.tiles_per_split = p.tiles_per_split; a.total_tiles = p.total_tiles;
    a.S = p.S; a.G = p.G; a.J = 0; a.Ju = 0; a.cert = 1; a.boot = 0; a.append = 0; a.cap = p.cap;
    a.slots = nullptr; a.cnt = t.cnt; a.maxlast = nullptr;
    a.cand = const_cast<unsigned long long*>(t.cand);
    a.qsel = qsel; a.thr = t.thr;
    rc = launch_scan(scan_dtype, d, p.tile_rows, a, st);
    if (rc) return rc;
    return launch_select_all(t, st);
}

int ip_topk(const void* q, long nq, const void* c, long n, int d, int k, int dtype, long id_offset,
            float corpus_max_norm, float* D_out, long* I_out, int* status, int* unproven_count, void* state,
            size_t state_bytes, void* ws, size_t ws_bytes, hipStream_t st) {
    if (dtype != DT_F32 && dtype != DT_BF16) { set_error("ip_topk: dtype must be 0 (f32) or 1 (bf16), got %d", dtype); return SSS_EINVAL; }
    return ip_topk_impl(q, nq, c, dtype, 0, 0.f, c, dtype, n, d, k, id_offset, corpus_max_norm, D_out, I_out, status,
                        unproven_count, state, state_bytes, ws, ws_bytes, st);
}

int ip_topk_split(const float* q, long nq, const float* c, const void* c_split, long n, int d, int k, long id_offset,
                  float corpus_max_norm, float* D_out, long* I_out, int* status, int* unproven_count, void* state,
                  size_t state_bytes, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!c_split || (reinterpret_cast<uintptr_t>(c_split) & 15)) { set_error("ip_topk_split: split image missing or not 16-byte aligned"); return SSS_EINVAL; }
    return ip_topk_impl(q, nq, c_split, DT_SPLIT, 0, 0.f, c, DT_F32, n, d, k, id_offset, corpus_max_norm, D_out, I_out, status,
                        unproven_count, state, state_bytes, ws, ws_bytes, st);
}

int ip_topk_f16(const float* q, long nq, const float* c, const void* c_f16, int corpus_shift, float corpus_resid, long n, int d, int k,
                long id_offset, float corpus_max_norm, float* D_out, long* I_out, int* status, int* unproven_count,
                void* state, size_t state_bytes, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!c_f16 || (reinterpret_cast<uintptr_t>(c_f16) & 15)) { set_error("ip_topk_f16: f16 image missing or not 16-byte aligned"); return SSS_EINVAL; }
    if (corpus_shift < -160 || corpus_shift > 160) { set_error("ip_topk_f16: corpus_shift out of range"); return SSS_EINVAL; }
    if (!(corpus_resid >= 0.f)) { set_error("ip_topk_f16: corpus_resid_norm must be >= 0"); return SSS_EINVAL; }
    return ip_topk_impl(q, nq, c_f16, DT_F16, corpus_shift, corpus_resid, c, DT_F32, n, d, k, id_offset, corpus_max_norm, D_out, I_out,
                        status, unproven_count, state, state_bytes, ws, ws_bytes, st);
}

}  // namespace sss

// extern "C" surface of libsss (declared in include/sss.h) + error plumbing.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/sss.h"
#include "sss_common.h"
#include "scan.h"
#include "kargs.h"

namespace sss {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SSS_EHIP;
    }
    return SSS_OK;
}

// implemented in the kernel translation units
size_t ip_topk_workspace_bytes(long nq, long n, int d, int k, int dtype);
size_t ip_topk_state_bytes(long nq);
int ip_topk(const void*, long, const void*, long, int, int, int, long, float, float*, long*, int*, int*, void*, size_t, void*,
            size_t, hipStream_t);
int ip_topk_split(const float*, long, const float*, const void*, long, int, int, long, float, float*, long*, int*, int*, void*,
                  size_t, void*, size_t, hipStream_t);
int split_bf16(const float*, long, int, unsigned short*, hipStream_t);
int ip_topk_f16(const float*, long, const float*, const void*, int, float, long, int, int, long, float, float*, long*, int*, int*,
                void*, size_t, void*, size_t, hipStream_t);
int f16_resid_max(const float*, const unsigned short*, long, int, int, float*, hipStream_t);
size_t ip_topk_scan_workspace_bytes(long, long, int, int, int);
size_t ip_topk_long_workspace_bytes(long, long, int, int);
int ip_topk_long(const void*, long, const void*, int, const void*, int, float, long, int, int, long, float, float*, long*, int*, void*,
                 size_t, hipStream_t);
size_t ip_topk_threshold_workspace_bytes(long, long, int, int);
int ip_topk_threshold(const void*, const int*, long, const void*, int, const void*, int, int, float, long, int, int, long, float,
                      float*, long*, int*, void*, size_t, hipStream_t);
int abs_max(const float*, long, float*, hipStream_t);
int scale_f16(const float*, long, int, unsigned short*, hipStream_t);
int topk_merge(const float*, long, const long*, long, int, long, int, float*, long*, hipStream_t);
int scan_boot_expired(int);
int profile_enable(int);
int profile_read(double*, int*);
size_t ip_topk_exhaustive_workspace_bytes(long nsel, long n);
int ip_topk_exhaustive(const void*, const int*, long, const void*, long, int, int, int, long, int, const float*, float*, long*,
                       void*, size_t, hipStream_t);
int normalize_rows(float*, long, int, long, float, int, hipStream_t);
int row_norm_max(const void*, long, int, int, float*, hipStream_t);
int f32_to_bf16(const float*, long, unsigned short*, hipStream_t);
int gather_rows(const float*, const long*, long, int, float*, long, hipStream_t);
int gather_concat_rows(const float*, const long*, int, const float*, long, int, int, long, float*, long, hipStream_t);
int linear_f32(const float*, long, const float*, long, const float*, float*, long, long, int, int, hipStream_t);
int gat_aggregate(const float*, long, const float*, long, const float*, long, const int*, const int*, long, int,
                  const float*, int, long, float*, long, hipStream_t);
int csr_weighted_sum(const float*, long, const int*, const int*, const float*, long, int, float*, long, hipStream_t);
int gru_combine(const float*, long, const float*, long, const float*, long, int, const float*, long, long, int,
                float*, long, hipStream_t);
int pool_expand(const float*, const float*, long, const int*, const int*, long, long, int, int, const float*, float*,
                long, hipStream_t);
int segment_pool(const float*, long, const int*, const int*, long, long, int, const float*, long, const float*, long,
                 const float*, float*, long, hipStream_t);
int segment_ptr(const long*, long, long, int*, hipStream_t);

int linear_grouped(LinBatch&, hipStream_t);
int layer_update(const LayerArgs&, hipStream_t);
int pool_expand_mean(const float*, const float*, long, const int*, const int*, const int*, const int*, long, long, int, int,
                     const float*, float*, long, float*, long, hipStream_t);
int pool_attention(const float*, long, const float*, long, const float*, long, const float*, const int*, const int*, long, long,
                   int, int, float, int, float*, long, hipStream_t);
int pool_attention_tab(const float*, long, const float*, long, const float*, const float*, const float*, const float*, const int*,
                       const int*, const int*, const int*, long, long, long, int, int, int, float, float*, long, hipStream_t);
int csr_mean(const float*, long, const int*, const int*, long, int, float*, long, hipStream_t);
int segment_reduce(const float*, long, const float*, const int*, long, int, int, float*, long, hipStream_t);
int attention_dot_pool(const float*, long, const int*, long, int, float*, long, hipStream_t);
int item_vote(const float*, const long*, long, int, const long*, const int*, long, long, int, long*, double*, int*, hipStream_t);
size_t hamming_workspace_bytes(long nq, long n);
int hamming_capacity(long nq, long n);
int hamming_topk(const unsigned char*, long, const unsigned char*, long, int, int, long, int*, long*, int*, void*, size_t, hipStream_t);
size_t hamming_exhaustive_workspace_bytes(long nsel, long n);
int hamming_topk_exhaustive(const unsigned char*, const int*, long, const unsigned char*, long, int, int, long, int*, long*, void*,
                            size_t, hipStream_t);
int pack_sign_bits(const float*, long, int, long, unsigned char*, int, hipStream_t);
size_t graph_scratch_ints(long S);
int graph_counts(const long*, const unsigned char*, const long*, long, int*, int*, int*, hipStream_t);
int graph_fill(const long*, const unsigned char*, const long*, const long*, long, const int*, const GraphOut&, hipStream_t);

}  // namespace sss

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int sss_version(void) { return 240; }
const char* sss_last_error(void) { return sss::g_err; }

int sss_normalize_rows(float* x, int64_t n, int d, int64_t ld, float eps, int rule, void* stream) {
    return sss::normalize_rows(x, n, d, ld, eps, rule, ST(stream));
}
int sss_row_norm_max(const void* x, int64_t n, int d, int dtype, float* out, void* stream) {
    return sss::row_norm_max(x, n, d, dtype, out, ST(stream));
}
int sss_f32_to_bf16(const float* x, int64_t count, uint16_t* y, void* stream) {
    return sss::f32_to_bf16(x, count, y, ST(stream));
}
size_t sss_ip_topk_state_bytes(int64_t nq) { return sss::ip_topk_state_bytes(nq); }
size_t sss_ip_topk_workspace_bytes(int64_t nq, int64_t n, int d, int k, int dtype) {
    return sss::ip_topk_workspace_bytes(nq, n, d, k, dtype);
}
int sss_ip_topk(const void* q, int64_t nq, const void* corpus, int64_t n, int d, int k, int dtype, int64_t id_offset,
                float corpus_max_norm, float* D_out, int64_t* I_out, int32_t* status, int32_t* unproven_count,
                void* state, size_t state_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    return sss::ip_topk(q, nq, corpus, n, d, k, dtype, id_offset, corpus_max_norm, D_out,
                        reinterpret_cast<long*>(I_out), status, unproven_count, state, state_bytes, workspace,
                        workspace_bytes, ST(stream));
}
int sss_split_bf16(const float* x, int64_t n, int d, uint16_t* y, void* stream) {
    return sss::split_bf16(x, n, d, y, ST(stream));
}
int sss_ip_topk_split(const float* q, int64_t nq, const float* corpus, const uint16_t* corpus_split, int64_t n, int d,
                      int k, int64_t id_offset, float corpus_max_norm, float* D_out, int64_t* I_out, int32_t* status,
                      int32_t* unproven_count, void* state, size_t state_bytes, void* workspace,
                      size_t workspace_bytes, void* stream) {
    return sss::ip_topk_split(q, nq, corpus, corpus_split, n, d, k, id_offset, corpus_max_norm, D_out,
                              reinterpret_cast<long*>(I_out), status, unproven_count, state, state_bytes, workspace,
                              workspace_bytes, ST(stream));
}
int sss_abs_max(const float* x, int64_t count, float* out, void* stream) { return sss::abs_max(x, count, out, ST(stream)); }
int sss_f16_shift(float amax) { return sss::f16_shift(amax); }
int sss_scale_f16(const float* x, int64_t count, int shift, uint16_t* y, void* stream) {
    return sss::scale_f16(x, count, shift, y, ST(stream));
}
size_t sss_ip_topk_f16_workspace_bytes(int64_t nq, int64_t n, int d, int k) {
    return sss::ip_topk_scan_workspace_bytes(nq, n, d, k, sss::DT_F16);
}
int sss_f16_resid_max(const float* x, const uint16_t* y, int64_t n, int d, int shift, float* out, void* stream) {
    return sss::f16_resid_max(x, y, n, d, shift, out, ST(stream));
}
int sss_ip_topk_f16(const float* q, int64_t nq, const float* corpus, const uint16_t* corpus_f16, int corpus_shift,
                    float corpus_resid_norm, int64_t n, int d, int k, int64_t id_offset, float corpus_max_norm, float* D_out, int64_t* I_out,
                    int32_t* status, int32_t* unproven_count, void* state, size_t state_bytes, void* workspace,
                    size_t workspace_bytes, void* stream) {
    return sss::ip_topk_f16(q, nq, corpus, corpus_f16, corpus_shift, corpus_resid_norm, n, d, k, id_offset, corpus_max_norm, D_out,
                            reinterpret_cast<long*>(I_out), status, unproven_count, state, state_bytes, workspace,
                            workspace_bytes, ST(stream));
}
size_t sss_ip_topk_long_workspace_bytes(int64_t nq, int64_t n, int d, int dtype) {
    return sss::ip_topk_long_workspace_bytes(nq, n, d, dtype);
}
int sss_ip_topk_long(const void* q, int64_t nq, const void* corpus, int dtype, const void* scan_image, int corpus_shift,
                     float corpus_resid_norm, int64_t n, int d, int k, int64_t id_offset, float corpus_max_norm, float* D_out,
                     int64_t* I_out, int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
    if (dtype != 0 && dtype != 1) { sss::set_error("ip_topk_long: dtype must be 0 (f32) or 1 (bf16)"); return SSS_EINVAL; }
    return sss::ip_topk_long(q, nq, corpus, dtype, scan_image, corpus_shift, corpus_resid_norm, n, d, k, id_offset,
                             corpus_max_norm, D_out, reinterpret_cast<long*>(I_out), status, workspace, workspace_bytes,
                             ST(stream));
}
size_t sss_ip_topk_threshold_workspace_bytes(int64_t nsel, int64_t n, int d, int scan) {
    return sss::ip_topk_threshold_workspace_bytes(nsel, n, d, scan);
}
int sss_ip_topk_threshold(const void* q, const int32_t* qsel, int64_t nsel, const void* corpus, int dtype, const void* scan_image,
                          int scan, int corpus_shift, float corpus_resid_norm, int64_t n, int d, int k, int64_t id_offset,
                          float corpus_max_norm, float* D_out, int64_t* I_out, int32_t* status, void* workspace,
                          size_t workspace_bytes, void* stream) {
    return sss::ip_topk_threshold(q, qsel, nsel, corpus, dtype, scan_image, scan, corpus_shift, corpus_resid_norm, n, d, k,
                                  id_offset, corpus_max_norm, D_out, reinterpret_cast<long*>(I_out), status, workspace,
                                  workspace_bytes, ST(stream));
}
size_t sss_ip_topk_exhaustive_workspace_bytes(int64_t nsel, int64_t n) {
    return sss::ip_topk_exhaustive_workspace_bytes(nsel, n);
}
int sss_ip_topk_exhaustive(const void* q, const int32_t* qsel, int64_t nsel, const void* corpus, int64_t n,
                           int d, int k, int dtype, int64_t id_offset, int metric, float* D_out, int64_t* I_out,
                           void* workspace, size_t workspace_bytes, void* stream) {
    return sss::ip_topk_exhaustive(q, qsel, nsel, corpus, n, d, k, dtype, id_offset, metric, nullptr, D_out,
                                   reinterpret_cast<long*>(I_out), workspace, workspace_bytes, ST(stream));
}
int sss_ip_topk_exhaustive_lb(const void* q, const int32_t* qsel, int64_t nsel, const void* corpus, int64_t n,
                              int d, int k, int dtype, int64_t id_offset, const float* lower_bound, float* D_out,
                              int64_t* I_out, void* workspace, size_t workspace_bytes, void* stream) {
    return sss::ip_topk_exhaustive(q, qsel, nsel, corpus, n, d, k, dtype, id_offset, 0, lower_bound, D_out,
                                   reinterpret_cast<long*>(I_out), workspace, workspace_bytes, ST(stream));
}
int sss_topk_merge(const float* D_in, int64_t d_shard_stride, const int64_t* I_in, int64_t i_shard_stride,
                   int shards, int64_t nq, int k, float* D_out, int64_t* I_out, void* stream) {
    return sss::topk_merge(D_in, d_shard_stride, reinterpret_cast<const long*>(I_in), i_shard_stride, shards, nq,
                           k, D_out, reinterpret_cast<long*>(I_out), ST(stream));
}
int sss_scan_boot_expired(int reset) { return sss::scan_boot_expired(reset); }
int sss_profile_enable(int on) { return sss::profile_enable(on); }
int sss_profile_read(double* total_ms, int* launches) { return sss::profile_read(total_ms, launches); }
int sss_gather_rows(const float* table, const int64_t* ids, int64_t n, int d, float* out, int64_t ld_out,
                    void* stream) {
    return sss::gather_rows(table, reinterpret_cast<const long*>(ids), n, d, out, ld_out, ST(stream));
}

int sss_gather_concat_rows(const float* table, const int64_t* ids, int d_id, const float* feat, int64_t ld_feat, int d_feat,
                           int d_pad, int64_t n, float* out, int64_t ld_out, void* stream) {
    return sss::gather_concat_rows(table, reinterpret_cast<const long*>(ids), d_id, feat, ld_feat, d_feat, d_pad, n, out, ld_out,
                                   ST(stream));
}

int sss_linear(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy,
               int64_t n, int m, int k, void* stream) {
    return sss::linear_f32(x, ldx, w, ldw, bias, y, ldy, n, m, k, ST(stream));
}
int sss_gat_aggregate(const float* xs, int64_t ld_xs, const float* a_src, int64_t ld_as, const float* a_dst,
                      int64_t ld_ad, const int32_t* rowptr, const int32_t* col, int64_t n_dst, int h,
                      const float* bias, int relu, int64_t n_self_loop, float* out, int64_t ld_out, void* stream) {
    return sss::gat_aggregate(xs, ld_xs, a_src, ld_as, a_dst, ld_ad, rowptr, col, n_dst, h, bias, relu, n_self_loop, out,
                              ld_out, ST(stream));
}
int sss_csr_weighted_sum(const float* m, int64_t ld_m, const int32_t* rowptr, const int32_t* col, const float* w,
                         int64_t n_dst, int h, float* out, int64_t ld_out, void* stream) {
    return sss::csr_weighted_sum(m, ld_m, rowptr, col, w, n_dst, h, out, ld_out, ST(stream));
}
int sss_gru_combine(const float* gi, int64_t ld_gi, const float* gh, int64_t ld_gh, const float* x, int64_t ld_x,
                    int d_x, const float* add, int64_t ld_add, int64_t n, int h, float* out, int64_t ld_out,
                    void* stream) {
    return sss::gru_combine(gi, ld_gi, gh, ld_gh, x, ld_x, d_x, add, ld_add, n, h, out, ld_out, ST(stream));
}
int sss_pool_expand(const float* lin_p, const float* lin_q, int64_t ld_lin, const int32_t* src_row,
                    const int32_t* pos_id, int64_t n_clicks, int64_t n_exp, int d_lin, int p, const float* pos_emb,
                    float* node, int64_t ld_node, void* stream) {
    return sss::pool_expand(lin_p, lin_q, ld_lin, src_row, pos_id, n_clicks, n_exp, d_lin, p, pos_emb, node, ld_node,
                            ST(stream));
}
int sss_segment_pool(const float* node, int64_t ld_node, const int32_t* pptr, const int32_t* qptr, int64_t n_clicks,
                     int64_t n_graphs, int d, const float* a, int64_t ld_a, const float* bcoarse, int64_t ld_b,
                     const float* watt, float* out, int64_t ld_out, void* stream) {
    return sss::segment_pool(node, ld_node, pptr, qptr, n_clicks, n_graphs, d, a, ld_a, bcoarse, ld_b, watt, out,
                             ld_out, ST(stream));
}
int sss_linear_grouped(const sss_linear_problem* problems, int n_problems, int k, void* stream) {
    if (!problems || n_problems < 1 || n_problems > 4) { sss::set_error("linear_grouped: 1..4 problems"); return SSS_EINVAL; }
    sss::LinBatch b;
    b.nprob = n_problems; b.K = k;
    for (int i = 0; i < n_problems; ++i) {
        const sss_linear_problem& s = problems[i];
        sss::LinProb& p = b.p[i];
        p.x = s.x; p.ldx = s.ldx; p.ids = reinterpret_cast<const long*>(s.ids); p.table = s.table; p.xcopy = s.xcopy;
        p.ld_xcopy = s.ld_xcopy; p.w = s.w; p.ldw = s.ldw; p.bias = s.bias; p.y = s.y; p.ldy = s.ldy; p.n = s.n; p.m = s.m;
        p.act = s.act; p.post_scale = s.post_scale; p.post_shift = s.post_shift; p.tiles_m = 0; p.tile_begin = 0;
        if (s.act < 0 || s.act > 4 || (s.post_scale == nullptr) != (s.post_shift == nullptr)) {
            sss::set_error("linear_grouped: problem %d: act must be 0..4, post_scale / post_shift come together", i);
            return SSS_EINVAL;
        }
    }
    return sss::linear_grouped(b, ST(stream));
}
int sss_hetero_layer_update(const sss_layer_args* a, void* stream) {
    if (!a) { sss::set_error("hetero_layer_update: null args"); return SSS_EINVAL; }
    sss::LayerArgs l;
    l.Yp = a->yp; l.ldyp = a->ld_yp; l.Yq = a->yq; l.ldyq = a->ld_yq; l.h = a->h; l.d_x = a->d_x;
    l.rowptr_qp = a->rowptr_qp; l.col_qp = a->col_qp; l.rowptr_pp = a->rowptr_pp; l.col_pp = a->col_pp; l.w_pp = a->w_pp;
    l.bias_qp = a->bias_qp; l.b_ih = a->b_ih; l.xin_p = a->xin_p; l.ld_xin = a->ld_xin; l.out_p = a->out_p;
    l.ld_outp = a->ld_out_p; l.Np = a->np; l.rowptr_pq = a->rowptr_pq; l.col_pq = a->col_pq; l.bias_pq = a->bias_pq;
    l.out_q = a->out_q; l.ld_outq = a->ld_out_q; l.Nq = a->nq; l.n_self_loop = a->n_self_loop;
    l.row_p = reinterpret_cast<const long*>(a->row_p); l.row_q = reinterpret_cast<const long*>(a->row_q);
    l.x0_p = a->x0_p; l.ld_x0p = a->ld_x0_p; l.xq_table = a->xq_table; l.ld_xq = a->ld_xq; l.x0_q = a->x0_q; l.ld_x0q = a->ld_x0_q;
    return sss::layer_update(l, ST(stream));
}
int sss_pool_expand_mean(const float* lin_p, const float* lin_q, int64_t ld_lin, const int32_t* src_row,
                         const int32_t* pos_id, const int32_t* pptr, const int32_t* qptr, int64_t n_clicks,
                         int64_t n_graphs, int d_lin, int p, const float* pos_emb, float* node, int64_t ld_node,
                         float* coarse, int64_t ld_coarse, void* stream) {
    return sss::pool_expand_mean(lin_p, lin_q, ld_lin, src_row, pos_id, pptr, qptr, n_clicks, n_graphs, d_lin, p, pos_emb,
                                 node, ld_node, coarse, ld_coarse, ST(stream));
}
int sss_pool_attention(const float* node, int64_t ld_node, const float* a, int64_t ld_a, const float* b, int64_t ld_b,
                       const float* watt, const int32_t* pptr, const int32_t* qptr, int64_t n_clicks, int64_t n_graphs,
                       int d, int normalize, float eps, int reduce_sum, float* out, int64_t ld_out, void* stream) {
    return sss::pool_attention(node, ld_node, a, ld_a, b, ld_b, watt, pptr, qptr, n_clicks, n_graphs, d, normalize, eps,
                               reduce_sum, out, ld_out, ST(stream));
}
int sss_pool_attention_tab(const float* t, int64_t ld_t, const float* ac, int64_t ld_ac, const float* tanhpos, const float* a2tab,
                           const float* c2tab, const float* watt, const int32_t* src_row, const int32_t* pos_id,
                           const int32_t* pptr, const int32_t* qptr, int64_t n_clicks, int64_t np, int64_t n_graphs, int d_lin,
                           int p, int normalize, float eps, float* out, int64_t ld_out, void* stream) {
    return sss::pool_attention_tab(t, ld_t, ac, ld_ac, tanhpos, a2tab, c2tab, watt, src_row, pos_id, pptr, qptr, n_clicks, np,
                                   n_graphs, d_lin, p, normalize, eps, out, ld_out, ST(stream));
}
int sss_csr_mean(const float* x, int64_t ld_x, const int32_t* rowptr, const int32_t* col, int64_t n_dst, int d, float* out,
                 int64_t ld_out, void* stream) {
    return sss::csr_mean(x, ld_x, rowptr, col, n_dst, d, out, ld_out, ST(stream));
}
int sss_segment_reduce(const float* x, int64_t ld_x, const float* w, const int32_t* ptr, int64_t n_graphs, int d, int mode,
                       float* out, int64_t ld_out, void* stream) {
    return sss::segment_reduce(x, ld_x, w, ptr, n_graphs, d, mode, out, ld_out, ST(stream));
}
int sss_attention_dot_pool(const float* x, int64_t ld_x, const int32_t* ptr, int64_t n_graphs, int d, float* out,
                           int64_t ld_out, void* stream) {
    return sss::attention_dot_pool(x, ld_x, ptr, n_graphs, d, out, ld_out, ST(stream));
}
int sss_pack_sign_bits(const float* x, int64_t n, int c, int64_t ldx, uint8_t* out, int nbytes, void* stream) {
    return sss::pack_sign_bits(x, n, c, ldx, out, nbytes, ST(stream));
}
size_t sss_hamming_topk_workspace_bytes(int64_t nq, int64_t n) { return sss::hamming_workspace_bytes(nq, n); }
int sss_hamming_topk_capacity(int64_t nq, int64_t n) { return sss::hamming_capacity(nq, n); }
int sss_hamming_topk(const uint8_t* q, int64_t nq, const uint8_t* codes, int64_t n, int nbytes, int k, int64_t id_offset,
                     int32_t* D_out, int64_t* I_out, int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
    return sss::hamming_topk(q, nq, codes, n, nbytes, k, id_offset, D_out, reinterpret_cast<long*>(I_out), status, workspace,
                             workspace_bytes, ST(stream));
}
size_t sss_hamming_topk_exhaustive_workspace_bytes(int64_t nsel, int64_t n) {
    return sss::hamming_exhaustive_workspace_bytes(nsel, n);
}
int sss_hamming_topk_exhaustive(const uint8_t* q, const int32_t* qsel, int64_t nsel, const uint8_t* codes, int64_t n,
                                int nbytes, int k, int64_t id_offset, int32_t* D_out, int64_t* I_out, void* workspace,
                                size_t workspace_bytes, void* stream) {
    return sss::hamming_topk_exhaustive(q, qsel, nsel, codes, n, nbytes, k, id_offset, D_out, reinterpret_cast<long*>(I_out),
                                        workspace, workspace_bytes, ST(stream));
}
size_t sss_graph_scratch_ints(int64_t n_sessions) { return sss::graph_scratch_ints(n_sessions); }
int sss_graph_counts(const int64_t* sess_ptr, const uint8_t* is_search, const int64_t* item_id, int64_t n_sessions,
                     int32_t* bases, int32_t* scratch, int32_t* err, void* stream) {
    return sss::graph_counts(reinterpret_cast<const long*>(sess_ptr), is_search, reinterpret_cast<const long*>(item_id),
                             n_sessions, bases, scratch, err, ST(stream));
}
int sss_graph_fill(const int64_t* sess_ptr, const uint8_t* is_search, const int64_t* item_id, const int64_t* query_tok,
                   int64_t n_sessions, const int32_t* bases, const sss_graph_out* o, void* stream) {
    if (!o) { sss::set_error("graph_fill: null outputs"); return SSS_EINVAL; }
    sss::GraphOut g;
    g.q_x = reinterpret_cast<long*>(o->q_x); g.q_batch = reinterpret_cast<long*>(o->q_batch); g.q_pos = o->q_pos;
    g.p_x = reinterpret_cast<long*>(o->p_x); g.p_batch = reinterpret_cast<long*>(o->p_batch);
    g.p_cnt = reinterpret_cast<long*>(o->p_cnt);
    g.rowptr_qp = o->rowptr_qp; g.col_qp = o->col_qp; g.rowptr_pq = o->rowptr_pq; g.col_pq = o->col_pq;
    g.rowptr_pp = o->rowptr_pp; g.col_pp = o->col_pp; g.w_pp = o->w_pp; g.src_row = o->src_row; g.pos_id = o->pos_id;
    return sss::graph_fill(reinterpret_cast<const long*>(sess_ptr), is_search, reinterpret_cast<const long*>(item_id),
                           reinterpret_cast<const long*>(query_tok), n_sessions, bases, g, ST(stream));
}
int sss_knn_item_vote(const float* D, const int64_t* I, int64_t nq, int s, const int64_t* items_ptr, const int32_t* items,
                      int64_t id_offset, int64_t n_sessions, int k, int64_t* out_items, double* out_weights,
                      int32_t* status, void* stream) {
    return sss::item_vote(D, reinterpret_cast<const long*>(I), nq, s, reinterpret_cast<const long*>(items_ptr), items,
                          id_offset, n_sessions, k, reinterpret_cast<long*>(out_items), out_weights, status, ST(stream));
}
int sss_segment_ptr(const int64_t* batch, int64_t n, int64_t n_graphs, int32_t* ptr, void* stream) {
    return sss::segment_ptr(reinterpret_cast<const long*>(batch), n, n_graphs, ptr, ST(stream));
}

}  // extern "C"

// extern "C" surface of libsss (declared in include/sss.h) + error plumbing.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/sss.h"
#include "sss_common.h"

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
int ip_topk(const void*, long, const void*, long, int, int, int, long, float, float*, long*, int*, void*,
            size_t, hipStream_t);
int topk_merge(const float*, long, const long*, long, int, long, int, float*, long*, hipStream_t);
int profile_enable(int);
int profile_read(double*, int*);
size_t ip_topk_exhaustive_workspace_bytes(long nsel, long n);
int ip_topk_exhaustive(const void*, const int*, long, const void*, long, int, int, int, long, int, float*, long*,
                       void*, size_t, hipStream_t);
int normalize_rows(float*, long, int, long, float, int, hipStream_t);
int row_norm_max(const void*, long, int, int, float*, hipStream_t);
int f32_to_bf16(const float*, long, unsigned short*, hipStream_t);
int gather_rows(const float*, const long*, long, int, float*, long, hipStream_t);
int linear_f32(const float*, long, const float*, long, const float*, float*, long, long, int, int, hipStream_t);
int gat_aggregate(const float*, long, const float*, long, const float*, long, const int*, const int*, long, int,
                  const float*, int, float*, long, hipStream_t);
int csr_weighted_sum(const float*, long, const int*, const int*, const float*, long, int, float*, long, hipStream_t);
int gru_combine(const float*, long, const float*, long, const float*, long, int, const float*, long, long, int,
                float*, long, hipStream_t);
int pool_expand(const float*, const float*, long, const int*, const int*, long, long, int, int, const float*, float*,
                long, hipStream_t);
int segment_pool(const float*, long, const int*, const int*, long, long, int, const float*, long, const float*, long,
                 const float*, float*, long, hipStream_t);
int segment_ptr(const long*, long, long, int*, hipStream_t);

}  // namespace sss

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int sss_version(void) { return 200; }
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
size_t sss_ip_topk_workspace_bytes(int64_t nq, int64_t n, int d, int k, int dtype) {
    return sss::ip_topk_workspace_bytes(nq, n, d, k, dtype);
}
int sss_ip_topk(const void* q, int64_t nq, const void* corpus, int64_t n, int d, int k, int dtype, int64_t id_offset,
                float corpus_max_norm, float* D_out, int64_t* I_out, int32_t* status, void* workspace,
                size_t workspace_bytes, void* stream) {
    return sss::ip_topk(q, nq, corpus, n, d, k, dtype, id_offset, corpus_max_norm, D_out,
                        reinterpret_cast<long*>(I_out), status, workspace, workspace_bytes, ST(stream));
}
size_t sss_ip_topk_exhaustive_workspace_bytes(int64_t nsel, int64_t n) {
    return sss::ip_topk_exhaustive_workspace_bytes(nsel, n);
}
int sss_ip_topk_exhaustive(const void* q, const int32_t* qsel, int64_t nsel, const void* corpus, int64_t n,
                           int d, int k, int dtype, int64_t id_offset, int metric, float* D_out, int64_t* I_out,
                           void* workspace, size_t workspace_bytes, void* stream) {
    return sss::ip_topk_exhaustive(q, qsel, nsel, corpus, n, d, k, dtype, id_offset, metric, D_out,
                                   reinterpret_cast<long*>(I_out), workspace, workspace_bytes, ST(stream));
}
int sss_topk_merge(const float* D_in, int64_t d_shard_stride, const int64_t* I_in, int64_t i_shard_stride,
                   int shards, int64_t nq, int k, float* D_out, int64_t* I_out, void* stream) {
    return sss::topk_merge(D_in, d_shard_stride, reinterpret_cast<const long*>(I_in), i_shard_stride, shards, nq,
                           k, D_out, reinterpret_cast<long*>(I_out), ST(stream));
}
int sss_profile_enable(int on) { return sss::profile_enable(on); }
int sss_profile_read(double* total_ms, int* launches) { return sss::profile_read(total_ms, launches); }
int sss_gather_rows(const float* table, const int64_t* ids, int64_t n, int d, float* out, int64_t ld_out,
                    void* stream) {
    return sss::gather_rows(table, reinterpret_cast<const long*>(ids), n, d, out, ld_out, ST(stream));
}

int sss_linear(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy,
               int64_t n, int m, int k, void* stream) {
    return sss::linear_f32(x, ldx, w, ldw, bias, y, ldy, n, m, k, ST(stream));
}
int sss_gat_aggregate(const float* xs, int64_t ld_xs, const float* a_src, int64_t ld_as, const float* a_dst,
                      int64_t ld_ad, const int32_t* rowptr, const int32_t* col, int64_t n_dst, int h,
                      const float* bias, int relu, float* out, int64_t ld_out, void* stream) {
    return sss::gat_aggregate(xs, ld_xs, a_src, ld_as, a_dst, ld_ad, rowptr, col, n_dst, h, bias, relu, out, ld_out,
                              ST(stream));
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
int sss_segment_ptr(const int64_t* batch, int64_t n, int64_t n_graphs, int32_t* ptr, void* stream) {
    return sss::segment_ptr(reinterpret_cast<const long*>(batch), n, n_graphs, ptr, ST(stream));
}

}  // extern "C"

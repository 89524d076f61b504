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
size_t ip_topk_workspace_bytes(long nq, long n, int d, int k);
int ip_topk_f32(const float*, long, const float*, long, int, int, long, float, float*, long*, int*, void*,
                size_t, hipStream_t);
int topk_merge(const float*, const long*, int, long, int, float*, long*, hipStream_t);
size_t ip_topk_exhaustive_workspace_bytes(long nsel, long n);
int ip_topk_exhaustive(const float*, const int*, long, const float*, long, int, int, long, int, float*, long*,
                       void*, size_t, hipStream_t);
int normalize_rows(float*, long, int, long, float, int, hipStream_t);
int row_norm_max(const float*, long, int, float*, hipStream_t);
int gather_rows(const float*, const long*, long, int, float*, long, hipStream_t);

}  // namespace sss

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int sss_version(void) { return 100; }
const char* sss_last_error(void) { return sss::g_err; }

int sss_normalize_rows(float* x, int64_t n, int d, int64_t ld, float eps, int rule, void* stream) {
    return sss::normalize_rows(x, n, d, ld, eps, rule, ST(stream));
}
int sss_row_norm_max(const float* x, int64_t n, int d, float* out, void* stream) {
    return sss::row_norm_max(x, n, d, out, ST(stream));
}
size_t sss_ip_topk_workspace_bytes(int64_t nq, int64_t n, int d, int k) {
    return sss::ip_topk_workspace_bytes(nq, n, d, k);
}
int sss_ip_topk(const float* q, int64_t nq, const float* corpus, int64_t n, int d, int k, int64_t id_offset,
                float corpus_max_norm, float* D_out, int64_t* I_out, int32_t* status, void* workspace,
                size_t workspace_bytes, void* stream) {
    return sss::ip_topk_f32(q, nq, corpus, n, d, k, id_offset, corpus_max_norm, D_out,
                            reinterpret_cast<long*>(I_out), status, workspace, workspace_bytes, ST(stream));
}
size_t sss_ip_topk_exhaustive_workspace_bytes(int64_t nsel, int64_t n) {
    return sss::ip_topk_exhaustive_workspace_bytes(nsel, n);
}
int sss_ip_topk_exhaustive(const float* q, const int32_t* qsel, int64_t nsel, const float* corpus, int64_t n,
                           int d, int k, int64_t id_offset, int metric, float* D_out, int64_t* I_out,
                           void* workspace, size_t workspace_bytes, void* stream) {
    return sss::ip_topk_exhaustive(q, qsel, nsel, corpus, n, d, k, id_offset, metric, D_out,
                                   reinterpret_cast<long*>(I_out), workspace, workspace_bytes, ST(stream));
}
int sss_topk_merge(const float* D_in, const int64_t* I_in, int shards, int64_t nq, int k, float* D_out,
                   int64_t* I_out, void* stream) {
    return sss::topk_merge(D_in, reinterpret_cast<const long*>(I_in), shards, nq, k, D_out,
                           reinterpret_cast<long*>(I_out), ST(stream));
}
int sss_gather_rows(const float* table, const int64_t* ids, int64_t n, int d, float* out, int64_t ld_out,
                    void* stream) {
    return sss::gather_rows(table, reinterpret_cast<const long*>(ids), n, d, out, ld_out, ST(stream));
}

}  // extern "C"

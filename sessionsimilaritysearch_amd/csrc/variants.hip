// Other conv / pool variants of the reference on the same CSR + segment kernels (gfx950;
// SURVEY.md section 8(f) row 4).  All HBM-bound row kernels: one LPR-lane group per target node or
// per graph, 16 bytes per lane per access; rows wider than 64 lanes x 4 floats (the reference runs these at
// gnn_nout = 800, config.py:15-16) are walked in column chunks of LPR float4s per lane.
//   k_csr_mean          <- the mean aggregation inside PyG SAGEConv (model/gnn.py:89-121:
//                          three SAGEConv((-1,-1), h) layers made heterogeneous by to_hetero)
//   k_segment_reduce    <- global_mean_pool / global_add_pool / global_max_pool of GraphPooling
//                          (model/gnn.py:123-143), with an optional per-row weight (the
//                          last_click_mask product of SRGNN_Pooling, model/gnn.py:172)
//   k_attention_dot_pool<- AttentionPooling (model/gnn.py:145-161): att_i = <x_i, mean_g>, as a
//                          row-wise dot instead of the reference's dense [n_nodes, n_graphs] matrix
#include "sss_common.h"

namespace sss {

template <int LPR>
__global__ __launch_bounds__(256) void k_csr_mean(const float* __restrict__ x, long ld_x, const int* __restrict__ rowptr,
                                                  const int* __restrict__ col, long n_dst, int d,
                                                  float* __restrict__ out, long ld_out) {
    const int sub = threadIdx.x % LPR;
    const long i = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (i >= n_dst) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    const float inv = e1 > e0 ? 1.f / (float)(e1 - e0) : 0.f;      // no neighbours: zeros (scatter-mean semantics)
    for (int c4 = sub * 4; c4 < d; c4 += LPR * 4) {                // one pass for d <= 256; column chunks beyond
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = e0; e < e1; ++e) {
            const float4 v = *reinterpret_cast<const float4*>(x + (long)col[e] * ld_x + c4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(out + i * ld_out + c4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}

// mode 0 mean, 1 add, 2 max over rows [ptr[g], ptr[g+1]); w (may be null) multiplies row r first.
template <int LPR>
__global__ __launch_bounds__(256) void k_segment_reduce(const float* __restrict__ x, long ld_x, const float* __restrict__ w,
                                                        const int* __restrict__ ptr, long n_graphs, int d, int mode,
                                                        float* __restrict__ out, long ld_out) {
    const int sub = threadIdx.x % LPR;
    const long g = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (g >= n_graphs) return;
    const int r0 = ptr[g], r1 = ptr[g + 1];
    for (int c4 = sub * 4; c4 < d; c4 += LPR * 4) {                // one pass for d <= 256; column chunks beyond
        float4 acc = mode == 2 ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = r0; r < r1; ++r) {
            float4 v = *reinterpret_cast<const float4*>(x + (long)r * ld_x + c4);
            if (w) { const float s = w[r]; v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
            if (mode == 2) { acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y); acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w); }
            else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
        if (r1 <= r0) acc = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (mode == 0) { const float inv = 1.f / (float)(r1 - r0); acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv; }
        *reinterpret_cast<float4*>(out + g * ld_out + c4) = acc;
    }
}

// out[g] = mean_i(x_i * <x_i, mean_g>) over the rows of graph g.  A lane owns up to NCH float4 columns (c4 = 4 (sub +
// j LPR)): NCH = 1 for rows of at most 256 floats, 8 for wider ones (up to 2048: the reference's 800 and 1600).
template <int LPR, int NCH>
__global__ __launch_bounds__(256) void k_attention_dot_pool(const float* __restrict__ x, long ld_x, const int* __restrict__ ptr,
                                                            long n_graphs, int d, float* __restrict__ out, long ld_out) {
    const int sub = threadIdx.x % LPR;
    const long g = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    const bool in_g = g < n_graphs;
    int r0 = 0, r1 = 0;
    if (in_g) { r0 = ptr[g]; r1 = ptr[g + 1]; }
    const int cnt = r1 - r0;
    float4 m[NCH], acc[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) { m[j] = make_float4(0.f, 0.f, 0.f, 0.f); acc[j] = m[j]; }
    for (int r = r0; r < r1; ++r) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c4 = (sub + j * LPR) * 4;
            if (c4 < d) {
                const float4 v = *reinterpret_cast<const float4*>(x + (long)r * ld_x + c4);
                m[j].x += v.x; m[j].y += v.y; m[j].z += v.z; m[j].w += v.w;
            }
        }
    }
    const float inv = cnt > 0 ? 1.f / (float)cnt : 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) { m[j].x *= inv; m[j].y *= inv; m[j].z *= inv; m[j].w *= inv; }
    int cmax = cnt;                                     // groups of one wave may differ: keep the shuffles convergent
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
    for (int t = 0; t < cmax; ++t) {
        float4 v[NCH];
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c4 = (sub + j * LPR) * 4;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in_g && t < cnt && c4 < d) v[j] = *reinterpret_cast<const float4*>(x + (long)(r0 + t) * ld_x + c4);
            part += v[j].x * m[j].x + v[j].y * m[j].y + v[j].z * m[j].z + v[j].w * m[j].w;
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) part += __shfl_xor(part, o);
#pragma unroll
        for (int j = 0; j < NCH; ++j) { acc[j].x += part * v[j].x; acc[j].y += part * v[j].y; acc[j].z += part * v[j].z; acc[j].w += part * v[j].w; }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c4 = (sub + j * LPR) * 4;
        if (in_g && c4 < d)
            *reinterpret_cast<float4*>(out + g * ld_out + c4) = make_float4(acc[j].x * inv, acc[j].y * inv, acc[j].z * inv, acc[j].w * inv);
    }
}

static int lanes4(int d) {
    const int nv = d / 4;
    int l = 1;
    while (l < nv && l < 64) l <<= 1;
    return l;
}
#define SSS_LPRV(lpr, CALL)                             \
    switch (lpr) {                                      \
        case 1: { constexpr int L = 1; CALL; } break;   \
        case 2: { constexpr int L = 2; CALL; } break;   \
        case 4: { constexpr int L = 4; CALL; } break;   \
        case 8: { constexpr int L = 8; CALL; } break;   \
        case 16: { constexpr int L = 16; CALL; } break; \
        case 32: { constexpr int L = 32; CALL; } break; \
        default: { constexpr int L = 64; CALL; } break; \
    }
constexpr int MAX_WIDTH = 2048;     // 8 column chunks of 64 lanes x 4 floats (k_attention_dot_pool keeps them in registers)
static bool ok_rows(int d, long a, long b) { return d > 0 && d % 4 == 0 && d <= MAX_WIDTH && a % 4 == 0 && b % 4 == 0 && a >= d && b >= d; }

int csr_mean(const float* x, long ld_x, const int* rowptr, const int* col, long n_dst, int d, float* out, long ld_out, hipStream_t st) {
    if (n_dst < 0 || !ok_rows(d, ld_x, ld_out)) { set_error("csr_mean: need d %% 4 == 0, d <= 2048, 16-byte aligned row strides"); return SSS_EINVAL; }
    if (n_dst == 0) return SSS_OK;
    const int lpr = lanes4(d);
    const long per = 256 / lpr;
    SSS_LPRV(lpr, hipLaunchKernelGGL(k_csr_mean<L>, dim3((unsigned)((n_dst + per - 1) / per)), dim3(256), 0, st, x, ld_x, rowptr, col, n_dst, d, out, ld_out));
    return check_launch("k_csr_mean");
}
int segment_reduce(const float* x, long ld_x, const float* w, const int* ptr, long n_graphs, int d, int mode, float* out, long ld_out,
                   hipStream_t st) {
    if (n_graphs < 0 || mode < 0 || mode > 2 || !ok_rows(d, ld_x, ld_out)) { set_error("segment_reduce: need mode in {0,1,2}, d %% 4 == 0, d <= 2048"); return SSS_EINVAL; }
    if (n_graphs == 0) return SSS_OK;
    const int lpr = lanes4(d);
    const long per = 256 / lpr;
    SSS_LPRV(lpr, hipLaunchKernelGGL(k_segment_reduce<L>, dim3((unsigned)((n_graphs + per - 1) / per)), dim3(256), 0, st, x, ld_x, w, ptr, n_graphs, d, mode, out, ld_out));
    return check_launch("k_segment_reduce");
}
int attention_dot_pool(const float* x, long ld_x, const int* ptr, long n_graphs, int d, float* out, long ld_out, hipStream_t st) {
    if (n_graphs < 0 || !ok_rows(d, ld_x, ld_out)) { set_error("attention_dot_pool: need d %% 4 == 0, d <= 2048"); return SSS_EINVAL; }
    if (n_graphs == 0) return SSS_OK;
    const int lpr = lanes4(d);
    const long per = 256 / lpr;
    if (d <= 256) {
        SSS_LPRV(lpr, hipLaunchKernelGGL((k_attention_dot_pool<L, 1>), dim3((unsigned)((n_graphs + per - 1) / per)), dim3(256), 0, st, x, ld_x, ptr, n_graphs, d, out, ld_out));
    } else {
        hipLaunchKernelGGL((k_attention_dot_pool<64, 8>), dim3((unsigned)((n_graphs + 3) / 4)), dim3(256), 0, st, x, ld_x, ptr, n_graphs, d, out, ld_out);
    }
    return check_launch("k_attention_dot_pool");
}

}  // namespace sss

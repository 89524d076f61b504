// Session-encoder kernels (gfx950): HeteroGGNN message passing + positional-attention pooling.
//
// Reference ops replaced (op semantics Appendix A):
//   k_linear_grouped    <- the dense node transforms inside PyG GATConv (lin_src), GatedGraphConv
//                          (x @ weight), torch GRUCell (W_ih, W_hh) and the nn.Linear layers of
//                          PositionalAttentionPooling (reference model/gnn.py:54,58,186-190)
//   k_gat_aggregate     <- GATConv.propagate: leaky_relu(0.2) + per-target softmax + weighted sum
//                          + bias, over a CSR-by-target adjacency (model/gnn.py:54 via HeteroConv)
//   k_csr_weighted_sum  <- GatedGraphConv.propagate (aggr='add') (model/gnn.py:58)
//   k_gru_combine       <- torch.nn.GRUCell gate math + HeteroConv 'sum' + relu (model/gnn.py:59,72)
//   k_pool_expand / k_segment_mean / k_pool_finish <- PositionalAttentionPooling.forward
//                          (model/gnn.py:193-217)
// Dense contractions run on v_mfma_f32_32x32x2_f32 (bit-exact f32 fma chains, the same rate as
// the f32 VALU peak); everything per-edge / per-node is HBM-bound and moves 16 B per lane.
#include "sss_common.h"
#include "kargs.h"

namespace sss {

int normalize_rows(float* x, long n, int d, long ld, float eps, int rule, hipStream_t st);    // rowops.hip

constexpr int LT = 64;      // GEMM tile rows (X) and columns (W rows): see k_linear_grouped below

int linear_grouped(LinBatch& b, hipStream_t st);

// ------------------------------------------------------------------------------------------
// Per target node i (CSR by target): e_ij = leaky_relu(as[j] + ad[i], 0.2);
// w_ij = exp(e_ij - max_j e) / (sum_j exp(e_ij - max) + 1e-16); out[i] = sum_j w_ij xs[j] + bias
// (+ relu).  Targets without incoming edges get bias.  One group of LPR lanes per target, each
// lane owns float4 columns lane, lane+LPR, ... of the h-wide row.
template <int LPR>
__global__ __launch_bounds__(256) void k_gat_aggregate(
    const float* __restrict__ xs, long ld_xs, const float* __restrict__ a_src, long ld_as,
    const float* __restrict__ a_dst, long ld_ad, const int* __restrict__ rowptr,
    const int* __restrict__ col, long n_dst, int h, const float* __restrict__ bias, int relu,
    long n_self_loop, float* __restrict__ out, long ld_out) {
    const int sub = threadIdx.x % LPR;
    const long per_block = 256 / LPR;
    const int nv = h / 4;
    auto score = [&](long j, float ad) {
        const float v = a_src[j * ld_as] + ad;
        return v > 0.f ? v : 0.2f * v;
    };
    for (long i = (long)blockIdx.x * per_block + threadIdx.x / LPR; i < n_dst; i += (long)gridDim.x * per_block) {
        const int e0 = rowptr[i], e1 = rowptr[i + 1];
        const float ad = a_dst[i * ld_ad];
        // n_self_loop > 0: PyG's self-loop rewrite on the fly (drop source == target, append one i -> i)
        const long skip = n_self_loop > 0 ? i : -1, self = i < n_self_loop ? i : -1;
        float mx = -INFINITY;
        for (int e = e0; e < e1; ++e)
            if (col[e] != skip) mx = fmaxf(mx, score(col[e], ad));
        if (self >= 0) mx = fmaxf(mx, score(self, ad));
        float den = 0.f;
        for (int e = e0; e < e1; ++e)
            if (col[e] != skip) den += expf(score(col[e], ad) - mx);
        if (self >= 0) den += expf(score(self, ad) - mx);
        const float inv = 1.f / (den + 1e-16f);
        for (int c = sub; c < nv; c += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            auto add = [&](long j) {
                const float w = expf(score(j, ad) - mx) * inv;
                const float4 x = *reinterpret_cast<const float4*>(xs + j * ld_xs + c * 4);
                acc.x += w * x.x; acc.y += w * x.y; acc.z += w * x.z; acc.w += w * x.w;
            };
            for (int e = e0; e < e1; ++e)
                if (col[e] != skip) add(col[e]);
            if (self >= 0) add(self);
            if (bias) {
                const float4 b = *reinterpret_cast<const float4*>(bias + c * 4);
                acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
            }
            if (relu) {
                acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f);
                acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
            }
            *reinterpret_cast<float4*>(out + i * ld_out + c * 4) = acc;
        }
    }
}

// out[i] = sum_{e in row i} (w[e] *) m[col[e]]   (GatedGraphConv aggregate, aggr='add')
template <int LPR>
__global__ __launch_bounds__(256) void k_csr_weighted_sum(const float* __restrict__ m, long ld_m,
                                                          const int* __restrict__ rowptr,
                                                          const int* __restrict__ col,
                                                          const float* __restrict__ w, long n_dst, int h,
                                                          float* __restrict__ out, long ld_out) {
    const int sub = threadIdx.x % LPR;
    const long per_block = 256 / LPR;
    const int nv = h / 4;
    for (long i = (long)blockIdx.x * per_block + threadIdx.x / LPR; i < n_dst; i += (long)gridDim.x * per_block) {
        const int e0 = rowptr[i], e1 = rowptr[i + 1];
        for (int c = sub; c < nv; c += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int e = e0; e < e1; ++e) {
                const float4 x = *reinterpret_cast<const float4*>(m + (long)col[e] * ld_m + c * 4);
                const float ww = w ? w[e] : 1.f;
                acc.x += ww * x.x; acc.y += ww * x.y; acc.z += ww * x.z; acc.w += ww * x.w;
            }
            *reinterpret_cast<float4*>(out + i * ld_out + c * 4) = acc;
        }
    }
}

// GRUCell(input = aggregated message, hidden = x) + the GAT term + relu:
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) n + z x
//   out = relu(add + h')        gi / gh already contain b_ih / b_hh.  x is zero-padded to h.
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__global__ __launch_bounds__(256) void k_gru_combine(const float* __restrict__ gi, long ld_gi,
                                                     const float* __restrict__ gh, long ld_gh,
                                                     const float* __restrict__ x, long ld_x, int d_x,
                                                     const float* __restrict__ add, long ld_add, long n,
                                                     int h, float* __restrict__ out, long ld_out) {
    const int nv = h / 4;
    const long total = n * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long i = idx / nv;
        const int c = (int)(idx % nv) * 4;
        const float4 ir = *reinterpret_cast<const float4*>(gi + i * ld_gi + c);
        const float4 iz = *reinterpret_cast<const float4*>(gi + i * ld_gi + h + c);
        const float4 in = *reinterpret_cast<const float4*>(gi + i * ld_gi + 2 * h + c);
        const float4 hr = *reinterpret_cast<const float4*>(gh + i * ld_gh + c);
        const float4 hz = *reinterpret_cast<const float4*>(gh + i * ld_gh + h + c);
        const float4 hn = *reinterpret_cast<const float4*>(gh + i * ld_gh + 2 * h + c);
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c + 3 < d_x) xv = *reinterpret_cast<const float4*>(x + i * ld_x + c);
        else {
            if (c + 0 < d_x) xv.x = x[i * ld_x + c + 0];
            if (c + 1 < d_x) xv.y = x[i * ld_x + c + 1];
            if (c + 2 < d_x) xv.z = x[i * ld_x + c + 2];
        }
        const float4 av = add ? *reinterpret_cast<const float4*>(add + i * ld_add + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 o;
#define SSS_GRU1(f)                                                        \
    {                                                                      \
        const float rr = sigmoidf_(ir.f + hr.f);                           \
        const float zz = sigmoidf_(iz.f + hz.f);                           \
        const float nn = tanhf(in.f + rr * hn.f);                          \
        const float hp = (1.f - zz) * nn + zz * xv.f;                      \
        o.f = fmaxf(av.f + hp, 0.f);                                       \
    }
        SSS_GRU1(x) SSS_GRU1(y) SSS_GRU1(z) SSS_GRU1(w)
#undef SSS_GRU1
        *reinterpret_cast<float4*>(out + i * ld_out + c) = o;
    }
}

// ---------------------------------------------------------------------------- pooling pieces
// Expanded pooling nodes (product clicks first, then queries), model/gnn.py:199-208:
//   node[e, 0:Dl] = tanh(lin[src_row[e], 0:Dl]),  node[e, Dl:Dl+P] = tanh(pos_emb[pos_id[e], :])
__global__ __launch_bounds__(256) void k_pool_expand(const float* __restrict__ lin_p, const float* __restrict__ lin_q,
                                                     long ld_lin, const int* __restrict__ src_row,
                                                     const int* __restrict__ pos_id, long n_clicks, long n_exp,
                                                     int Dl, int P, const float* __restrict__ pos_emb,
                                                     float* __restrict__ node, long ld_node) {
    const int D = Dl + P;
    const long total = n_exp * D;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long e = idx / D;
        const int c = (int)(idx % D);
        float v;
        if (c < Dl) {
            const float* lin = e < n_clicks ? lin_p : lin_q;
            v = lin[(long)src_row[e] * ld_lin + c];
        } else {
            v = pos_emb[(long)pos_id[e] * P + (c - Dl)];
        }
        node[e * ld_node + c] = tanhf(v);
    }
}

// mean over the expanded nodes of graph g: product-click rows [pptr[g], pptr[g+1]) and query rows
// n_clicks + [qptr[g], qptr[g+1]) (global_mean_pool, Appendix A.4).  One LPR-lane group per graph.
// weight == nullptr: plain mean.  Otherwise the attention-weighted mean of model/gnn.py:214-217:
//   att_n = sum_c watt[c] * sigmoid(A[n, c] + bcoarse[g, c]);  out[g] = mean_n(node[n] * att_n)
template <int LPR>
__global__ __launch_bounds__(256) void k_segment_pool(const float* __restrict__ node, long ld_node,
                                                      const int* __restrict__ pptr, const int* __restrict__ qptr,
                                                      long n_clicks, long n_graphs, int D,
                                                      const float* __restrict__ A, long ld_a,
                                                      const float* __restrict__ bcoarse, long ld_b,
                                                      const float* __restrict__ watt,
                                                      float* __restrict__ out, long ld_out, int reduce_sum = 0) {
    const int sub = threadIdx.x % LPR;
    const long per_block = 256 / LPR;
    const int nv = D / 4;
    for (long g = (long)blockIdx.x * per_block + threadIdx.x / LPR; g < n_graphs; g += (long)gridDim.x * per_block) {
        const int p0 = pptr[g], p1 = pptr[g + 1], q0 = qptr[g], q1 = qptr[g + 1];
        const int cnt = (p1 - p0) + (q1 - q0);
        const float invc = reduce_sum ? 1.f : 1.f / (float)(cnt > 0 ? cnt : 1);     // sum: SRGNN_Pooling's global_add_pool
        // Wide rows with attention (the reference's D = 1600: 400 float4 columns over 64 lanes): ONE sweep over the
        // graph's nodes -- a node's attention weight is computed once and applied to all of the lane's (up to eight)
        // columns; the column-chunked sweeps below would recompute it, and re-read the node's A row, per chunk.
        // Same additions in the same order per output element.
        if (watt && nv > 2 * LPR && nv <= 8 * LPR) {
            float4 acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 0; t < cnt; ++t) {
                const long nrow = t < (p1 - p0) ? (long)(p0 + t) : n_clicks + (long)(q0 + t - (p1 - p0));
                float part = 0.f;
                for (int c = sub; c < nv; c += LPR) {
                    const float4 a = *reinterpret_cast<const float4*>(A + nrow * ld_a + c * 4);
                    const float4 b = *reinterpret_cast<const float4*>(bcoarse + g * ld_b + c * 4);
                    const float4 w = *reinterpret_cast<const float4*>(watt + c * 4);
                    part += w.x * sigmoidf_(a.x + b.x) + w.y * sigmoidf_(a.y + b.y) +
                            w.z * sigmoidf_(a.z + b.z) + w.w * sigmoidf_(a.w + b.w);
                }
#pragma unroll
                for (int o = LPR / 2; o > 0; o >>= 1) part += __shfl_xor(part, o);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = sub + j * LPR;
                    if (c < nv) {
                        const float4 v = *reinterpret_cast<const float4*>(node + nrow * ld_node + c * 4);
                        acc[j].x += part * v.x; acc[j].y += part * v.y; acc[j].z += part * v.z; acc[j].w += part * v.w;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = sub + j * LPR;
                if (c < nv)
                    *reinterpret_cast<float4*>(out + g * ld_out + c * 4) =
                        make_float4(acc[j].x * invc, acc[j].y * invc, acc[j].z * invc, acc[j].w * invc);
            }
            continue;
        }
        // two float4 columns per lane per sweep (named accumulators, never runtime-indexed);
        // rows wider than 8 * LPR floats take several sweeps over the graph's nodes
        for (int cb = 0; cb < nv; cb += 2 * LPR) {
            const int c0 = cb + sub, c1 = cb + sub + LPR;
            float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 0; t < cnt; ++t) {
                const long nrow = t < (p1 - p0) ? (long)(p0 + t) : n_clicks + (long)(q0 + t - (p1 - p0));
                float att = 1.f;
                if (watt) {
                    float part = 0.f;
                    for (int c = sub; c < nv; c += LPR) {
                        const float4 a = *reinterpret_cast<const float4*>(A + nrow * ld_a + c * 4);
                        const float4 b = *reinterpret_cast<const float4*>(bcoarse + g * ld_b + c * 4);
                        const float4 w = *reinterpret_cast<const float4*>(watt + c * 4);
                        part += w.x * sigmoidf_(a.x + b.x) + w.y * sigmoidf_(a.y + b.y) +
                                w.z * sigmoidf_(a.z + b.z) + w.w * sigmoidf_(a.w + b.w);
                    }
#pragma unroll
                    for (int o = LPR / 2; o > 0; o >>= 1) part += __shfl_xor(part, o);
                    att = part;
                }
                if (c0 < nv) {
                    const float4 v = *reinterpret_cast<const float4*>(node + nrow * ld_node + c0 * 4);
                    acc0.x += att * v.x; acc0.y += att * v.y; acc0.z += att * v.z; acc0.w += att * v.w;
                }
                if (c1 < nv) {
                    const float4 v = *reinterpret_cast<const float4*>(node + nrow * ld_node + c1 * 4);
                    acc1.x += att * v.x; acc1.y += att * v.y; acc1.z += att * v.z; acc1.w += att * v.w;
                }
            }
            if (c0 < nv)
                *reinterpret_cast<float4*>(out + g * ld_out + c0 * 4) =
                    make_float4(acc0.x * invc, acc0.y * invc, acc0.z * invc, acc0.w * invc);
            if (c1 < nv)
                *reinterpret_cast<float4*>(out + g * ld_out + c1 * 4) =
                    make_float4(acc1.x * invc, acc1.y * invc, acc1.z * invc, acc1.w * invc);
        }
    }
}

// ptr[g] = first index i with batch[i] >= g (batch sorted ascending), g in [0, n_graphs]
__global__ void k_segment_ptr(const long* __restrict__ batch, long n, long n_graphs, int* __restrict__ ptr) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_graphs) return;
    long lo = 0, hi = n;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (batch[mid] < g) lo = mid + 1; else hi = mid;
    }
    ptr[g] = (int)lo;
}

// ------------------------------------------------------------------------------ host launchers
static int lanes_for(int h) {
    const int nv = h / 4;
    int l = 1;
    while (l < nv && l < 64) l <<= 1;
    return l;
}
static unsigned grid_rows(long n, int lpr) {
    const long rpb = 256 / lpr;
    long g = (n + rpb - 1) / rpb;
    if (g > 4096) g = 4096;
    return (unsigned)(g < 1 ? 1 : g);
}
#define SSS_LPR_SWITCH(lpr, CALL)                       \
    switch (lpr) {                                      \
        case 1: { constexpr int L = 1; CALL; } break;   \
        case 2: { constexpr int L = 2; CALL; } break;   \
        case 4: { constexpr int L = 4; CALL; } break;   \
        case 8: { constexpr int L = 8; CALL; } break;   \
        case 16: { constexpr int L = 16; CALL; } break; \
        case 32: { constexpr int L = 32; CALL; } break; \
        default: { constexpr int L = 64; CALL; } break; \
    }

// Y[N, M] = X[N, K] * W[M, K]^T (+ bias[M]): one problem of the grouped kernel (round 3: the separate 64 x 64
// kernel this entry point used to launch computed the identical fma chains 8-20 % slower on every shape).
int linear_f32(const float* X, long ldx, const float* W, long ldw, const float* bias, float* Y, long ldy,
               long N, int M, int K, hipStream_t st) {
    if (N < 0 || M <= 0 || K <= 0 || K % 32 || ldx % 4 || ldw % 4 || ldx < K || ldw < K || ldy < M) {
        set_error("linear: need K %% 32 == 0, ldx/ldw %% 4 == 0 and >= K, ldy >= M (N=%ld M=%d K=%d)", N, M, K);
        return SSS_EINVAL;
    }
    if (N == 0) return SSS_OK;
    LinBatch b = {};
    b.nprob = 1; b.K = K;
    LinProb& p = b.p[0];
    p.x = X; p.ldx = ldx; p.ids = nullptr; p.table = nullptr; p.xcopy = nullptr; p.ld_xcopy = 0;
    p.w = W; p.ldw = ldw; p.bias = bias; p.y = Y; p.ldy = ldy; p.n = N; p.m = M; p.act = 0;
    return linear_grouped(b, st);
}

int gat_aggregate(const float* xs, long ld_xs, const float* a_src, long ld_as, const float* a_dst, long ld_ad,
                  const int* rowptr, const int* col, long n_dst, int h, const float* bias, int relu, long n_self_loop,
                  float* out, long ld_out, hipStream_t st) {
    if (n_dst < 0 || h <= 0 || h % 4 || ld_xs % 4 || ld_out % 4 || ld_out < h) {
        set_error("gat_aggregate: need h %% 4 == 0 and 16-byte aligned row strides");
        return SSS_EINVAL;
    }
    if (n_dst == 0) return SSS_OK;
    const int lpr = lanes_for(h);
    SSS_LPR_SWITCH(lpr, hipLaunchKernelGGL(k_gat_aggregate<L>, dim3(grid_rows(n_dst, L)), dim3(256), 0, st, xs, ld_xs,
                                           a_src, ld_as, a_dst, ld_ad, rowptr, col, n_dst, h, bias, relu, n_self_loop, out,
                                           ld_out));
    return check_launch("k_gat_aggregate");
}

int csr_weighted_sum(const float* m, long ld_m, const int* rowptr, const int* col, const float* w, long n_dst,
                     int h, float* out, long ld_out, hipStream_t st) {
    if (n_dst < 0 || h <= 0 || h % 4 || ld_m % 4 || ld_out % 4 || ld_out < h) {
        set_error("csr_weighted_sum: need h %% 4 == 0 and 16-byte aligned row strides");
        return SSS_EINVAL;
    }
    if (n_dst == 0) return SSS_OK;
    const int lpr = lanes_for(h);
    SSS_LPR_SWITCH(lpr, hipLaunchKernelGGL(k_csr_weighted_sum<L>, dim3(grid_rows(n_dst, L)), dim3(256), 0, st, m, ld_m,
                                           rowptr, col, w, n_dst, h, out, ld_out));
    return check_launch("k_csr_weighted_sum");
}

int gru_combine(const float* gi, long ld_gi, const float* gh, long ld_gh, const float* x, long ld_x, int d_x,
                const float* add, long ld_add, long n, int h, float* out, long ld_out, hipStream_t st) {
    if (n < 0 || h <= 0 || h % 4 || d_x > h || ld_gi % 4 || ld_gh % 4 || ld_out % 4 || (add && ld_add % 4) ||
        (d_x >= 4 && ld_x % 4)) {
        set_error("gru_combine: need h %% 4 == 0, d_x <= h and 16-byte aligned row strides");
        return SSS_EINVAL;
    }
    if (n == 0) return SSS_OK;
    long blocks = (n * (h / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gru_combine, dim3((unsigned)blocks), dim3(256), 0, st, gi, ld_gi, gh, ld_gh, x, ld_x, d_x, add,
                       ld_add, n, h, out, ld_out);
    return check_launch("k_gru_combine");
}

int pool_expand(const float* lin_p, const float* lin_q, long ld_lin, const int* src_row, const int* pos_id,
                long n_clicks, long n_exp, int Dl, int P, const float* pos_emb, float* node, long ld_node,
                hipStream_t st) {
    if (n_exp < 0 || n_clicks < 0 || n_clicks > n_exp || Dl <= 0 || P < 0 || ld_node < Dl + P) {
        set_error("pool_expand: bad arguments");
        return SSS_EINVAL;
    }
    if (n_exp == 0) return SSS_OK;
    long blocks = (n_exp * (Dl + P) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pool_expand, dim3((unsigned)blocks), dim3(256), 0, st, lin_p, lin_q, ld_lin, src_row, pos_id,
                       n_clicks, n_exp, Dl, P, pos_emb, node, ld_node);
    return check_launch("k_pool_expand");
}

int segment_pool(const float* node, long ld_node, const int* pptr, const int* qptr, long n_clicks, long n_graphs,
                 int D, const float* A, long ld_a, const float* bcoarse, long ld_b, const float* watt, float* out,
                 long ld_out, hipStream_t st) {
    if (n_graphs < 0 || D <= 0 || D % 4 || ld_node % 4 || ld_out % 4 ||
        (watt && (!A || !bcoarse || ld_a % 4 || ld_b % 4))) {
        set_error("segment_pool: need D %% 4 == 0 and 16-byte aligned row strides");
        return SSS_EINVAL;
    }
    if (n_graphs == 0) return SSS_OK;
    const int lpr = lanes_for(D);
    SSS_LPR_SWITCH(lpr, hipLaunchKernelGGL(k_segment_pool<L>, dim3(grid_rows(n_graphs, L)), dim3(256), 0, st, node,
                                           ld_node, pptr, qptr, n_clicks, n_graphs, D, A, ld_a, bcoarse, ld_b, watt,
                                           out, ld_out));
    return check_launch("k_segment_pool");
}

int segment_ptr(const long* batch, long n, long n_graphs, int* ptr, hipStream_t st) {
    if (n < 0 || n_graphs < 0) { set_error("segment_ptr: bad arguments"); return SSS_EINVAL; }
    hipLaunchKernelGGL(k_segment_ptr, dim3((unsigned)((n_graphs + 1 + 255) / 256)), dim3(256), 0, st, batch, n, n_graphs, ptr);
    return check_launch("k_segment_ptr");
}

}  // namespace sss

// =============================================================================================
// Fused encoder path (8 launches per forward; DESIGN.md "encoder"):
//   k_linear_grouped   one launch for up to 4 node-linear problems (products + queries of a layer,
//                      or the pooling's lin_p + lin_q, or node_lin + coarse_lin); layer 0 reads its
//                      X rows straight from the feature tables (NodeAsinEmbedding gather fused in)
//   k_layer_update     per target node: GAT segment-softmax aggregate + GatedGraphConv aggregate +
//                      GRU gates + HeteroConv sum + relu, everything in registers
//   k_pool_expand_mean tanh([lin ; pos]) rows + per-graph mean
//   k_pool_attention   attention-weighted per-graph mean (+ optional L2 normalise)
namespace sss {


template <int KC>
__global__ __launch_bounds__(256, 2) void k_linear_grouped(const LinBatch B) {
    constexpr int CPR = KC / 4;
    constexpr int NLD = LT * CPR / 256;
    constexpr int RS = CPR;                     // LDS row stride in 16-byte chunks
    constexpr int SW = RS < 16 ? RS - 1 : 15;   // XOR swizzle mask (stays inside the row)
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    float* lx = lds_raw;
    float* lw = lds_raw + LT * RS * 4;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < B.nprob && (int)blockIdx.x >= B.p[i].tile_begin) pi = i;
    // copy the chosen problem's fields through scalar selects (no runtime-indexed struct access)
    const float* X = B.p[0].x; long ldx = B.p[0].ldx; const long* ids = B.p[0].ids; const float* table = B.p[0].table;
    float* xcopy = B.p[0].xcopy; long ldc = B.p[0].ld_xcopy; const float* W = B.p[0].w; long ldw = B.p[0].ldw;
    const float* bias = B.p[0].bias; float* Y = B.p[0].y; long ldy = B.p[0].ldy; long N = B.p[0].n; int M = B.p[0].m;
    int tiles_m = B.p[0].tiles_m, tile_begin = B.p[0].tile_begin, act = B.p[0].act;
    const float* post_s = B.p[0].post_scale; const float* post_t = B.p[0].post_shift;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (pi == i) {
            act = B.p[i].act; post_s = B.p[i].post_scale; post_t = B.p[i].post_shift;
            X = B.p[i].x; ldx = B.p[i].ldx; ids = B.p[i].ids; table = B.p[i].table; xcopy = B.p[i].xcopy; ldc = B.p[i].ld_xcopy;
            W = B.p[i].w; ldw = B.p[i].ldw; bias = B.p[i].bias; Y = B.p[i].y; ldy = B.p[i].ldy; N = B.p[i].n; M = B.p[i].m;
            tiles_m = B.p[i].tiles_m; tile_begin = B.p[i].tile_begin;
        }
    const int K = B.K;
    const int t = (int)blockIdx.x - tile_begin;
    const long row0 = (long)(t / tiles_m) * LT;
    const int col0 = (t % tiles_m) * LT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int xr = wr * 32 + r, wrow = wc * 32 + r;

    f32x16 acc = {0};
    f32x4 sx[NLD], sw[NLD];          // register staging of one K chunk (ext-vector type: stays in VGPRs)
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid + 256 * i;
            const int tr = p / CPR, c = p % CPR;
            long gr = row0 + tr; if (gr > N - 1) gr = N - 1;
            int gw = col0 + tr; if (gw > M - 1) gw = M - 1;
            const float* xrow = ids ? table + ids[gr] * (long)K : X + gr * ldx;
            sx[i] = *reinterpret_cast<const f32x4*>(xrow + k0 + c * 4);
            sw[i] = *reinterpret_cast<const f32x4*>(W + (long)gw * ldw + k0 + c * 4);
            if (ids && xcopy && col0 == 0 && row0 + tr < N)          // the gathered rows = slice 0 of the node buffer
                *reinterpret_cast<f32x4*>(xcopy + gr * ldc + k0 + c * 4) = sx[i];
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += KC) {
        if (k0 > 0) __syncthreads();                         // previous chunk fully consumed
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid + 256 * i;
            const int tr = p / CPR, c = p % CPR;
            const int cs = c ^ (tr & SW);
            *reinterpret_cast<f32x4*>(lx + (tr * RS + cs) * 4) = sx[i];
            *reinterpret_cast<f32x4*>(lw + (tr * RS + cs) * 4) = sw[i];
        }
        __syncthreads();
        if (k0 + KC < K) fetch(k0 + KC);                     // next chunk's loads fly under this chunk's MFMAs
        float4 a = *reinterpret_cast<const float4*>(lx + (xr * RS + (h ^ (xr & SW))) * 4);
        float4 b = *reinterpret_cast<const float4*>(lw + (wrow * RS + (h ^ (wrow & SW))) * 4);
#pragma unroll
        for (int u = 0; u < KC / 8; ++u) {
            float4 na = a, nb = b;
            if (u + 1 < KC / 8) {
                na = *reinterpret_cast<const float4*>(lx + (xr * RS + ((2 * u + 2 + h) ^ (xr & SW))) * 4);
                nb = *reinterpret_cast<const float4*>(lw + (wrow * RS + ((2 * u + 2 + h) ^ (wrow & SW))) * 4);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            a = na; b = nb;
        }
    }
    const int col = col0 + wc * 32 + r;
    if (col < M) {
        const float bv = bias ? bias[col] : 0.f;
        const float ps = post_s ? post_s[col] : 1.f, pt = post_s ? post_t[col] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const long row = row0 + wr * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
            float v = acc[j] + bv;
            if (act == 1) v = fmaxf(v, 0.f);
            else if (act == 2) v = tanhf(v);
            else if (act == 3) v = v > 0.f ? 1.f : v < 0.f ? -1.f : v;     // torch.sign (0 and NaN pass through)
            else if (act == 4) v = tanhf(tanhf(v));
            if (post_s) {
#pragma clang fp contract(off)                                               // multiply, round, add, round: no fused fma here
                v = fmaxf(v * ps + pt, 0.f);                                // bn(x) in eval mode as scale / shift, then relu
            }
            if (row < N) Y[row * ldy + col] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// One LPR-lane group per TARGET node (LPR = h / 4: a lane owns one float4 column of the h-wide
// row).  Targets [0, Np) are products, [Np, Np + Nq) queries.
//   product i:  t1 = GAT(query -> product)  (Appendix A.2: leaky_relu(0.2), per-target softmax,
//                    + 1e-16 in the denominator, + bias)
//               gi = sum_{j -> i} w_ij u[j] + b_ih   with u = x (W_g W_ih^T): the GatedGraphConv
//                    message transform and the GRU input transform applied BEFORE the (linear)
//                    aggregation, so no second GEMM is needed (Appendix A.3)
//               out = relu(t1 + (1 - z) n + z x),  r, z, n = GRU gates of (gi, gh = W_hh x + b_hh)
//   query i:    out = relu(GAT(product -> query))
// Column layout of the node-linear outputs (written by k_linear_grouped, see encoder.py):
//   Yp [Np, 7h+32]: xs_p | u_r u_z u_n | gh_r gh_z gh_n | a_src(pq) a_dst(qp)
//   Yq [Nq,  h+32]: xs_q | a_src(qp) a_dst(pq)

__device__ __forceinline__ float4 f4_fma(float w, float4 x, float4 a) {
    a.x += w * x.x; a.y += w * x.y; a.z += w * x.z; a.w += w * x.w;
    return a;
}
__device__ __forceinline__ float leaky02(float v) { return v > 0.f ? v : 0.2f * v; }

// softmax-weighted sum over the incoming edges of one target: returns sum_e softmax_e * xs[col[e]][c4] + bias.
// self_i >= 0 applies PyG's GATConv(add_self_loops=True) edge rewrite on the fly (Appendix A.2):
// edges whose source index equals the target index are dropped and ONE edge self_i -> target is
// appended after the target's other edges (what remove_self_loops + add_self_loops + a stable
// sort by target produce).
__device__ __forceinline__ float4 gat_row(const float* xs, long ld, const float* a_src_col, float ad, const int* col,
                                          int e0, int e1, int c4, const float* bias, long skip_i, long self_i,
                                          const long* row_of = nullptr) {
    // one pass (online softmax): a new maximum rescales what has been summed so far, so every
    // edge's index / score / row is loaded exactly once
    float mx = -INFINITY, den = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto edge = [&](long jn) {
        const long j = row_of ? row_of[jn] : jn;           // table mode: the source node's table row
        const float v = leaky02(a_src_col[j * ld] + ad);
        const float4 x = *reinterpret_cast<const float4*>(xs + j * ld + c4);
        if (v > mx) {
            const float sc = expf(mx - v);                 // exp(-inf) = 0 on the first edge
            den *= sc; acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc;
            mx = v;
        }
        const float ex = expf(v - mx);
        den += ex;
        acc = f4_fma(ex, x, acc);
    };
    for (int e = e0; e < e1; ++e) {
        const long j = col[e];
        if (j != skip_i) edge(j);
    }
    if (self_i >= 0) edge(self_i);
    const float inv = 1.f / (den + 1e-16f);
    const float4 b = *reinterpret_cast<const float4*>(bias + c4);
    return make_float4(acc.x * inv + b.x, acc.y * inv + b.y, acc.z * inv + b.z, acc.w * inv + b.w);
}

template <int LPR>
__global__ __launch_bounds__(256) void k_layer_update(const LayerArgs A) {
    const int sub = threadIdx.x % LPR;
    const long per_block = 256 / LPR;
    const long t = (long)blockIdx.x * per_block + threadIdx.x / LPR;
    const int h = A.h, c4 = sub * 4;
    if (c4 >= h || t >= A.Np + A.Nq) return;
    if (t >= A.Np) {                                            // ---- query target
        const long i = t - A.Np;
        const long ri = A.row_q ? A.row_q[i] : i;
        const float ad = A.Yq[ri * A.ldyq + h + 1];
        const bool lp = i < A.n_self_loop;
        float4 o = gat_row(A.Yp, A.ldyp, A.Yp + 7 * h, ad, A.col_pq, A.rowptr_pq[i], A.rowptr_pq[i + 1], c4, A.bias_pq,
                           A.n_self_loop > 0 ? i : -1, lp ? i : -1, A.row_p);
        if (A.x0_q) {                                           // raw query features -> slice 0 of the node buffer
            const float* xq = A.xq_table + ri * A.ld_xq;
            float* dq = A.x0_q + i * A.ld_x0q;
            if (c4 + 3 < A.d_x) *reinterpret_cast<float4*>(dq + c4) = *reinterpret_cast<const float4*>(xq + c4);
            else for (int u = 0; u < 4; ++u) if (c4 + u < A.d_x) dq[c4 + u] = xq[c4 + u];
        }
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        *reinterpret_cast<float4*>(A.out_q + i * A.ld_outq + c4) = o;
        return;
    }
    const long i = t;                                           // ---- product target
    const long ri = A.row_p ? A.row_p[i] : i;
    const float* yi = A.Yp + ri * A.ldyp;
    const float4 t1 = gat_row(A.Yq, A.ldyq, A.Yq + h, yi[7 * h + 1], A.col_qp, A.rowptr_qp[i], A.rowptr_qp[i + 1], c4, A.bias_qp,
                              A.n_self_loop > 0 ? i : -1, i < A.n_self_loop ? i : -1, A.row_q);
    float4 gr = *reinterpret_cast<const float4*>(A.b_ih + c4);
    float4 gz = *reinterpret_cast<const float4*>(A.b_ih + h + c4);
    float4 gn = *reinterpret_cast<const float4*>(A.b_ih + 2 * h + c4);
    for (int e = A.rowptr_pp[i]; e < A.rowptr_pp[i + 1]; ++e) {
        const long jn = A.col_pp[e];
        const float* uj = A.Yp + (A.row_p ? A.row_p[jn] : jn) * A.ldyp + h + c4;
        const float w = A.w_pp ? A.w_pp[e] : 1.f;
        gr = f4_fma(w, *reinterpret_cast<const float4*>(uj), gr);
        gz = f4_fma(w, *reinterpret_cast<const float4*>(uj + h), gz);
        gn = f4_fma(w, *reinterpret_cast<const float4*>(uj + 2 * h), gn);
    }
    const float4 hr = *reinterpret_cast<const float4*>(yi + 4 * h + c4);
    const float4 hz = *reinterpret_cast<const float4*>(yi + 5 * h + c4);
    const float4 hn = *reinterpret_cast<const float4*>(yi + 6 * h + c4);
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xi = A.xin_p + ri * A.ld_xin;
    if (c4 + 3 < A.d_x) xv = *reinterpret_cast<const float4*>(xi + c4);
    else {
        if (c4 + 0 < A.d_x) xv.x = xi[c4 + 0];
        if (c4 + 1 < A.d_x) xv.y = xi[c4 + 1];
        if (c4 + 2 < A.d_x) xv.z = xi[c4 + 2];
    }
    if (A.x0_p) {                                               // raw product features -> slice 0 of the node buffer
        float* dp = A.x0_p + i * A.ld_x0p;
        if (c4 + 3 < A.d_x) *reinterpret_cast<float4*>(dp + c4) = xv;
        else {
            if (c4 + 0 < A.d_x) dp[c4 + 0] = xv.x;
            if (c4 + 1 < A.d_x) dp[c4 + 1] = xv.y;
            if (c4 + 2 < A.d_x) dp[c4 + 2] = xv.z;
        }
    }
    float4 o;
#define SSS_GRU1(f)                                                        \
    {                                                                      \
        const float rr = sigmoidf_(gr.f + hr.f);                           \
        const float zz = sigmoidf_(gz.f + hz.f);                           \
        const float nn = tanhf(gn.f + rr * hn.f);                          \
        o.f = fmaxf(t1.f + (1.f - zz) * nn + zz * xv.f, 0.f);              \
    }
    SSS_GRU1(x) SSS_GRU1(y) SSS_GRU1(z) SSS_GRU1(w)
#undef SSS_GRU1
    *reinterpret_cast<float4*>(A.out_p + i * A.ld_outp + c4) = o;
}

// ------------------------------------------------------------------------------------------
// PositionalAttentionPooling, first half (model/gnn.py:195-211): one LPR-lane group per graph g
// walks its expanded rows (product clicks [pptr[g], pptr[g+1]), then query rows
// n_clicks + [qptr[g], qptr[g+1])), writes node[e] = tanh([lin[src_row[e]] ; pos_emb[pos_id[e]]])
// and the graph mean coarse[g] (global_mean_pool).
template <int LPR>
__global__ __launch_bounds__(256) void k_pool_expand_mean(const float* __restrict__ lin_p, const float* __restrict__ lin_q,
                                                          long ld_lin, const int* __restrict__ src_row,
                                                          const int* __restrict__ pos_id, const int* __restrict__ pptr,
                                                          const int* __restrict__ qptr, long n_clicks, long n_graphs,
                                                          int Dl, int P, const float* __restrict__ pos_emb,
                                                          float* __restrict__ node, long ld_node,
                                                          float* __restrict__ coarse, long ld_coarse) {
    const int sub = threadIdx.x % LPR;
    const long g = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    const int D = Dl + P, c4 = sub * 4;
    if (g >= n_graphs || c4 >= D) return;
    const int p0 = pptr[g], p1 = pptr[g + 1], q0 = qptr[g], q1 = qptr[g + 1];
    const int np_ = p1 - p0, cnt = np_ + (q1 - q0);
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    // Rows are independent: four rows' loads are issued before the first tanh so their latencies
    // overlap (the per-row chain src_row -> lin row would otherwise serialise the whole graph).
    auto row_of = [&](int t) -> long { return t < np_ ? (long)(p0 + t) : n_clicks + (long)(q0 + t - np_); };
    auto load_row = [&](int t, float (&v)[4], long& e) {
        e = row_of(t);
        const float* lin = (t < np_ ? lin_p : lin_q) + (long)src_row[e] * ld_lin;
        const float* pe = pos_emb + (long)pos_id[e] * P;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c4 + u;
            v[u] = c < Dl ? lin[c] : pe[c - Dl];
        }
    };
    auto finish_row = [&](const float (&v)[4], long e) {
        const float4 o = make_float4(tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3]));
        *reinterpret_cast<float4*>(node + e * ld_node + c4) = o;
        m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
    };
    int t = 0;
    for (; t + 4 <= cnt; t += 4) {
        float v0[4], v1[4], v2[4], v3[4];
        long e0, e1, e2, e3;
        load_row(t, v0, e0); load_row(t + 1, v1, e1); load_row(t + 2, v2, e2); load_row(t + 3, v3, e3);
        finish_row(v0, e0); finish_row(v1, e1); finish_row(v2, e2); finish_row(v3, e3);
    }
    for (; t < cnt; ++t) {
        float v0[4];
        long e0;
        load_row(t, v0, e0);
        finish_row(v0, e0);
    }
    const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
    *reinterpret_cast<float4*>(coarse + g * ld_coarse + c4) = make_float4(m.x * inv, m.y * inv, m.z * inv, m.w * inv);
}

// Second half (model/gnn.py:212-217): att_e = watt . sigmoid(a[e] + b[g]);  out[g] = mean_e(node[e] * att_e);
// normalize != 0 additionally applies the reference's `normalize` (util_amazon_filtered.py:28-31,
// x / sqrt(max(sum x^2, eps))) to the row, saving a pass over the session vectors.
template <int LPR>
__global__ __launch_bounds__(256) void k_pool_attention(const float* __restrict__ node, long ld_node,
                                                        const float* __restrict__ Aa, long ld_a,
                                                        const float* __restrict__ Bc, long ld_b,
                                                        const float* __restrict__ watt, const int* __restrict__ pptr,
                                                        const int* __restrict__ qptr, long n_clicks, long n_graphs,
                                                        int D, int normalize, float eps, int reduce_sum,
                                                        float* __restrict__ out, long ld_out) {
    const int sub = threadIdx.x % LPR;
    const long g = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    const int c4 = sub * 4;
    const bool live = g < n_graphs && c4 < D;                   // dead lanes still take part in the shuffles
    int p0 = 0, p1 = 0, q0 = 0, q1 = 0;
    if (g < n_graphs) { p0 = pptr[g]; p1 = pptr[g + 1]; q0 = qptr[g]; q1 = qptr[g + 1]; }
    const int cnt = (p1 - p0) + (q1 - q0);
    float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = w4, acc = w4;
    if (live) { w4 = *reinterpret_cast<const float4*>(watt + c4); b4 = *reinterpret_cast<const float4*>(Bc + g * ld_b + c4); }
    auto row_of = [&](int t) -> long { return t < (p1 - p0) ? (long)(p0 + t) : n_clicks + (long)(q0 + t - (p1 - p0)); };
    auto load_row = [&](int t, float4& a, float4& v) {
        a = make_float4(0.f, 0.f, 0.f, 0.f); v = a;
        if (live && t < cnt) {
            const long e = row_of(t);
            a = *reinterpret_cast<const float4*>(Aa + e * ld_a + c4);
            v = *reinterpret_cast<const float4*>(node + e * ld_node + c4);
        }
    };
    auto finish_row = [&](const float4& a, const float4& v) {      // every lane of the group takes part in the shuffles
        float part = live ? w4.x * sigmoidf_(a.x + b4.x) + w4.y * sigmoidf_(a.y + b4.y) + w4.z * sigmoidf_(a.z + b4.z) +
                                w4.w * sigmoidf_(a.w + b4.w) : 0.f;
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) part += __shfl_xor(part, o);
        acc = f4_fma(part, v, acc);
    };
    // cnt is uniform inside a lane group but differs between the groups of a wave: pad to the
    // wave-wide maximum so the shuffles stay convergent; padded rows load nothing and add 0.
    int cmax = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
    for (int t = 0; t < cmax; t += 4) {                             // four rows' loads in flight
        float4 a0, v0, a1, v1, a2, v2, a3, v3;
        load_row(t, a0, v0); load_row(t + 1, a1, v1); load_row(t + 2, a2, v2); load_row(t + 3, a3, v3);
        finish_row(a0, v0); finish_row(a1, v1); finish_row(a2, v2); finish_row(a3, v3);
    }
    const float inv = reduce_sum ? 1.f : 1.f / (float)(cnt > 0 ? cnt : 1);       // sum: SRGNN_Pooling's global_add_pool
    acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    if (normalize) {
        float ss = acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float den = sqrtf(fmaxf(ss, eps));
        acc.x /= den; acc.y /= den; acc.z /= den; acc.w /= den;
    }
    if (live) *reinterpret_cast<float4*>(out + g * ld_out + c4) = acc;
}

// PositionalAttentionPooling without materialising the expanded rows (3 launches instead of 4).
// An expanded row e is [tanh(lin[src(e)]) ; tanh(pos_emb[pid(e)])]: every LINEAR map of it splits into
// a per-NODE part and a per-POSITION table entry,
//     node_emb_lin(row e)   = A1[src(e)] + A2tab[pid(e)]      A1 = T Wn[:, :Dl]^T,  A2tab = tanh(P) Wn[:, Dl:]^T + bn
//     coarse_rep_lin(row e) = C1[src(e)] + C2tab[pid(e)]      C1 = T Wc[:, :Dl]^T,  C2tab = tanh(P) Wc[:, Dl:]^T
// with T = tanh(lin) per node (GEMM epilogue).  coarse_rep_lin of the graph MEAN is the mean of the
// rows' images, so one lane group per graph needs two passes over its rows and no [n_exp, D] buffers:
//   pass 1: bc = mean_e(C1[src] + C2tab[pid])
//   pass 2: att_e = watt . sigmoid(A1[src] + A2tab[pid] + bc);  out = mean_e(row_e * att_e)  (+ normalise)
// T / AC rows: products [0, Np), queries [Np, Np + Nq).  AC = [A1 | C1], 2D wide.
// One WAVE per graph: LPR lanes own the float4 columns of a row, the wave's 64 / LPR row slots take
// rows slot, slot + RS, ...; the (row -> node, position) indices of the whole graph are loaded once,
// one row per lane, and broadcast with shuffles, so both passes issue their gathers without a
// dependent index load in front.
template <int LPR>
__global__ __launch_bounds__(256) void k_pool_attention_tab(const float* __restrict__ T, long ld_t, const float* __restrict__ AC,
                                                            long ld_ac, const float* __restrict__ tanhpos,
                                                            const float* __restrict__ A2tab, const float* __restrict__ C2tab,
                                                            const float* __restrict__ watt, const int* __restrict__ src_row,
                                                            const int* __restrict__ pos_id, const int* __restrict__ pptr,
                                                            const int* __restrict__ qptr, long n_clicks, long Np,
                                                            long n_graphs, int Dl, int P, int normalize, float eps,
                                                            float* __restrict__ out, long ld_out) {
    constexpr int RS = 64 / LPR;                                 // row slots per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, slot = lane / LPR;
    const long g = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (g >= n_graphs) return;                                   // whole wave
    const int D = Dl + P, c4 = sub * 4;
    const bool colok = c4 < D;
    const int p0 = pptr[g], p1 = pptr[g + 1], q0 = qptr[g], q1 = qptr[g + 1];
    const int np_ = p1 - p0, cnt = np_ + (q1 - q0);
    const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
    float4 bc = make_float4(0.f, 0.f, 0.f, 0.f), acc = bc, w4 = bc;
    if (colok) w4 = *reinterpret_cast<const float4*>(watt + c4);
    for (int base = 0; base < cnt; base += 64) {                 // graphs longer than 64 rows: chunks (bc needs all rows first)
        const int t_l = base + lane;
        int my_node = 0, my_pid = 0;
        if (t_l < cnt) {
            const long e = t_l < np_ ? (long)(p0 + t_l) : n_clicks + (long)(q0 + t_l - np_);
            my_node = (int)((t_l < np_ ? 0 : Np) + src_row[e]);
            my_pid = pos_id[e];
        }
        const int nrow = min(64, cnt - base);
        for (int r0 = 0; r0 < nrow; r0 += RS) {                  // pass 1 (this chunk): sum of C1[node] + C2tab[pid]
            const int r = r0 + slot;
            const int node = __shfl(my_node, r & 63), pid = __shfl(my_pid, r & 63);
            if (r < nrow && colok) {
                const float4 v = *reinterpret_cast<const float4*>(AC + (long)node * ld_ac + D + c4);
                const float4 w = *reinterpret_cast<const float4*>(C2tab + (long)pid * D + c4);
                bc.x += v.x + w.x; bc.y += v.y + w.y; bc.z += v.z + w.z; bc.w += v.w + w.w;
            }
        }
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {                         // combine the row slots
        bc.x += __shfl_xor(bc.x, o); bc.y += __shfl_xor(bc.y, o); bc.z += __shfl_xor(bc.z, o); bc.w += __shfl_xor(bc.w, o);
    }
    bc.x *= inv; bc.y *= inv; bc.z *= inv; bc.w *= inv;
    for (int base = 0; base < cnt; base += 64) {                 // pass 2: attention-weighted sum
        const int t_l = base + lane;
        int my_node = 0, my_pid = 0;
        if (t_l < cnt) {
            const long e = t_l < np_ ? (long)(p0 + t_l) : n_clicks + (long)(q0 + t_l - np_);
            my_node = (int)((t_l < np_ ? 0 : Np) + src_row[e]);
            my_pid = pos_id[e];
        }
        const int nrow = min(64, cnt - base);
        for (int r0 = 0; r0 < nrow; r0 += RS) {
            const int r = r0 + slot;
            const int node = __shfl(my_node, r & 63), pid = __shfl(my_pid, r & 63);
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            float part = 0.f;
            if (r < nrow && colok) {
                const float4 a1 = *reinterpret_cast<const float4*>(AC + (long)node * ld_ac + c4);
                const float4 a2 = *reinterpret_cast<const float4*>(A2tab + (long)pid * D + c4);
                float xv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c4 + j;
                    xv[j] = c < Dl ? T[(long)node * ld_t + c] : tanhpos[(long)pid * P + (c - Dl)];
                }
                x = make_float4(xv[0], xv[1], xv[2], xv[3]);
                part = w4.x * sigmoidf_(a1.x + a2.x + bc.x) + w4.y * sigmoidf_(a1.y + a2.y + bc.y) +
                       w4.z * sigmoidf_(a1.z + a2.z + bc.z) + w4.w * sigmoidf_(a1.w + a2.w + bc.w);
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) part += __shfl_xor(part, o);
            acc = f4_fma(part, x, acc);
        }
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    }
    acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    if (normalize) {
        float ss = acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float den = sqrtf(fmaxf(ss, eps));
        acc.x /= den; acc.y /= den; acc.z /= den; acc.w /= den;
    }
    if (slot == 0 && colok) *reinterpret_cast<float4*>(out + g * ld_out + c4) = acc;
}

// ------------------------------------------------------------------------------ host launchers
int linear_grouped(LinBatch& b, hipStream_t st) {
    if (b.nprob < 1 || b.nprob > 4 || b.K <= 0 || b.K % 32) { set_error("linear_grouped: 1..4 problems, K %% 32 == 0"); return SSS_EINVAL; }
    int total = 0;
    for (int i = 0; i < b.nprob; ++i) {
        LinProb& p = b.p[i];
        if (p.n < 0 || p.m <= 0 || p.ldw % 4 || p.ldw < b.K || p.ldy < p.m || (!p.ids && (p.ldx % 4 || p.ldx < b.K)) ||
            (p.ids && (!p.table || (p.xcopy && (p.ld_xcopy % 4 || p.ld_xcopy < b.K))))) {
            set_error("linear_grouped: problem %d has bad strides / shapes", i);
            return SSS_EINVAL;
        }
        p.tiles_m = (p.m + LT - 1) / LT;
        p.tile_begin = total;
        total += (int)((p.n + LT - 1) / LT) * p.tiles_m;
    }
    for (int i = b.nprob; i < 4; ++i) { b.p[i] = b.p[0]; b.p[i].tile_begin = 0x7fffffff; }
    if (total == 0) return SSS_OK;
    // K chunks of 32: 16 KiB of LDS per workgroup, so eight workgroups share a CU, the whole grid of a
    // query-batch launch is resident at once and the load / MFMA / store phases of different workgroups
    // overlap (these launches are latency-bound, not FLOP-bound; measured against 64- and 128-chunks)
    const int lds32 = 2 * LT * 8 * 16;
    // (chunks of 64 on deep, chip-filling problems -- corpus build, the reference's 768 / 800-wide layers -- measured the same)
    hipLaunchKernelGGL(k_linear_grouped<32>, dim3((unsigned)total), dim3(256), lds32, st, b);
    return check_launch("k_linear_grouped");
}

int layer_update(const LayerArgs& a, hipStream_t st) {
    if (a.h <= 0 || a.h % 4 || a.h > 256 || a.Np < 0 || a.Nq < 0 || a.d_x > a.h || a.ldyp % 4 || a.ldyq % 4 || a.ld_outp % 4 ||
        a.ld_outq % 4 || (a.d_x >= 4 && a.ld_xin % 4) || a.ldyp < 7 * a.h + 2 || a.ldyq < a.h + 2) {
        set_error("layer_update: need h %% 4 == 0, h <= 256, d_x <= h, 16-byte aligned row strides, ldyp >= 7h+2, ldyq >= h+2");
        return SSS_EINVAL;
    }
    if ((a.x0_p && a.ld_x0p % 4) || (a.x0_q && (!a.xq_table || a.ld_xq % 4 || a.ld_x0q % 4))) {
        set_error("layer_update: x0_p / x0_q need 16-byte aligned row strides, x0_q needs xq_table");
        return SSS_EINVAL;
    }
    const long total = a.Np + a.Nq;
    if (total == 0) return SSS_OK;
    const int lpr = lanes_for(a.h);
    const long per = 256 / lpr;
    SSS_LPR_SWITCH(lpr, hipLaunchKernelGGL(k_layer_update<L>, dim3((unsigned)((total + per - 1) / per)), dim3(256), 0, st, a));
    return check_launch("k_layer_update");
}

int pool_expand_mean(const float* lin_p, const float* lin_q, long ld_lin, const int* src_row, const int* pos_id,
                     const int* pptr, const int* qptr, long n_clicks, long n_graphs, int Dl, int P, const float* pos_emb,
                     float* node, long ld_node, float* coarse, long ld_coarse, hipStream_t st) {
    const int D = Dl + P;
    if (n_graphs < 0 || Dl <= 0 || P < 0 || D % 4 || D > 256 || ld_node % 4 || ld_node < D || ld_coarse % 4 || ld_coarse < D) {
        set_error("pool_expand_mean: need (Dl + P) %% 4 == 0, <= 256, 16-byte aligned row strides");
        return SSS_EINVAL;
    }
    if (n_graphs == 0) return SSS_OK;
    const int lpr = lanes_for(D);
    const long per = 256 / lpr;
    SSS_LPR_SWITCH(lpr, hipLaunchKernelGGL(k_pool_expand_mean<L>, dim3((unsigned)((n_graphs + per - 1) / per)), dim3(256), 0, st,
                                           lin_p, lin_q, ld_lin, src_row, pos_id, pptr, qptr, n_clicks, n_graphs, Dl, P, pos_emb,
                                           node, ld_node, coarse, ld_coarse));
    return check_launch("k_pool_expand_mean");
}

int pool_attention(const float* node, long ld_node, const float* Aa, long ld_a, const float* Bc, long ld_b, const float* watt,
                   const int* pptr, const int* qptr, long n_clicks, long n_graphs, int D, int normalize, float eps, int reduce_sum,
                   float* out, long ld_out, hipStream_t st) {
    if (n_graphs < 0 || D <= 0 || D % 4 || ld_node % 4 || ld_a % 4 || ld_b % 4 || ld_out % 4 || ld_out < D) {
        set_error("pool_attention: need D %% 4 == 0, 16-byte aligned row strides");
        return SSS_EINVAL;
    }
    if (n_graphs == 0) return SSS_OK;
    const int lpr = lanes_for(D);
    if (D > 256) {
        // rows wider than one float4 column per lane (the reference's gnn_nout = 800, config.py:16): the column-chunked
        // sweep of k_segment_pool, which computes a node's attention weight once for all of a lane's columns
        SSS_LPR_SWITCH(lpr, hipLaunchKernelGGL(k_segment_pool<L>, dim3(grid_rows(n_graphs, L)), dim3(256), 0, st, node, ld_node,
                                               pptr, qptr, n_clicks, n_graphs, D, Aa, ld_a, Bc, ld_b, watt, out, ld_out, reduce_sum));
        const int rc = check_launch("k_segment_pool");
        if (rc || !normalize) return rc;
        return normalize_rows(out, n_graphs, D, ld_out, eps, 0, st);
    }
    const long per = 256 / lpr;
    SSS_LPR_SWITCH(lpr, hipLaunchKernelGGL(k_pool_attention<L>, dim3((unsigned)((n_graphs + per - 1) / per)), dim3(256), 0, st, node,
                                           ld_node, Aa, ld_a, Bc, ld_b, watt, pptr, qptr, n_clicks, n_graphs, D, normalize, eps,
                                           reduce_sum, out, ld_out));
    return check_launch("k_pool_attention");
}

int pool_attention_tab(const float* T, long ld_t, const float* AC, long ld_ac, const float* tanhpos, const float* A2tab,
                       const float* C2tab, const float* watt, const int* src_row, const int* pos_id, const int* pptr,
                       const int* qptr, long n_clicks, long Np, long n_graphs, int Dl, int P, int normalize, float eps, float* out,
                       long ld_out, hipStream_t st) {
    const int D = Dl + P;
    if (n_graphs < 0 || Dl <= 0 || P < 0 || D % 4 || D > 256 || ld_t < Dl || ld_ac % 4 || ld_ac < 2 * D || ld_out % 4 || ld_out < D) {
        set_error("pool_attention_tab: need (Dl + P) %% 4 == 0, <= 256, ld_ac >= 2 (Dl + P), 16-byte aligned row strides");
        return SSS_EINVAL;
    }
    if (n_graphs == 0) return SSS_OK;
    const int lpr = lanes_for(D);
    const long per = 4;                                          // one wave per graph
    SSS_LPR_SWITCH(lpr, hipLaunchKernelGGL(k_pool_attention_tab<L>, dim3((unsigned)((n_graphs + per - 1) / per)), dim3(256), 0, st, T,
                                           ld_t, AC, ld_ac, tanhpos, A2tab, C2tab, watt, src_row, pos_id, pptr, qptr, n_clicks, Np,
                                           n_graphs, Dl, P, normalize, eps, out, ld_out));
    return check_launch("k_pool_attention_tab");
}

}  // namespace sss

// Session-encoder kernels (gfx950): HeteroGGNN message passing + positional-attention pooling.
//
// Reference ops replaced (op semantics Appendix A):
//   k_linear_f32        <- the dense node transforms inside PyG GATConv (lin_src), GatedGraphConv
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

namespace sss {

// ------------------------------------------------------------------------------------------
// Y[N, M] = X[N, K] * W[M, K]^T (+ bias[M]).  64 x 64 block tile, 4 waves of 32 x 32 (one MFMA
// tile each), K consumed in chunks of up to 128 staged ONCE per chunk through LDS (register
// staged, 16 float4 loads in flight per thread), 16-byte chunk XOR swizzle -> conflict-free
// ds_read_b128.  The encoder's shapes are K = 64..384 with a few thousand rows per query batch:
// per-workgroup latency, not FLOPs, bounds them, so the K loop is 1-3 steps and two workgroups
// share a CU (64 KiB LDS each).
constexpr int LT = 64;      // tile rows (X) and columns (W rows)

template <int KC>           // K chunk: 128, 64 or 32 (K % KC == 0)
__global__ __launch_bounds__(256, 2) void k_linear_f32(const float* __restrict__ X, long ldx,
                                                       const float* __restrict__ W, long ldw,
                                                       const float* __restrict__ bias, float* __restrict__ Y,
                                                       long ldy, long N, int M, int K) {
    constexpr int CPR = KC / 4;                 // 16-byte chunks per staged row
    constexpr int NLD = LT * CPR / 256;         // float4 loads per thread per operand (8, 4, 2)
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];        // [X|W][64 rows][32 chunks]
    float* lx = lds_raw;
    float* lw = lds_raw + LT * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const long row0 = (long)blockIdx.x * LT;
    const int col0 = blockIdx.y * LT;
    const int xr = wr * 32 + r, wrow = wc * 32 + r;

    f32x16 acc = {0};
    for (int k0 = 0; k0 < K; k0 += KC) {
        f32x4 sx[NLD], sw[NLD];          // ext-vector type: stays in VGPRs (HIP's float4 struct array went to scratch)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid + 256 * i;
            const int tr = p / CPR, c = p % CPR;
            long gr = row0 + tr; if (gr > N - 1) gr = N - 1;
            int gw = col0 + tr; if (gw > M - 1) gw = M - 1;
            sx[i] = *reinterpret_cast<const f32x4*>(X + gr * ldx + k0 + c * 4);
            sw[i] = *reinterpret_cast<const f32x4*>(W + (long)gw * ldw + k0 + c * 4);
        }
        if (k0 > 0) __syncthreads();                         // previous chunk fully consumed
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid + 256 * i;
            const int tr = p / CPR, c = p % CPR;
            const int cs = c ^ (tr & 15);                    // rows are 32 chunks apart: stays in the row
            *reinterpret_cast<f32x4*>(lx + (tr * 32 + cs) * 4) = sx[i];
            *reinterpret_cast<f32x4*>(lw + (tr * 32 + cs) * 4) = sw[i];
        }
        __syncthreads();
        float4 a = *reinterpret_cast<const float4*>(lx + (xr * 32 + (h ^ (xr & 15))) * 4);
        float4 b = *reinterpret_cast<const float4*>(lw + (wrow * 32 + (h ^ (wrow & 15))) * 4);
#pragma unroll
        for (int u = 0; u < KC / 8; ++u) {
            float4 na = a, nb = b;
            if (u + 1 < KC / 8) {                            // fragments one k-group ahead
                na = *reinterpret_cast<const float4*>(lx + (xr * 32 + ((2 * u + 2 + h) ^ (xr & 15))) * 4);
                nb = *reinterpret_cast<const float4*>(lw + (wrow * 32 + ((2 * u + 2 + h) ^ (wrow & 15))) * 4);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            a = na; b = nb;
        }
    }
    // C/D map of 32x32: col = lane & 31 (W row), row = (j & 3) + 8 * (j >> 2) + 4 * h (X row)
    const int col = col0 + wc * 32 + r;
    if (col < M) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const long row = row0 + wr * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
            if (row < N) Y[row * ldy + col] = acc[j] + bv;
        }
    }
}

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
    float* __restrict__ out, long ld_out) {
    const int sub = threadIdx.x % LPR;
    const long per_block = 256 / LPR;
    const int nv = h / 4;
    for (long i = (long)blockIdx.x * per_block + threadIdx.x / LPR; i < n_dst; i += (long)gridDim.x * per_block) {
        const int e0 = rowptr[i], e1 = rowptr[i + 1];
        const float ad = a_dst[i * ld_ad];
        float mx = -INFINITY;
        for (int e = e0; e < e1; ++e) {
            float v = a_src[(long)col[e] * ld_as] + ad;
            v = v > 0.f ? v : 0.2f * v;
            mx = fmaxf(mx, v);
        }
        float den = 0.f;
        for (int e = e0; e < e1; ++e) {
            float v = a_src[(long)col[e] * ld_as] + ad;
            v = v > 0.f ? v : 0.2f * v;
            den += expf(v - mx);
        }
        const float inv = 1.f / (den + 1e-16f);
        for (int c = sub; c < nv; c += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int e = e0; e < e1; ++e) {
                const long j = col[e];
                float v = a_src[j * ld_as] + ad;
                v = v > 0.f ? v : 0.2f * v;
                const float w = expf(v - mx) * inv;
                const float4 x = *reinterpret_cast<const float4*>(xs + j * ld_xs + c * 4);
                acc.x += w * x.x; acc.y += w * x.y; acc.z += w * x.z; acc.w += w * x.w;
            }
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
                                                      float* __restrict__ out, long ld_out) {
    const int sub = threadIdx.x % LPR;
    const long per_block = 256 / LPR;
    const int nv = D / 4;
    for (long g = (long)blockIdx.x * per_block + threadIdx.x / LPR; g < n_graphs; g += (long)gridDim.x * per_block) {
        const int p0 = pptr[g], p1 = pptr[g + 1], q0 = qptr[g], q1 = qptr[g + 1];
        const int cnt = (p1 - p0) + (q1 - q0);
        const float invc = 1.f / (float)(cnt > 0 ? cnt : 1);
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

int linear_f32(const float* X, long ldx, const float* W, long ldw, const float* bias, float* Y, long ldy,
               long N, int M, int K, hipStream_t st) {
    if (N < 0 || M <= 0 || K <= 0 || K % 32 || ldx % 4 || ldw % 4 || ldx < K || ldw < K || ldy < M) {
        set_error("linear: need K %% 32 == 0, ldx/ldw %% 4 == 0 and >= K, ldy >= M (N=%ld M=%d K=%d)", N, M, K);
        return SSS_EINVAL;
    }
    if (N == 0) return SSS_OK;
    dim3 grid((unsigned)((N + LT - 1) / LT), (unsigned)((M + LT - 1) / LT));
    const int lds = 2 * LT * 128 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear_f32<128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear_f32<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear_f32<32>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    if (K % 128 == 0)
        hipLaunchKernelGGL(k_linear_f32<128>, grid, dim3(256), lds, st, X, ldx, W, ldw, bias, Y, ldy, N, M, K);
    else if (K % 64 == 0)
        hipLaunchKernelGGL(k_linear_f32<64>, grid, dim3(256), lds, st, X, ldx, W, ldw, bias, Y, ldy, N, M, K);
    else
        hipLaunchKernelGGL(k_linear_f32<32>, grid, dim3(256), lds, st, X, ldx, W, ldw, bias, Y, ldy, N, M, K);
    return check_launch("k_linear_f32");
}

int gat_aggregate(const float* xs, long ld_xs, const float* a_src, long ld_as, const float* a_dst, long ld_ad,
                  const int* rowptr, const int* col, long n_dst, int h, const float* bias, int relu, float* out,
                  long ld_out, hipStream_t st) {
    if (n_dst < 0 || h <= 0 || h % 4 || ld_xs % 4 || ld_out % 4 || ld_out < h) {
        set_error("gat_aggregate: need h %% 4 == 0 and 16-byte aligned row strides");
        return SSS_EINVAL;
    }
    if (n_dst == 0) return SSS_OK;
    const int lpr = lanes_for(h);
    SSS_LPR_SWITCH(lpr, hipLaunchKernelGGL(k_gat_aggregate<L>, dim3(grid_rows(n_dst, L)), dim3(256), 0, st, xs, ld_xs,
                                           a_src, ld_as, a_dst, ld_ad, rowptr, col, n_dst, h, bias, relu, out, ld_out));
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

// Argument structures shared by the kernel translation units (gnn.hip, graphbuild.hip) and the C-ABI
// wrappers (capi.hip): ONE definition, so the two sides cannot drift apart.
#pragma once
#include "sss_common.h"

namespace sss {

struct LinProb {
    const float* x; long ldx;          // X rows (ignored when ids != nullptr)
    const long* ids; const float* table;   // gather mode: X[r] = table[ids[r]] (table row stride = K)
    float* xcopy; long ld_xcopy;       // gather mode: also written here (slice 0 of the node buffer); may be null
    const float* w; long ldw; const float* bias;
    float* y; long ldy;
    long n; int m;
    int act;                           // epilogue: 0 none, 1 relu, 2 tanh, 3 sign, 4 tanh(tanh(.))
    const float* post_scale = nullptr; // optional second epilogue stage, per output column: v = relu(v * post_scale + post_shift)
    const float* post_shift = nullptr; //   (BatchNorm1d in eval mode followed by the reference MLP's relu, model/model.py:63-65)
    int tiles_m, tile_begin;           // filled by the launcher
};
struct LinBatch { LinProb p[4]; int nprob; int K; };

struct LayerArgs {
    const float* Yp; long ldyp; const float* Yq; long ldyq; int h;
    const int* rowptr_qp; const int* col_qp; const int* rowptr_pp; const int* col_pp; const float* w_pp;
    const float* bias_qp; const float* b_ih; const float* xin_p; long ld_xin; int d_x; float* out_p; long ld_outp; long Np;
    const int* rowptr_pq; const int* col_pq; const float* bias_pq; float* out_q; long ld_outq; long Nq;
    long n_self_loop;                  // > 0: PyG bipartite self-loop rewrite on the fly, n = min(Nq, Np)
    // table mode (layer 0 over embedding-table features): Yp / Yq / xin_p are per TABLE ROW and node i uses
    // row row_p[i] / row_q[i]; the raw feature rows are copied to x0_p / x0_q (slice 0 of the node buffers)
    const long* row_p = nullptr; const long* row_q = nullptr;
    float* x0_p = nullptr; long ld_x0p = 0;
    const float* xq_table = nullptr; long ld_xq = 0; float* x0_q = nullptr; long ld_x0q = 0;
};

struct GraphOut {
    long* q_x; long* q_batch; int* q_pos;                         // [Nq]
    long* p_x; long* p_batch; long* p_cnt;                        // [Np]
    int* rowptr_qp; int* col_qp;                                  // [Np+1], [E]   targets = products, col = query node
    int* rowptr_pq; int* col_pq;                                  // [Nq+1], [E]   targets = queries,  col = product node
    int* rowptr_pp; int* col_pp; float* w_pp;                     // [Np+1], [Epp] targets = products, col = product node
    int* src_row; int* pos_id;                                    // [n_exp = Xp + Nq]
};

}  // namespace sss

/* libsss -- C ABI of the MI355X-native session-similarity hot path.
 *
 * The reference (ZongyueQin/SessionSimilaritySearch) is pure Python; the path sits behind three
 * Python call surfaces rather than an FFI:
 *   (i)   emb = data_encoder(data)                       test_amazon_filterd.py:498,553
 *         (UnifyPoolingGraphLevelEncoder.forward,        model/model.py:279-351)
 *   (ii)  build_index(emb, metric) / index.search(x, k)  test_amazon_filterd.py:207-223,578
 *         (faiss.IndexFlatIP / IndexFlatL2)
 *   (iii) normalize(vec)                                 util_amazon_filtered.py:28-31
 * The Python drop-ins in sessionsimilaritysearch_amd/ keep those names and call ONLY the entry
 * points below (ctypes).  INTEGRATION.md shows the binding a maintainer of the reference adds.
 *
 * Conventions (every entry point):
 *   - all buffers are CALLER-OWNED DEVICE pointers (tensor.data_ptr()); nothing is allocated
 *     or freed here and there is no host synchronisation: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream);
 *   - return 0 on success, -1 bad argument, -2 workspace too small, -3 HIP error;
 *     sss_last_error() returns the thread-local message of the last failure;
 *   - re-entrant per stream; no global state except the error string.
 */
#ifndef SSS_H
#define SSS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int sss_version(void);
const char* sss_last_error(void);

/* ---- (iii) normalize -- util_amazon_filtered.py:28-31 (rule 0) and fine_tune_ours.py:38-40
 * (rule 1).  In place on fp32 [n, d] rows with row stride `ld` floats.
 *   rule 0: x / sqrt(max(sum x^2, eps))      (reference eps = 1e-6)
 *   rule 1: x / (sqrt(sum x^2) + eps)        (reference eps = 1e-4)
 * If row_norm_max != NULL it receives max over rows of the POST-normalisation 2-norm
 * (one float, atomically maximised; caller zeroes it) -- used by the index for its error bound. */
int sss_normalize_rows(float* x, int64_t n, int d, int64_t ld, float eps, int rule, void* stream);

/* max over rows of ||x_i||_2 -> *out (device float, caller zeroes).  dtype 0: x is float32
 * (d % 4 == 0); dtype 1: x is bfloat16 (d % 8 == 0). */
int sss_row_norm_max(const void* x, int64_t n, int d, int dtype, float* out, void* stream);

/* float32 -> bfloat16 (round to nearest even) of `count` contiguous elements (count % 8 == 0):
 * how a corpus / query block enters the bf16 index (BASELINE config C5). */
int sss_f32_to_bf16(const float* x, int64_t count, uint16_t* y, void* stream);

/* ---- (ii) IndexFlatIP.search -- test_amazon_filterd.py:578 (also :61,661; fine_tune_ours.py:882).
 * dtype 0: q [nq, d] and corpus [n, d] float32 (row-major, as IndexFlatIP.add stored it), d in
 * {64,128,256}; dtype 1: both bfloat16, d in {128,256,512} (f32 accumulate on the bf16 MFMA).
 * k <= 500.  Writes D_out [nq, k] fp32 (descending) and I_out [nq, k] int64 = row + id_offset,
 * ordered by (score desc, id asc); missing results: I = -1, D = -FLT_MAX (faiss convention).
 * Scores are the canonical ones of DESIGN.md (float64 sequential dot of the stored elements,
 * rounded to float32).  status [nq] int32: 0 = proven exact, 1 = not proven (caller re-runs those
 * queries through sss_ip_topk_exhaustive); unproven_count (may be NULL): device int32 that is
 * incremented once per unproven query (never reset here), so a caller can run many batches
 * without a host sync and check once.  corpus_max_norm = max row 2-norm of the corpus (for
 * the error bound).  workspace: 256-byte aligned, sss_ip_topk_workspace_bytes() bytes (0 = shape
 * not supported by the fused path), contents irrelevant.  state: sss_ip_topk_state_bytes(nq) bytes,
 * 16-byte aligned, ZERO before the first call; every successful call hands it back zeroed (the
 * kernels clear the words they used), so a search is two launches and no memset.  Since the
 * whole buffer is zero between calls, one buffer serves any (nq', n, k, dtype) with nq' at most
 * the nq it was sized for; it must not be shared by searches in flight on different streams. */
size_t sss_ip_topk_state_bytes(int64_t nq);
size_t sss_ip_topk_workspace_bytes(int64_t nq, int64_t n, int d, int k, int dtype);
int sss_ip_topk(const void* q, int64_t nq, const void* corpus, int64_t n, int d, int k, int dtype,
                int64_t id_offset, float corpus_max_norm, float* D_out, int64_t* I_out,
                int32_t* status, int32_t* unproven_count, void* state, size_t state_bytes,
                void* workspace, size_t workspace_bytes, void* stream);

/* The same search for a float32 corpus with the scan on the bf16 MFMA ("split" scan, the default of
 * the Python FlatIndex): corpus_split [n, 2d] bfloat16 is the image sss_split_bf16 makes of the
 * corpus rows -- [hi(d) | lo(d)], hi = rne_bf16(x), lo = rne_bf16(x - hi), 4 bytes per element like
 * the f32 row.  The scan scores hi*hi + hi*lo + lo*hi (three bf16 MFMA passes, f32 accumulate:
 * 16/3 of the f32 MFMA rate) to pick the candidates; the candidates are re-scored from the float32
 * rows exactly as in sss_ip_topk, and the per-query proof uses the split's own error bound
 * ((3.03 * 2^-16 + 3d * 2^-23) |q| |c|), so D_out / I_out / status obey the same contract:
 * identical results for every query with status 0.  q float32 [nq, d]; d in {64,128,256};
 * workspace: sss_ip_topk_workspace_bytes(nq, n, d, k, 0); state as above. */
int sss_split_bf16(const float* x, int64_t n, int d, uint16_t* y, void* stream);
int sss_ip_topk_split(const float* q, int64_t nq, const float* corpus, const uint16_t* corpus_split,
                      int64_t n, int d, int k, int64_t id_offset, float corpus_max_norm, float* D_out,
                      int64_t* I_out, int32_t* status, int32_t* unproven_count, void* state,
                      size_t state_bytes, void* workspace, size_t workspace_bytes, void* stream);

/* The same search with the candidates found by ONE float16 MFMA pass over a scaled float16 image of
 * the float32 corpus (half the bytes, a third of the split scan's matrix work; the default of the
 * Python FlatIndex for d in {128,256,512}):
 *   corpus_f16 [n, d] float16 = round_to_nearest_even(corpus * 2^corpus_shift), made by
 *   sss_scale_f16; corpus_shift = sss_f16_shift(largest |element| of the corpus) (sss_abs_max), which
 *   puts that element in [2^12, 2^13) -- an exact scaling far from both ends of the f16 range.  A
 *   shift stays valid while every element times 2^shift is below 65504 (the index re-scales when
 *   an added row exceeds 2^15).
 * The kernel scales and rounds each float32 query the same way (its own shift), so scan scores
 * are the true scores times a per-query power of two: thresholds and candidate choice are
 * unaffected, and the proof divides it out.  Error bound of a scan score:
 * Rc |q| + (|c| + Rc) Rq + (2^-25 sqrt(d) + d 2^-23) |q| |c|, with Rq the query's own rounding
 * residual norm (measured by the kernel) and Rc = corpus_resid_norm = the largest row norm of
 * (corpus_f16 * 2^-corpus_shift - corpus), measured by sss_f16_resid_max when the image is built
 * (any upper bound is valid; the worst case is 2^-11 * corpus_max_norm).  Coarser than the other
 * scans (~4e-4 |q||c| at d = 128), so more near-ties are left unproven (status != 0) and go to
 * sss_ip_topk_exhaustive; results for status 0 are identical.
 * q float32 [nq, d]; workspace: sss_ip_topk_f16_workspace_bytes(nq, n, d, k); state as above. */
int sss_abs_max(const float* x, int64_t count, float* out, void* stream);   /* max |x_i| -> *out (device, caller zeroes); count % 4 == 0 */
int sss_f16_shift(float amax);                                              /* host helper: the shift for a largest magnitude */
int sss_scale_f16(const float* x, int64_t count, int shift, uint16_t* y, void* stream);   /* count % 8 == 0 */
int sss_f16_resid_max(const float* x, const uint16_t* y, int64_t n, int d, int shift, float* out, void* stream);   /* *out: device float, caller zeroes */
size_t sss_ip_topk_f16_workspace_bytes(int64_t nq, int64_t n, int d, int k);
int sss_ip_topk_f16(const float* q, int64_t nq, const float* corpus, const uint16_t* corpus_f16,
                    int corpus_shift, float corpus_resid_norm, int64_t n, int d, int k, int64_t id_offset, float corpus_max_norm,
                    float* D_out, int64_t* I_out, int32_t* status, int32_t* unproven_count, void* state,
                    size_t state_bytes, void* workspace, size_t workspace_bytes, void* stream);

/* Threshold rung: resolves queries a fused search (sss_ip_topk / _split / _f16) left with status != 0 at
 * matrix-core speed, before the exhaustive kernels are needed.  qsel [nsel] int32 = those query rows.  Row
 * qsel[i] of D_out still holds the fused search's k-th re-scored candidate in column k-1 -- a valid LOWER BOUND
 * of the true k-th score; one more scan of `scan_image` for just these queries keeps EVERY row whose scan score
 * lies above (bound - scan error bound - one float32 ulp) and re-scores them all canonically (float64, from
 * `corpus`): exact for near ties and for exact ties (duplicate rows) alike.  Resolved queries get their rows of
 * D_out / I_out rewritten and status 0; a query with more than 8192 such rows (or NaNs) keeps its status and goes
 * to sss_ip_topk_exhaustive.  scan: 0 / 1 = the corpus itself (f32 / bf16 index: pass scan_image = corpus),
 * 2 = the [hi | lo] bf16 image (sss_split_bf16), 3 = the scaled float16 image (corpus_shift / corpus_resid_norm
 * as for sss_ip_topk_f16; ignored otherwise).  Serves the same result contract as the reference's
 * `index.search` (test_amazon_filterd.py:578).  workspace (256-byte aligned):
 * sss_ip_topk_threshold_workspace_bytes(nsel, n, d, scan). */
size_t sss_ip_topk_threshold_workspace_bytes(int64_t nsel, int64_t n, int d, int scan);
int sss_ip_topk_threshold(const void* q, const int32_t* qsel, int64_t nsel, const void* corpus, int dtype,
                          const void* scan_image, int scan, int corpus_shift, float corpus_resid_norm, int64_t n,
                          int d, int k, int64_t id_offset, float corpus_max_norm, float* D_out, int64_t* I_out,
                          int32_t* status, void* workspace, size_t workspace_bytes, void* stream);

/* LONG rows (the reference's own D = 1600 session vectors, K = 100: pretrain_filtered_amazon.py:281,
 * test_amazon_filterd.py:459,578 -- `index.search(normalize(emb), K)`): any d % 64 == 0 with rows of at most
 * 16384 bytes (d <= 4096 float32, <= 8192 bfloat16), k <= 1024 (what the exhaustive path of a status-1 query resolves).
 * A K-tiled MFMA contraction (256 queries x 256 rows per workgroup tile, both operands streamed through LDS in
 * 128-byte slabs) whose top-k rides on thresholds instead of running lists: a few evenly spread row samples of
 * growing size give, level by level, a tighter lower bound of each query's k-th score; the last pass over every
 * row keeps exactly the rows that can still reach that bound and re-scores them all canonically (float64, from
 * `corpus`).  status 0 = exact; 1 = more than 8192 rows (4096 for rows beyond 10240 bytes) could reach the bound (mass ties): re-run through
 * sss_ip_topk_exhaustive(_lb) -- column k-1 of such a row of D_out still holds a valid lower bound (or -FLT_MAX).
 * dtype 0: q / corpus float32, scan_image = the scaled float16 image of the corpus (sss_scale_f16; corpus_shift /
 * corpus_resid_norm as for sss_ip_topk_f16; the queries are scaled + rounded to float16 internally).
 * dtype 1: q / corpus bfloat16, scan_image = corpus (shift / residual ignored).
 * workspace (256-byte aligned): sss_ip_topk_long_workspace_bytes(nq, n, d, dtype). */
size_t sss_ip_topk_long_workspace_bytes(int64_t nq, int64_t n, int d, int dtype);
int sss_ip_topk_long(const void* q, int64_t nq, const void* corpus, int dtype, const void* scan_image,
                     int corpus_shift, float corpus_resid_norm, int64_t n, int d, int k, int64_t id_offset,
                     float corpus_max_norm, float* D_out, int64_t* I_out, int32_t* status, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Exhaustive exact search for a (small) set of queries: qsel [nsel] int32 are the query rows of
 * q to process; results are written to rows qsel[i] of D_out / I_out.  Any n, any d % 4 == 0
 * (dtype 0) or d % 8 == 0 (dtype 1), k <= 1024.  metric: 0 = inner product, 1 = squared L2
 * (IndexFlatL2, test_amazon_filterd.py:215-217; D ascending).
 * workspace: sss_ip_topk_exhaustive_workspace_bytes(nsel, n). */
size_t sss_ip_topk_exhaustive_workspace_bytes(int64_t nsel, int64_t n);
int sss_ip_topk_exhaustive(const void* q, const int32_t* qsel, int64_t nsel, const void* corpus,
                           int64_t n, int d, int k, int dtype, int64_t id_offset, int metric,
                           float* D_out, int64_t* I_out, void* workspace, size_t workspace_bytes,
                           void* stream);

/* The same (inner-product metric) with a per-query LOWER bound of the k-th best score, lower_bound [nsel]
 * float32 -- e.g. D_out[q][k-1] of a fused search that returned status != 0: its k-th re-scored candidate
 * is a real row's canonical score.  A float32 pre-test then skips the float64 chain for every row that
 * provably scores below the bound (d in {64,128,256} float32 / {128,256} bfloat16; other shapes ignore it).
 * Results are identical to sss_ip_topk_exhaustive; pass -FLT_MAX where no bound is known. */
int sss_ip_topk_exhaustive_lb(const void* q, const int32_t* qsel, int64_t nsel, const void* corpus,
                              int64_t n, int d, int k, int dtype, int64_t id_offset,
                              const float* lower_bound, float* D_out, int64_t* I_out, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- multi-GPU: merge per-shard results after the all-gather (no reference equivalent; the
 * reference is single process).  Shard s's [nq, k] block starts at D_in + s * d_shard_stride
 * (floats) / I_in + s * i_shard_stride (int64s); output [nq, k] by (score desc, id asc). */
int sss_topk_merge(const float* D_in, int64_t d_shard_stride, const int64_t* I_in, int64_t i_shard_stride,
                   int shards, int64_t nq, int k, float* D_out, int64_t* I_out, void* stream);

/* ---- measurement aid (bench.py roofline leg): when enabled, every launch of the dominant
 * scoring kernel inside sss_ip_topk is bracketed by a hipEvent pair on its own stream;
 * sss_profile_read synchronises them and returns the summed duration and the launch count
 * since the last read (state is per device: the calling thread's current device; at most 512
 * launches between reads). */
int sss_profile_enable(int on);
/* Diagnostic counter of the fused scans (csrc/scan.hip): the number of waves, since the last reset, whose bounded wait
 * for the shared admission threshold at the end of the first warm-up tile ran out -- the workgroups of a launch were not
 * co-resident (another stream's kernels held CUs).  Results stay exact (the affected queries lose their proof and go
 * through the threshold rung), but the search is several times slower: a value other than 0 explains such a cliff.
 * Synchronises the current device; reset != 0 zeroes the counter.  Returns the count, or a negative code. */
int sss_scan_boot_expired(int reset);
int sss_profile_read(double* total_ms, int* launches);

/* ---- (i) encoder pieces.  NodeAsinEmbedding.forward -- model/NodeEmbedding.py:137-138:
 * out[i, :] = table[ids[i], :]; out row stride ld_out floats (writes slice 0 of the node buffer). */
int sss_gather_rows(const float* table, const int64_t* ids, int64_t n, int d, float* out,
                    int64_t ld_out, void* stream);

/* use_id_embedding=True of UnifyPoolingGraphLevelEncoder.forward -- model/model.py:288-289:
 * embedding['product'] = torch.concat((a, b), dim=1), a = the id embedding rows (NodeAsinEmbedding), b = the node's
 * text features: out[i] = [table[ids[i]] (d_id floats) | feat[i] (d_feat floats) | d_pad zeros], one pass.
 * feat == NULL writes zeros in its place (d_id == 0 with feat == NULL: a zero-padded row); every width % 4 == 0. */
int sss_gather_concat_rows(const float* table, const int64_t* ids, int d_id, const float* feat, int64_t ld_feat,
                           int d_feat, int d_pad, int64_t n, float* out, int64_t ld_out, void* stream);

/* Dense node transform  y[n, m] = x[n, k] * w[m, k]^T (+ bias[m])  on the f32 MFMA.  Replaces
 * the nn.Linear / lazy Linear / GRUCell matmuls inside PyG GATConv (lin_src), GatedGraphConv
 * (x @ weight; pass weight transposed) and PositionalAttentionPooling (reference
 * model/gnn.py:54,58,186-190).  k % 32 == 0; ldx, ldw multiples of 4 floats. */
int sss_linear(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y,
               int64_t ldy, int64_t n, int m, int k, void* stream);

/* GATConv.propagate for one edge type (instantiated model/gnn.py:54):
 * CSR by target (rowptr [n_dst+1], col = source ids, int32): e = leaky_relu(a_src[j] + a_dst[i], 0.2),
 * per-target softmax with the +1e-16 of PyG, out[i] = sum_j w_ij xs[j] + bias (relu != 0: then
 * max(.,0)).  a_src / a_dst are strided scalars (element i at a[i * ld]).
 * n_self_loop > 0 applies PyG's GATConv(add_self_loops=True) edge rewrite on the fly, in the
 * batch-global indices of the CSR: edges with source == target are dropped and one edge i -> i is
 * appended for every target i < n_self_loop (= min(n_src, n_dst)); 0: the CSR is used as is. */
int sss_gat_aggregate(const float* xs, int64_t ld_xs, const float* a_src, int64_t ld_as,
                      const float* a_dst, int64_t ld_ad, const int32_t* rowptr, const int32_t* col,
                      int64_t n_dst, int h, const float* bias, int relu, int64_t n_self_loop, float* out,
                      int64_t ld_out, void* stream);

/* GatedGraphConv.propagate, aggr='add' (Appendix A.3; model/gnn.py:58):
 * out[i] = sum_{e in row i} (w[e] if w else 1) * m[col[e]]. */
int sss_csr_weighted_sum(const float* m, int64_t ld_m, const int32_t* rowptr, const int32_t* col,
                         const float* w, int64_t n_dst, int h, float* out, int64_t ld_out, void* stream);

/* torch.nn.GRUCell gate math (inside GatedGraphConv) + HeteroConv(aggr='sum') + relu
 * (model/gnn.py:59,72): gi = W_ih m + b_ih [n,3h], gh = W_hh x + b_hh [n,3h], x zero-padded from
 * d_x to h columns; out = relu(add + (1-z) n + z x).  add may be NULL. */
int sss_gru_combine(const float* gi, int64_t ld_gi, const float* gh, int64_t ld_gh, const float* x,
                    int64_t ld_x, int d_x, const float* add, int64_t ld_add, int64_t n, int h, float* out,
                    int64_t ld_out, void* stream);

/* PositionalAttentionPooling.forward pieces (model/gnn.py:193-217).
 * expand: node[e, :d_lin] = tanh(lin[src_row[e]]), node[e, d_lin:] = tanh(pos_emb[pos_id[e]]);
 *         rows e < n_clicks read lin_p (product clicks, repeat_interleave by cnt), the rest lin_q.
 * segment_pool: per graph g over expanded rows [pptr[g],pptr[g+1]) and n_clicks+[qptr[g],qptr[g+1]):
 *         watt == NULL: mean (global_mean_pool); else mean(node * att), att = watt . sigmoid(a + bcoarse[g]).
 * segment_ptr: ptr[g] = lower_bound(batch, g) for g in [0, n_graphs] (batch sorted, int64). */
int sss_pool_expand(const float* lin_p, const float* lin_q, int64_t ld_lin, const int32_t* src_row,
                    const int32_t* pos_id, int64_t n_clicks, int64_t n_exp, int d_lin, int p,
                    const float* pos_emb, float* node, int64_t ld_node, void* stream);
int sss_segment_pool(const float* node, int64_t ld_node, const int32_t* pptr, const int32_t* qptr,
                     int64_t n_clicks, int64_t n_graphs, int d, const float* a, int64_t ld_a,
                     const float* bcoarse, int64_t ld_b, const float* watt, float* out, int64_t ld_out,
                     void* stream);
int sss_segment_ptr(const int64_t* batch, int64_t n, int64_t n_graphs, int32_t* ptr, void* stream);

/* ---- (i) fused encoder path: 8 launches per forward (DESIGN.md "encoder").
 *
 * sss_linear_grouped: up to 4 node-linear problems  y = x w^T (+ bias)  in ONE launch (the product
 * and query transforms of a HeteroGGNN layer -- model/gnn.py:54,58 --, the pooling's query_lin +
 * product_lin, or node_emb_lin + coarse_rep_lin -- model/gnn.py:186-190).  k % 32 == 0, shared
 * by all problems.  With ids != NULL the X rows are table[ids[r]] (NodeAsinEmbedding.forward,
 * model/NodeEmbedding.py:137-138, fused into the transform; table row stride = k) and, when
 * xcopy != NULL, are also written to xcopy[r * ld_xcopy] (slice 0 of the node buffer). */
typedef struct {
    const float* x; int64_t ldx;
    const int64_t* ids; const float* table; float* xcopy; int64_t ld_xcopy;
    const float* w; int64_t ldw; const float* bias;
    float* y; int64_t ldy;
    int64_t n; int32_t m; int32_t act;      /* epilogue activation: 0 none, 1 relu, 2 tanh, 3 sign, 4 tanh(tanh(.)) */
    const float* post_scale;                /* optional [m] (both or neither): after the activation, per output   */
    const float* post_shift;                /* column v = relu(v * post_scale + post_shift) -- BatchNorm1d in eval */
                                            /* mode + the relu the reference MLP applies to it (model/model.py:63-65) */
} sss_linear_problem;
int sss_linear_grouped(const sss_linear_problem* problems, int n_problems, int k, void* stream);

/* sss_hetero_layer_update: everything of one HeteroGGNN layer after the node transforms
 * (model/gnn.py:67-73): per product node GATConv(query->product) aggregate + GatedGraphConv
 * aggregate + GRUCell gates + HeteroConv sum + relu; per query node GATConv(product->query) +
 * relu.  Column layout of the transforms (produced by sss_linear_grouped, weights fused by the
 * caller):  yp [np, >= 7h+2]: xs_p | u_r u_z u_n | gh_r gh_z gh_n | alpha_src(pq) alpha_dst(qp)
 *           yq [nq, >=  h+2]: xs_q | alpha_src(qp) alpha_dst(pq)
 * where u = x (W_ggc W_ih^T) -- GRU input transform applied before the (linear) neighbour sum --
 * and gh = W_hh x + b_hh.  CSR by target, int32; n_self_loop as in sss_gat_aggregate (both GAT
 * directions).  h <= 256.
 * Table mode (layer 0 of an encoder whose node features are rows of an embedding table): the
 * transforms of a node depend only on its table row, so the caller transforms the TABLES once
 * (yp = item_table W^T [n_items, ..], yq = query_table W^T) and passes row_p / row_q (int64 table row of
 * every node; NULL = node i uses row i): every read of yp / yq / xin_p goes through them and the layer
 * needs no per-batch transform launch.  x0_p / x0_q (may be NULL): the node's raw feature row (d_x
 * floats; xin_p resp. xq_table) is also written there -- slice 0 of the node buffers. */
typedef struct {
    const float* yp; int64_t ld_yp; const float* yq; int64_t ld_yq; int32_t h; int32_t d_x;
    const int32_t* rowptr_qp; const int32_t* col_qp; const int32_t* rowptr_pp; const int32_t* col_pp;
    const float* w_pp; const float* bias_qp; const float* b_ih;
    const float* xin_p; int64_t ld_xin; float* out_p; int64_t ld_out_p; int64_t np;
    const int32_t* rowptr_pq; const int32_t* col_pq; const float* bias_pq;
    float* out_q; int64_t ld_out_q; int64_t nq;
    int64_t n_self_loop;
    const int64_t* row_p; const int64_t* row_q;                 /* table mode (NULL: identity) */
    float* x0_p; int64_t ld_x0_p;                               /* optional copy of the product feature rows */
    const float* xq_table; int64_t ld_xq; float* x0_q; int64_t ld_x0_q;   /* optional copy of the query feature rows */
} sss_layer_args;
int sss_hetero_layer_update(const sss_layer_args* args, void* stream);

/* PositionalAttentionPooling.forward in two kernels around one sss_linear_grouped call
 * (model/gnn.py:193-217): expand_mean writes node[e] = tanh([lin[src_row[e]] ; pos_emb[pos_id[e]]])
 * and coarse[g] = mean over the graph's expanded rows; attention computes
 * out[g] = mean_e(node[e] * (watt . sigmoid(a[e] + b[g]))) and, if normalize != 0, applies the
 * reference normalize (util_amazon_filtered.py:28-31) to the row; reduce_sum != 0 sums instead of
 * averaging (the global_add_pool of SRGNN_Pooling, model/gnn.py:178).  d = d_lin + p <= 256. */
int sss_pool_expand_mean(const float* lin_p, const float* lin_q, int64_t ld_lin, const int32_t* src_row,
                         const int32_t* pos_id, const int32_t* pptr, const int32_t* qptr, int64_t n_clicks,
                         int64_t n_graphs, int d_lin, int p, const float* pos_emb, float* node,
                         int64_t ld_node, float* coarse, int64_t ld_coarse, void* stream);
int sss_pool_attention(const float* node, int64_t ld_node, const float* a, int64_t ld_a, const float* b,
                       int64_t ld_b, const float* watt, const int32_t* pptr, const int32_t* qptr,
                       int64_t n_clicks, int64_t n_graphs, int d, int normalize, float eps, int reduce_sum,
                       float* out, int64_t ld_out, void* stream);

/* The same pooling without materialising the expanded rows (what the fused encoder uses: 6-7 launches
 * per forward).  A linear map of an expanded row [tanh(lin[src]) ; tanh(pos_emb[pid])] is a per-node
 * part plus a per-position table entry, so the caller prepares (weights only) tanhpos = tanh(pos_emb)
 * [p, p], a2tab = tanhpos Wn[:, d_lin:]^T + bn [p, d], c2tab = tanhpos Wc[:, d_lin:]^T [p, d], and per
 * batch t = tanh(lin) [np + nq, >= d_lin] (products first, then queries; sss_linear_grouped with the
 * tanh epilogue) and ac = t [Wn[:, :d_lin] ; Wc[:, :d_lin]]^T [np + nq, 2 d].  Then per graph:
 * out = mean_e(row_e * (watt . sigmoid(A1[src] + a2tab[pid] + mean_e'(C1[src'] + c2tab[pid'])))). */
int sss_pool_attention_tab(const float* t, int64_t ld_t, const float* ac, int64_t ld_ac, const float* tanhpos,
                           const float* a2tab, const float* c2tab, const float* watt, const int32_t* src_row,
                           const int32_t* pos_id, const int32_t* pptr, const int32_t* qptr, int64_t n_clicks,
                           int64_t np, int64_t n_graphs, int d_lin, int p, int normalize, float eps,
                           float* out, int64_t ld_out, void* stream);

/* ---- other conv / pool variants of the reference on the same CSR / segment layout (d <= 256):
 * csr_mean: out[i] = mean over the incoming edges of target i of x[col[e]] (0 without edges) -- the
 *   aggregation of PyG SAGEConv (model/gnn.py:89-121).
 * segment_reduce: out[g] = mean (mode 0) / sum (1) / max (2) of rows [ptr[g], ptr[g+1]) of x, each
 *   row first scaled by w[r] when w != NULL -- GraphPooling (model/gnn.py:123-143) and the
 *   last_click_mask sum of SRGNN_Pooling (model/gnn.py:172).
 * attention_dot_pool: out[g] = mean_i(x_i <x_i, mean_g>) -- AttentionPooling (model/gnn.py:145-161). */
int sss_csr_mean(const float* x, int64_t ld_x, const int32_t* rowptr, const int32_t* col, int64_t n_dst, int d,
                 float* out, int64_t ld_out, void* stream);
int sss_segment_reduce(const float* x, int64_t ld_x, const float* w, const int32_t* ptr, int64_t n_graphs, int d,
                       int mode, float* out, int64_t ld_out, void* stream);
int sss_attention_dot_pool(const float* x, int64_t ld_x, const int32_t* ptr, int64_t n_graphs, int d, float* out,
                           int64_t ld_out, void* stream);

/* ---- action table -> batched session graphs, on device: the structural part of
 * sequence_to_graph (util_amazon_filtered.py:98-230) + Batch.from_data_list
 * (test_amazon_filterd.py:485-488), straight into the CSR-by-target form the encoder kernels read.
 * Input: sessions stored contiguously, sess_ptr int64 [S+1]; per action is_search uint8, item_id
 * int64 (clicks), query_tok int64 (searches); at most 64 actions per session (*err != 0 otherwise).
 * Step 1, sss_graph_counts: bases int32 [5][S+1] = exclusive scans over sessions of (query nodes,
 *   product nodes, expanded product rows, click edges, unique transitions), grand totals at [.][S]
 *   -- the caller reads the 5 totals to size the outputs; rows 2 and 0 double as the pooling's
 *   per-graph pointers pptr / qptr, row 1 as the graph -> product-node pointer.
 *   scratch: sss_graph_scratch_ints(S) int32.
 * Step 2, sss_graph_fill writes every array of the sss_graph_out struct; sizes from the totals: Nq, Np, Xp =
 *   expanded product rows, E = clicks, Epp.  It holds node features ids / batch vectors / click counts, the
 *   three CSRs (qp: targets products, pq: targets queries, pp with count weights), and the
 *   pooling's src_row / pos_id [Xp + Nq] (expanded product rows first, then the query nodes). */
typedef struct {
    int64_t* q_x; int64_t* q_batch; int32_t* q_pos;
    int64_t* p_x; int64_t* p_batch; int64_t* p_cnt;
    int32_t* rowptr_qp; int32_t* col_qp;
    int32_t* rowptr_pq; int32_t* col_pq;
    int32_t* rowptr_pp; int32_t* col_pp; float* w_pp;
    int32_t* src_row; int32_t* pos_id;
} sss_graph_out;
size_t sss_graph_scratch_ints(int64_t n_sessions);
int sss_graph_counts(const int64_t* sess_ptr, const uint8_t* is_search, const int64_t* item_id, int64_t n_sessions,
                     int32_t* bases, int32_t* scratch, int32_t* err, void* stream);
int sss_graph_fill(const int64_t* sess_ptr, const uint8_t* is_search, const int64_t* item_id,
                   const int64_t* query_tok, int64_t n_sessions, const int32_t* bases,
                   const sss_graph_out* out, void* stream);

/* ---- binary-code (Hamming) index: the reference's compressed variant -- sign bits of the
 * BinarizeHead output (model/model.py:105-138), np.packbits((emb + 1) / 2),
 * faiss.IndexBinaryFlat(nbits).add / .search (fine_tune_ours.py:839-843,871-876).
 * sss_pack_sign_bits: x [n, c] fp32 (row stride ldx) -> out [n, nbytes] uint8, bit = ((int)((x+1)/2) != 0),
 *   first column in the most significant bit of byte 0, zero padded (numpy packbits).
 * sss_hamming_topk: q [nq, nbytes], codes [n, nbytes] uint8, nbytes in {16, 32, 64}; D_out [nq, k]
 *   int32 Hamming distances ascending, I_out [nq, k] int64 ids ordered by (distance asc, id asc),
 *   -1 / INT_MAX padded; status 0 = proven exact, 1 = re-run through the exhaustive entry point.
 *   k <= sss_hamming_topk_capacity(nq, n) (16 * splits, typically 1024); returns -1 for larger k (use
 *   the exhaustive entry point). */
int sss_pack_sign_bits(const float* x, int64_t n, int c, int64_t ldx, uint8_t* out, int nbytes, void* stream);
size_t sss_hamming_topk_workspace_bytes(int64_t nq, int64_t n);
int sss_hamming_topk_capacity(int64_t nq, int64_t n);
int sss_hamming_topk(const uint8_t* q, int64_t nq, const uint8_t* codes, int64_t n, int nbytes, int k,
                     int64_t id_offset, int32_t* D_out, int64_t* I_out, int32_t* status, void* workspace,
                     size_t workspace_bytes, void* stream);
size_t sss_hamming_topk_exhaustive_workspace_bytes(int64_t nsel, int64_t n);
int sss_hamming_topk_exhaustive(const uint8_t* q, const int32_t* qsel, int64_t nsel, const uint8_t* codes,
                                int64_t n, int nbytes, int k, int64_t id_offset, int32_t* D_out,
                                int64_t* I_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- neighbour-weighted item vote: get_prediction_by_knn after the search
 * (test_amazon_filterd.py:59-78; config C3's "aggregated top-10").  D [nq, s] fp32 and I [nq, s]
 * int64 are a search result (I < 0 = padding, skipped); session i's distinct items are
 * items[items_ptr[i - id_offset] .. items_ptr[i - id_offset + 1]) (the product.x of graph i).
 * Every item of neighbour j gets weight D[j]; weights are summed per item in float64 in neighbour
 * order; out_items [nq, k] int64 = the k heaviest items by (weight desc, first-seen order asc),
 * -1 padded; out_weights [nq, k] float64 may be NULL.  status [nq]: 0 ok, 1 = more than 16384
 * (neighbour, item) pairs for that query (not processed).  s < 32768. */
int sss_knn_item_vote(const float* D, const int64_t* I, int64_t nq, int s, const int64_t* items_ptr,
                      const int32_t* items, int64_t id_offset, int64_t n_sessions, int k,
                      int64_t* out_items, double* out_weights, int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif

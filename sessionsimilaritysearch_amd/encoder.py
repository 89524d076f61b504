"""Session encoder: drop-in for the reference's ``UnifyPoolingGraphLevelEncoder.forward``
(``model/model.py:279-351``): node features -> ``HeteroGGNN`` (``model/gnn.py:43-81``) ->
``PositionalAttentionPooling`` (``model/gnn.py:183-217``) -> one vector per session.

Every arithmetic step runs in the HIP kernels of ``libsss.so`` through the C ABI
(``include/sss.h``); torch owns device memory, the stream and the integer index preparation
(CSR by target, repeat_interleave).  There is no CPU fallback.

Out of scope, by design: the reference's BERT/ELECTRA text encoder
(``model/NodeEmbedding.py:100-125``).  Node input features are rows of two tables --
``item_table`` (``NodeAsinEmbedding``, ``model/NodeEmbedding.py:128-138``) and ``query_table``
(stand-in for the text features of query nodes) -- or precomputed float features passed as
``data[node_type].feat``.
"""
from __future__ import annotations

import ctypes
import itertools
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .sessions import ASIN_NUM, EDGE_PP, EDGE_PQ, EDGE_QP, MAX_SEQ_LEN, QUERY_VOCAB, ActionTable

ALPHA_PAD = 32          # extra output columns of the fused node transforms (2 used: alpha_src, alpha_dst)


@dataclass
class EncoderConfig:
    d_in: int = 128                 # node input feature width (query nodes; product nodes too unless d_id > 0)
    # use_id_embedding=True of the reference (model/model.py:264,288-289): embedding['product'] =
    # concat(id_embedding(x), text_features) -- d_id is the width of the id embedding (``item_table``), the text
    # features (d_in wide) come from ``data['product'].feat`` or the ``item_text_table`` stand-in.  0 = False, what
    # the deployed checkpoints use (pretrain_filtered_amazon.py:277,287).
    d_id: int = 0
    h: int = 128                    # GNN width (reference CFG.gnn_nout = 800, config.py:16)
    n_layers: int = 2               # reference CFG.gnn_nlayers = 3, config.py:21
    d_out: int = 128                # session vector width D (reference gnn_nout * 2 = 1600)
    max_seq_len: int = MAX_SEQ_LEN  # positional table is [P, P] (model/gnn.py:188)
    n_items: int = ASIN_NUM
    n_query: int = QUERY_VOCAB
    # "pyg_bipartite_global": PyG GATConv(add_self_loops=True) edge rewrite in batch-global
    # indices (SURVEY.md hard part H3) -- the parity default.  "none": independent graphs.
    self_loop_rule: str = "pyg_bipartite_global"

    @property
    def d_p(self) -> int:               # product-node input width
        return self.d_id + self.d_in

    @property
    def node_width(self) -> int:        # W of the product node outputs (and of the query ones when d_id == 0)
        return self.d_p + self.n_layers * self.h

    @property
    def node_width_q(self) -> int:
        return self.d_in + self.n_layers * self.h

    def validate(self):
        if self.d_in % 32 or self.h % 32 or self.d_out % 32 or self.d_id % 32 or self.d_id < 0:
            raise ValueError("d_in, d_id, h and d_out must be multiples of 32")
        if self.d_p > self.h:
            raise ValueError("GatedGraphConv needs d_in <= h (the reference hits the same ValueError)")
        if self.d_out <= self.max_seq_len:
            raise ValueError("d_out must exceed max_seq_len")
        if self.self_loop_rule not in ("pyg_bipartite_global", "none"):
            raise ValueError("unknown self_loop_rule")


def _uniform(gen, shape, bound):
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def init_weights(cfg: EncoderConfig, seed: int, random_bias: bool = True, tables: bool = True):
    """Random-init weights of the reference architecture: embeddings
    N(0,1); nn.Linear default U(+-1/sqrt(fan_in)); GAT lin/att Glorot; GGC weight and GRUCell
    U(+-1/sqrt(h)).  ``random_bias`` draws the GAT biases too (PyG's default is zeros)."""
    cfg.validate()
    g = torch.Generator().manual_seed(seed)
    h, D, P, W = cfg.h, cfg.d_out, cfg.max_seq_len, cfg.node_width
    w = {}
    if tables:
        w["item_table"] = torch.randn((cfg.n_items, cfg.d_id if cfg.d_id else cfg.d_in), generator=g)
        w["query_table"] = torch.randn((cfg.n_query, cfg.d_in), generator=g)
        if cfg.d_id:
            w["item_text_table"] = torch.randn((cfg.n_items, cfg.d_in), generator=g)
    for l in range(cfg.n_layers):
        din = cfg.d_in if l == 0 else h
        for name in ("gat_qp", "gat_pq"):
            # lazy (-1, -1) input widths: the source / target node types differ in width when d_id > 0
            d_src = din if (l > 0 or name == "gat_qp") else cfg.d_p
            d_dst = din if (l > 0 or name == "gat_pq") else cfg.d_p
            w[f"{name}.{l}.lin_src"] = _uniform(g, (h, d_src), math.sqrt(6.0 / (d_src + h)))
            w[f"{name}.{l}.lin_dst"] = _uniform(g, (h, d_dst), math.sqrt(6.0 / (d_dst + h)))
            ga = math.sqrt(6.0 / (1 + h))
            w[f"{name}.{l}.att_src"] = _uniform(g, (h,), ga)
            w[f"{name}.{l}.att_dst"] = _uniform(g, (h,), ga)
            w[f"{name}.{l}.bias"] = _uniform(g, (h,), 0.1) if random_bias else torch.zeros(h)
        b = 1.0 / math.sqrt(h)
        w[f"ggc.{l}.weight"] = _uniform(g, (h, h), b)
        w[f"ggc.{l}.w_ih"] = _uniform(g, (3 * h, h), b)
        w[f"ggc.{l}.w_hh"] = _uniform(g, (3 * h, h), b)
        w[f"ggc.{l}.b_ih"] = _uniform(g, (3 * h,), b)
        w[f"ggc.{l}.b_hh"] = _uniform(g, (3 * h,), b)
    for name, wd in (("pool.query_lin", cfg.node_width_q), ("pool.product_lin", W)):
        bw = 1.0 / math.sqrt(wd)
        w[name + ".w"] = _uniform(g, (D - P, wd), bw)
        w[name + ".b"] = _uniform(g, (D - P,), bw)
    w["pool.pos_emb"] = torch.randn((P, P), generator=g)
    bd = 1.0 / math.sqrt(D)
    w["pool.node_lin.w"] = _uniform(g, (D, D), bd)
    w["pool.node_lin.b"] = _uniform(g, (D,), bd)
    w["pool.coarse_lin.w"] = _uniform(g, (D, D), bd)
    w["pool.att_lin.w"] = _uniform(g, (D,), bd)
    return {k: v.float().contiguous() for k, v in w.items()}


def save_weights(path, weights):
    """Flat weight file: named float32 arrays (``.npz``).  The reference pickles whole
    ``nn.Module`` tuples (pretrain_filtered_amazon.py:605-610), which cannot be loaded without
    torch_geometric; this is the build's replacement."""
    np.savez(path, **{k: v.detach().cpu().numpy() for k, v in weights.items()})


def load_weights(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def build_csr(edge_index: torch.Tensor, n_dst: int, n_src: int = 0, self_loops: bool = False,
              edge_weight=None):
    """COO [2, E] (row 0 = source j, row 1 = target i) -> CSR by target (rowptr int32 [n_dst+1],
    col int32 [E']) keeping the original edge order inside every target's segment.  With
    ``self_loops`` the PyG GATConv rewrite is applied first (Appendix A.2): drop edges whose
    source index equals the target index, append (i -> i) for i < min(n_src, n_dst)."""
    src, dst = edge_index[0], edge_index[1]
    w = edge_weight
    if self_loops:
        keep = src != dst
        loop = torch.arange(min(n_src, n_dst), dtype=src.dtype, device=src.device)
        src = torch.cat([src[keep], loop])
        dst = torch.cat([dst[keep], loop])
        if w is not None:
            w = torch.cat([w[keep], torch.ones(loop.numel(), dtype=w.dtype, device=w.device)])
    order = torch.argsort(dst, stable=True)
    col = src[order].to(torch.int32).contiguous()
    counts = torch.bincount(dst, minlength=n_dst)
    rowptr = torch.zeros(n_dst + 1, dtype=torch.int32, device=dst.device)
    rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    wv = None if w is None else w[order].float().contiguous()
    return rowptr, col, wv


class PreparedBatch:
    """Device-resident batch: feature ids + CSR adjacencies + pooling index arrays
    (``SessionEncoder.prepare``)."""


class SessionEncoder:
    """``model.forward(data)`` drop-in (``UnifyPoolingGraphLevelEncoder``).  ``data`` is a
    ``SessionBatch`` (or any object with the same attributes, e.g. a PyG hetero batch whose
    node stores carry ``x`` ids) on the encoder's device."""

    _serials = itertools.count()

    def __init__(self, cfg: EncoderConfig, weights: dict, device=None, use_edge_weight: bool = False,
                 debug_nan_checks: bool = False, fused: bool = True):
        cfg.validate()
        self._serial = next(SessionEncoder._serials)     # keys the per-batch workspace / argument-block cache
        if not torch.cuda.is_available():
            raise _lib.SssError("no HIP device available: the encoder runs on MI355X only")
        _lib.lib()
        # use_id_embedding (d_id > 0): internally BOTH node types run at the product width d_p = d_id + d_in -- query
        # feature rows are zero-padded behind their d_in columns and every weight that reads them gets zero columns
        # there (`_internal_weights`), which leaves the arithmetic of the reference's mixed-width model unchanged
        # (a zero column adds exact zeros) and lets every kernel keep ONE input width.
        self.user_cfg = cfg
        self.d_id, self.d_feat = cfg.d_id, cfg.d_in
        if cfg.d_id:
            weights = self._internal_weights(cfg, weights)
            cfg = EncoderConfig(**{**cfg.__dict__, "d_in": cfg.d_p, "d_id": 0})
        self.cfg = cfg
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.use_edge_weight = use_edge_weight      # the deployed path passes none (model/model.py:317)
        self.debug_nan_checks = debug_nan_checks    # the reference's 3 host-syncing asserts
        self.fused = fused                          # 7-launch fused kernels where the shapes allow (fused_ok)
        self.training = False
        self._prepare(weights)

    # -- nn.Module-flavoured conveniences the reference scripts call
    def eval(self):
        self.training = False
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise _lib.SssError("SessionEncoder is bound to its construction device")
        return self

    def __call__(self, *a, **kw):
        return self.forward(*a, **kw)

    # ------------------------------------------------------------------ weight preparation
    @staticmethod
    def _internal_weights(cfg, w):
        """Reference-shaped weights of a use_id_embedding model -> the single-width internal form: product rows are
        [id embedding | text features] as the reference concatenates them (model/model.py:289), query rows
        [text features | d_id zeros]."""
        dq, did, h = cfg.d_in, cfg.d_id, cfg.h
        pad_cols = lambda t: torch.cat([t, t.new_zeros(t.shape[0], did)], dim=1)
        out = dict(w)
        out["id_table"] = w["item_table"]                           # kept for batches that bring their own text features
        if "item_text_table" in w:
            out["item_table"] = torch.cat([w["item_table"], w["item_text_table"]], dim=1)
        else:
            del out["item_table"]
        if "query_table" in w:
            out["query_table"] = pad_cols(w["query_table"])
        out["gat_qp.0.lin_src"] = pad_cols(w["gat_qp.0.lin_src"])   # reads query rows
        out["gat_pq.0.lin_dst"] = pad_cols(w["gat_pq.0.lin_dst"])
        ql = w["pool.query_lin.w"]
        out["pool.query_lin.w"] = torch.cat([ql[:, :dq], ql.new_zeros(ql.shape[0], did), ql[:, dq:]], dim=1)
        return out

    def _prepare(self, w):
        cfg, dev = self.cfg, self.device
        h = cfg.h
        f64 = lambda t: t.detach().to(torch.float64)
        d = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        self.item_table = d(w["item_table"]) if "item_table" in w else None
        self.query_table = d(w["query_table"]) if "query_table" in w else None
        self.id_table = d(w["id_table"]) if "id_table" in w else None      # use_id_embedding: the [V, d_id] NodeAsinEmbedding
        self.layers = []
        for l in range(cfg.n_layers):
            qp = {k: w[f"gat_qp.{l}.{k}"] for k in ("lin_src", "lin_dst", "att_src", "att_dst", "bias")}
            pq = {k: w[f"gat_pq.{l}.{k}"] for k in ("lin_src", "lin_dst", "att_src", "att_dst", "bias")}
            # alpha_src[j] = <lin_src x_j, att_src> = x_j . (lin_src^T att_src): one extra output row
            v = lambda lin, att: (f64(lin).T @ f64(att)).float()
            din = qp["lin_src"].shape[1]
            # product-side fused transform: [xs_p (p->q messages) | m = x Wg | gh = W_hh x | a_s(pq) | a_d(qp)]
            wp = torch.zeros((5 * h + ALPHA_PAD, din))
            wp[0:h] = pq["lin_src"]
            wp[h:2 * h, :] = w[f"ggc.{l}.weight"].T[:, :din]     # m = pad(x) @ weight
            wp[2 * h:5 * h, :] = w[f"ggc.{l}.w_hh"][:, :din]
            wp[5 * h] = v(pq["lin_src"], pq["att_src"])
            wp[5 * h + 1] = v(qp["lin_dst"], qp["att_dst"])
            bp = torch.zeros(5 * h + ALPHA_PAD)
            bp[2 * h:5 * h] = w[f"ggc.{l}.b_hh"]
            # query-side fused transform: [xs_q (q->p messages) | a_s(qp) | a_d(pq)]
            wq = torch.zeros((h + ALPHA_PAD, din))
            wq[0:h] = qp["lin_src"]
            wq[h] = v(qp["lin_src"], qp["att_src"])
            wq[h + 1] = v(pq["lin_dst"], pq["att_dst"])
            # fused-path product transform (k_layer_update's column layout):
            #   [xs_p | u = x (W_g W_ih^T) (3h) | gh = W_hh x + b_hh (3h) | a_s(pq) a_d(qp)]
            # the GRU input transform W_ih is applied BEFORE the neighbour sum (both are linear), so the
            # layer needs no second GEMM: gi[i] = sum_j u[j] + b_ih.  Product formed in float64.
            w7 = torch.zeros((7 * h + ALPHA_PAD, din))
            w7[0:h] = pq["lin_src"]
            w7[h:4 * h] = (f64(w[f"ggc.{l}.weight"])[:din, :] @ f64(w[f"ggc.{l}.w_ih"]).T).T.float()
            w7[4 * h:7 * h] = w[f"ggc.{l}.w_hh"][:, :din]
            w7[7 * h] = v(pq["lin_src"], pq["att_src"])
            w7[7 * h + 1] = v(qp["lin_dst"], qp["att_dst"])
            b7 = torch.zeros(7 * h + ALPHA_PAD)
            b7[4 * h:7 * h] = w[f"ggc.{l}.b_hh"]
            self.layers.append(dict(
                wp=d(wp), bp=d(bp), wq=d(wq), w_ih=d(w[f"ggc.{l}.w_ih"]), b_ih=d(w[f"ggc.{l}.b_ih"]),
                bias_qp=d(qp["bias"]), bias_pq=d(pq["bias"]), din=din, w7=d(w7), b7=d(b7)))
        # fused pooling (sss_pool_attention_tab): per-position tables and the stacked node-side weights
        P_, D_ = cfg.max_seq_len, cfg.d_out
        Dl_ = D_ - P_
        KT = (Dl_ + 31) // 32 * 32                      # T = tanh(lin) is stored KT wide (zero pad columns): a GEMM K
        tp = torch.tanh(f64(w["pool.pos_emb"]))
        wn, wc = f64(w["pool.node_lin.w"]), f64(w["pool.coarse_lin.w"])
        wnc = torch.zeros((2 * D_, KT), dtype=torch.float64)
        wnc[:D_, :Dl_] = wn[:, :Dl_]
        wnc[D_:, :Dl_] = wc[:, :Dl_]
        self.pool_tab = dict(KT=KT, tanhpos=d(tp.float()), a2=d((tp @ wn[:, Dl_:].T + f64(w["pool.node_lin.b"])).float()),
                             c2=d((tp @ wc[:, Dl_:].T).float()), wnc=d(wnc.float()))
        self.pool = dict(
            wq=d(w["pool.query_lin.w"]), bq=d(w["pool.query_lin.b"]),
            wp=d(w["pool.product_lin.w"]), bp=d(w["pool.product_lin.b"]),
            pos=d(w["pool.pos_emb"]), wn=d(w["pool.node_lin.w"]), bn=d(w["pool.node_lin.b"]),
            wc=d(w["pool.coarse_lin.w"]), watt=d(w["pool.att_lin.w"]))

    # ------------------------------------------------------------------ C-ABI call helpers
    def _st(self):
        return _lib.stream_ptr(self.device)

    def _linear(self, x, w, bias, n, m, k, out=None):
        if out is None:
            out = torch.empty((n, m), dtype=torch.float32, device=self.device)
        rc = _lib.lib().sss_linear(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
                                   0 if bias is None else bias.data_ptr(), out.data_ptr(), out.stride(0),
                                   n, m, k, self._st())
        _lib.check(rc, "sss_linear")
        return out

    def _gat(self, xs, a_src, a_dst, csr, n_dst, bias, relu, out, n_self_loop=0):
        rowptr, col, _ = csr
        rc = _lib.lib().sss_gat_aggregate(xs.data_ptr(), xs.stride(0), a_src.data_ptr(), a_src.stride(0),
                                          a_dst.data_ptr(), a_dst.stride(0), rowptr.data_ptr(), col.data_ptr(),
                                          n_dst, self.cfg.h, bias.data_ptr(), relu, n_self_loop, out.data_ptr(),
                                          out.stride(0), self._st())
        _lib.check(rc, "sss_gat_aggregate")

    # ------------------------------------------------------------------ batch preparation
    @torch.no_grad()
    def prepare(self, data):
        """Integer/index side of a batch, computed once and kept resident in HBM with it: the
        three CSR-by-target adjacencies (PyG self-loop rewrite applied per ``self_loop_rule``),
        the expanded-node index arrays of the pooling and the per-graph segment pointers.
        ``forward`` accepts the returned ``PreparedBatch`` and then launches arithmetic kernels
        only.  (In the reference this structural work is ``sequence_to_graph`` +
        ``Batch.from_data_list`` on the host, outside the model's forward.)"""
        if isinstance(data, PreparedBatch):
            return data
        if isinstance(data, ActionTable):
            return self.prepare_actions(data)
        cfg, L, dev = self.cfg, _lib.lib(), self.device
        q, p = data["query"], data["product"]
        pb = PreparedBatch()
        pb.q_batch = q.batch.to(dev, torch.int64).contiguous()
        pb.p_batch = p.batch.to(dev, torch.int64).contiguous()
        pb.Nq, pb.Np = int(pb.q_batch.shape[0]), int(pb.p_batch.shape[0])
        pb.B = int(getattr(data, "num_graphs", 0)) or int(max(pb.q_batch.max().item(), pb.p_batch.max().item()) + 1)
        pb.q_feat = getattr(q, "feat", None)
        pb.p_feat = getattr(p, "feat", None)
        pb.q_ids = None if pb.q_feat is not None else q.x.to(dev, torch.int64).contiguous()
        # (use_id_embedding: the product ids index the id embedding even when the text features are given)
        pb.p_ids = None if (pb.p_feat is not None and not self.d_id) else p.x.to(dev, torch.int64).contiguous()
        ei = data.edge_index_dict
        # the PyG self-loop rewrite is applied inside the aggregation kernels (n_self_loop)
        pb.n_self_loop = min(pb.Nq, pb.Np) if cfg.self_loop_rule == "pyg_bipartite_global" else 0
        ei_qp, ei_pq, ei_pp = (ei[k].to(dev, torch.int64) for k in (EDGE_QP, EDGE_PQ, EDGE_PP))
        pb.csr_qp = build_csr(ei_qp, pb.Np, pb.Nq, False)          # targets = products
        pb.csr_pq = build_csr(ei_pq, pb.Nq, pb.Np, False)          # targets = queries
        ew = None
        if self.use_edge_weight:
            ew = data.edge_weight_dict[EDGE_PP].to(dev, torch.float32)
        pb.csr_pp = build_csr(ei_pp, pb.Np, pb.Np, False, ew)
        cnt = p.cnt.to(dev, torch.int64)
        src_p = torch.repeat_interleave(torch.arange(pb.Np, device=dev, dtype=torch.int64), cnt)
        pb.n_clicks = int(src_p.shape[0])
        pb.src_row = torch.cat([src_p, torch.arange(pb.Nq, device=dev, dtype=torch.int64)]).to(torch.int32).contiguous()
        pb.pos_id = torch.cat([p.pos_emb_id.to(dev, torch.int64), q.pos_emb_id.to(dev, torch.int64)]) \
            .to(torch.int32).contiguous()
        if int(pb.pos_id.shape[0]) != pb.n_clicks + pb.Nq:
            raise _lib.SssError("product pos_emb_id must have sum(cnt) entries")
        self._check_ids(pb, check_min_pos=True)
        click_batch = pb.p_batch[src_p].contiguous()
        pb.pptr = torch.empty(pb.B + 1, dtype=torch.int32, device=dev)
        pb.qptr = torch.empty(pb.B + 1, dtype=torch.int32, device=dev)
        _lib.check(L.sss_segment_ptr(click_batch.data_ptr(), pb.n_clicks, pb.B, pb.pptr.data_ptr(), self._st()), "sss_segment_ptr")
        _lib.check(L.sss_segment_ptr(pb.q_batch.data_ptr(), pb.Nq, pb.B, pb.qptr.data_ptr(), self._st()), "sss_segment_ptr")
        return pb

    @torch.no_grad()
    def prepare_actions(self, actions):
        """Native graph construction (``csrc/graphbuild.hip``): a flat action table -- sessions
        stored contiguously: ``sess_ptr``, per action ``is_search`` / ``item_id`` / ``query_tok``
        (the CSV schema of the reference's decompose_data.py:13,30,42) -- becomes the prepared
        batch entirely on the device: what ``sequence_to_graph`` + ``Batch.from_data_list`` do on
        the host in the reference (util_amazon_filtered.py:98-230, test_amazon_filterd.py:485-488),
        followed by the CSR-by-target conversion.  Two kernel sweeps + one 5-integer read-back
        (the totals that size the outputs).  ``actions`` is an ``ActionTable`` (numpy) or an
        object with the same four attributes as device tensors."""
        cfg, L, dev = self.cfg, _lib.lib(), self.device
        st = self._st()

        def up(a, dtype):
            t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
            return t.to(dev, dtype).contiguous()
        sess_ptr = up(actions.sess_ptr, torch.int64)
        is_search = up(actions.is_search, torch.uint8)
        item_id = up(actions.item_id, torch.int64)
        query_tok = up(actions.query_tok, torch.int64)
        S_ = int(sess_ptr.shape[0] - 1)
        if S_ <= 0:
            raise _lib.SssError("prepare_actions: empty action table")
        bases = torch.empty((5, S_ + 1), dtype=torch.int32, device=dev)
        scratch = torch.empty(int(L.sss_graph_scratch_ints(S_)), dtype=torch.int32, device=dev)
        err = torch.empty(1, dtype=torch.int32, device=dev)
        _lib.check(L.sss_graph_counts(sess_ptr.data_ptr(), is_search.data_ptr(), item_id.data_ptr(), S_, bases.data_ptr(),
                                      scratch.data_ptr(), err.data_ptr(), st), "sss_graph_counts")
        lows = torch.stack([item_id.min(), query_tok.min()]).to(torch.int32) if item_id.numel() else bases.new_zeros(2)
        tot = torch.cat([bases[:, S_], err, lows]).tolist()        # the one host read-back of the build
        Nq, Np, Xp, E, Epp, bad, item_lo, tok_lo = (int(v) for v in tot)
        if bad:
            raise _lib.SssError("prepare_actions: a session has more than 64 actions")
        if item_lo < 0 or tok_lo < 0:
            raise IndexError("index out of range in self")         # negative ids: what nn.Embedding raises upstream
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
        i64 = lambda n: torch.empty(n, dtype=torch.int64, device=dev)
        pb = PreparedBatch()
        pb.Nq, pb.Np, pb.B, pb.n_clicks = Nq, Np, S_, Xp
        pb.q_ids, pb.q_batch, pb.q_pos = i64(Nq), i64(Nq), i32(Nq)
        pb.p_ids, pb.p_batch, pb.p_cnt = i64(Np), i64(Np), i64(Np)
        pb.q_feat = pb.p_feat = None
        rp_qp, c_qp, rp_pq, c_pq = i32(Np + 1), i32(E), i32(Nq + 1), i32(E)
        rp_pp, c_pp = i32(Np + 1), i32(Epp)
        w_pp = torch.empty(Epp, dtype=torch.float32, device=dev)
        pb.src_row, pb.pos_id = i32(Xp + Nq), i32(Xp + Nq)
        out = _lib.GraphOut(q_x=pb.q_ids.data_ptr(), q_batch=pb.q_batch.data_ptr(), q_pos=pb.q_pos.data_ptr(),
                            p_x=pb.p_ids.data_ptr(), p_batch=pb.p_batch.data_ptr(), p_cnt=pb.p_cnt.data_ptr(),
                            rowptr_qp=rp_qp.data_ptr(), col_qp=c_qp.data_ptr(), rowptr_pq=rp_pq.data_ptr(),
                            col_pq=c_pq.data_ptr(), rowptr_pp=rp_pp.data_ptr(), col_pp=c_pp.data_ptr(),
                            w_pp=w_pp.data_ptr(), src_row=pb.src_row.data_ptr(), pos_id=pb.pos_id.data_ptr())
        _lib.check(L.sss_graph_fill(sess_ptr.data_ptr(), is_search.data_ptr(), item_id.data_ptr(), query_tok.data_ptr(), S_,
                                    bases.data_ptr(), ctypes.byref(out), st), "sss_graph_fill")
        pb.csr_qp = (rp_qp, c_qp, None)
        pb.csr_pq = (rp_pq, c_pq, None)
        pb.csr_pp = (rp_pp, c_pp, w_pp if self.use_edge_weight else None)
        pb.w_pp = w_pp
        pb.qptr, pb.p_ptr, pb.pptr = bases[0], bases[1], bases[2]       # per-graph pointers come out of the scans
        pb.n_self_loop = min(Nq, Np) if cfg.self_loop_rule == "pyg_bipartite_global" else 0
        self._check_ids(pb)
        return pb

    def _check_ids(self, pb, check_min_pos=False):
        """Range checks of every index a kernel will dereference (positions, item / query feature ids),
        ONE host read-back; out of range raises what ``nn.Embedding`` raises upstream."""
        probes = []
        if pb.pos_id.numel():
            probes += [(pb.pos_id.max(), self.cfg.max_seq_len, False)]
            if check_min_pos:
                probes += [(pb.pos_id.min(), None, True)]
        p_table = self.id_table if (self.d_id and getattr(pb, "p_feat", None) is not None) else self.item_table
        for ids, table in ((pb.p_ids, p_table), (pb.q_ids, self.query_table)):
            if ids is not None and ids.numel() and table is not None:
                probes += [(ids.max(), int(table.shape[0]), False), (ids.min(), None, True)]
        if not probes:
            return
        vals = torch.stack([p[0].to(torch.int64) for p in probes]).tolist()
        for v, (_, bound, is_min) in zip(vals, probes):
            if (is_min and v < 0) or (not is_min and v >= bound):
                raise IndexError("index out of range in self")

    def _features(self, ids, feat, table, n, buf=None, product=False):
        W = self.cfg.node_width
        if buf is None:
            buf = torch.empty((n, W), dtype=torch.float32, device=self.device)
        if feat is not None and self.d_id:
            # use_id_embedding (model/model.py:288-289): product rows = [id_table[x] | feat], query rows = [feat | zeros]
            f = feat.to(self.device, torch.float32)
            if f.dim() != 2 or f.shape[1] != self.d_feat or f.stride(1) != 1 or f.stride(0) % 4:
                f = f.reshape(n, self.d_feat).contiguous()
            if product and (self.id_table is None or ids is None):
                raise _lib.SssError("use_id_embedding: product nodes need their item ids (.x) and the id table")
            rc = _lib.lib().sss_gather_concat_rows(self.id_table.data_ptr() if product else 0, ids.data_ptr() if product else 0,
                                                   self.d_id if product else 0, f.data_ptr(), f.stride(0), self.d_feat,
                                                   0 if product else self.d_id, n, buf.data_ptr(), buf.stride(0), self._st())
            _lib.check(rc, "sss_gather_concat_rows")
        elif feat is not None:
            buf[:, :self.cfg.d_in] = feat.to(self.device, torch.float32)
        else:
            if table is None:
                raise _lib.SssError("no feature table in the weights and no .feat on the batch")
            rc = _lib.lib().sss_gather_rows(table.data_ptr(), ids.data_ptr(), n, self.cfg.d_in,
                                            buf.data_ptr(), buf.stride(0), self._st())
            _lib.check(rc, "sss_gather_rows")
        return buf

    # ------------------------------------------------------------------ fused path (7 launches)
    def fused_ok(self) -> bool:
        """Shapes the fused kernels cover (one float4 column per lane: h, d_out <= 256); wider
        models -- the reference's h = 800, D = 1600 -- take the per-op kernels."""
        return self.fused and self.cfg.h <= 256 and self.cfg.d_out <= 256 and self.cfg.d_out % 4 == 0

    def _workspace(self, pb, fresh_nodes):
        """Intermediate buffers of the fused forward, kept with the prepared batch (no per-forward
        allocation); the node buffers are allocated fresh when the caller asked for them."""
        cfg, dev = self.cfg, self.device
        h, D, P, W = cfg.h, cfg.d_out, cfg.max_seq_len, cfg.node_width
        # The cache lives with the batch but belongs to THIS encoder: its buffers are sized by this
        # configuration and its argument blocks hold raw pointers to these weights.
        per_enc = getattr(pb, "_ws", None)
        if per_enc is None:
            per_enc = pb._ws = {}
        ws = per_enc.get(self._serial)
        if ws is None:
            e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            ldl = (D - P + 3) // 4 * 4
            n_exp = pb.n_clicks + pb.Nq
            KT = self.pool_tab["KT"]
            ws = dict(NQ=e(pb.Nq, W), NP=e(pb.Np, W), Yp=e(pb.Np, 7 * h + ALPHA_PAD), Yq=e(pb.Nq, h + ALPHA_PAD),
                      T=torch.zeros((pb.Np + pb.Nq, KT), dtype=torch.float32, device=dev),     # pad columns stay zero
                      AC=e(pb.Np + pb.Nq, 2 * D), calls={})
            per_enc[self._serial] = ws
        if fresh_nodes:
            ws = dict(ws, NQ=torch.empty((pb.Nq, W), dtype=torch.float32, device=dev),
                      NP=torch.empty((pb.Np, W), dtype=torch.float32, device=dev), calls={})
        return ws

    # Layer 0 over embedding-table features: a node's transforms depend only on its table row, so the
    # TABLES are transformed once (same kernel, hence the same bits as transforming the gathered rows) and
    # the layer kernel reads every node's row through its id -- one launch less per forward.
    TABLE_MODE_MAX_BYTES = 8 << 30

    def _layer0_tables(self):
        if self.item_table is None or self.query_table is None:
            return None
        if getattr(self, "_tabs", None) is None:
            h, lw = self.cfg.h, self.layers[0]
            ni, nqy = self.item_table.shape[0], self.query_table.shape[0]
            if (ni * (7 * h + ALPHA_PAD) + nqy * (h + ALPHA_PAD)) * 4 > self.TABLE_MODE_MAX_BYTES:
                self._tabs = ()
            else:
                Yp = torch.empty((ni, 7 * h + ALPHA_PAD), dtype=torch.float32, device=self.device)
                Yq = torch.empty((nqy, h + ALPHA_PAD), dtype=torch.float32, device=self.device)
                P_ = _lib.LinearProblem
                mk = lambda x, w, b, y, n, m: P_(x=x.data_ptr(), ldx=x.stride(0), ids=0, table=0, xcopy=0, ld_xcopy=0,
                                                 w=w.data_ptr(), ldw=w.stride(0), bias=0 if b is None else b.data_ptr(),
                                                 y=y.data_ptr(), ldy=y.stride(0), n=n, m=m, act=0)
                arr = (P_ * 2)(mk(self.item_table, lw["w7"], lw["b7"], Yp, ni, 7 * h + 2),
                               mk(self.query_table, lw["wq"], None, Yq, nqy, h + 2))
                _lib.check(_lib.lib().sss_linear_grouped(arr, 2, lw["din"], self._st()), "sss_linear_grouped")
                self._tabs = (Yp, Yq)
        return self._tabs or None

    def _fused_calls(self, pb, ws, gather):
        """ctypes argument blocks of the 7 launches that do not depend on the output tensor,
        built once per (prepared batch, workspace)."""
        key = "gather" if gather else "rows"
        if key in ws["calls"]:
            return ws["calls"][key]
        cfg = self.cfg
        h, D, P, W = cfg.h, cfg.d_out, cfg.max_seq_len, cfg.node_width
        NQ, NP, Yp, Yq = ws["NQ"], ws["NP"], ws["Yp"], ws["Yq"]
        P_ = _lib.LinearProblem

        def prob(x, w, bias, y, n, m, ids=None, table=None, xcopy=None):
            return P_(x=0 if x is None else x.data_ptr(), ldx=0 if x is None else x.stride(0),
                      ids=0 if ids is None else ids.data_ptr(), table=0 if table is None else table.data_ptr(),
                      xcopy=0 if xcopy is None else xcopy.data_ptr(), ld_xcopy=0 if xcopy is None else xcopy.stride(0),
                      w=w.data_ptr(), ldw=w.stride(0), bias=0 if bias is None else bias.data_ptr(),
                      y=y.data_ptr(), ldy=y.stride(0), n=n, m=m, act=0)

        steps, keep = [], []
        tabs = self._layer0_tables() if gather else None      # layer-0 transforms of the feature TABLES (weights only)
        for l, lw in enumerate(self.layers):
            off = 0 if l == 0 else cfg.d_in + (l - 1) * h
            din = lw["din"]
            xin_p, xin_q = NP[:, off:off + din], NQ[:, off:off + din]
            table0 = l == 0 and tabs is not None
            if table0:
                arr = None                                     # no per-batch transform: nodes read their table row's
            elif l == 0 and gather:
                pp = prob(None, lw["w7"], lw["b7"], Yp, pb.Np, 7 * h + 2, pb.p_ids, self.item_table, xin_p)
                pq = prob(None, lw["wq"], None, Yq, pb.Nq, h + 2, pb.q_ids, self.query_table, xin_q)
                arr = (P_ * 2)(pp, pq)
            else:
                pp = prob(xin_p, lw["w7"], lw["b7"], Yp, pb.Np, 7 * h + 2)
                pq = prob(xin_q, lw["wq"], None, Yq, pb.Nq, h + 2)
                arr = (P_ * 2)(pp, pq)
            out_p = NP[:, cfg.d_in + l * h: cfg.d_in + (l + 1) * h]
            out_q = NQ[:, cfg.d_in + l * h: cfg.d_in + (l + 1) * h]
            rp_qp, c_qp, _ = pb.csr_qp
            rp_pq, c_pq, _ = pb.csr_pq
            rp_pp, c_pp, w_pp = pb.csr_pp
            yp_l, yq_l = (tabs if table0 else (Yp, Yq))
            xsrc = self.item_table if table0 else xin_p
            la = _lib.LayerArgs(yp=yp_l.data_ptr(), ld_yp=yp_l.stride(0), yq=yq_l.data_ptr(), ld_yq=yq_l.stride(0), h=h, d_x=din,
                                rowptr_qp=rp_qp.data_ptr(), col_qp=c_qp.data_ptr(), rowptr_pp=rp_pp.data_ptr(),
                                col_pp=c_pp.data_ptr(), w_pp=0 if w_pp is None else w_pp.data_ptr(),
                                bias_qp=lw["bias_qp"].data_ptr(), b_ih=lw["b_ih"].data_ptr(),
                                xin_p=xsrc.data_ptr(), ld_xin=xsrc.stride(0), out_p=out_p.data_ptr(), ld_out_p=NP.stride(0),
                                np=pb.Np, rowptr_pq=rp_pq.data_ptr(), col_pq=c_pq.data_ptr(),
                                bias_pq=lw["bias_pq"].data_ptr(), out_q=out_q.data_ptr(), ld_out_q=NQ.stride(0), nq=pb.Nq,
                                n_self_loop=pb.n_self_loop,
                                row_p=pb.p_ids.data_ptr() if table0 else 0, row_q=pb.q_ids.data_ptr() if table0 else 0,
                                x0_p=NP.data_ptr() if table0 else 0, ld_x0_p=NP.stride(0),
                                xq_table=self.query_table.data_ptr() if table0 else 0,
                                ld_xq=self.query_table.stride(0) if table0 else 0,
                                x0_q=NQ.data_ptr() if table0 else 0, ld_x0_q=NQ.stride(0))
            if arr is not None:
                steps.append(("lin", arr, 2, din))
            steps.append(("layer", la))
            keep += [arr, la]
        pw, pt = self.pool, self.pool_tab
        Dl = D - P
        T, AC = ws["T"], ws["AC"]
        pp, pq = prob(NP, pw["wp"], pw["bp"], T[:pb.Np], pb.Np, Dl), prob(NQ, pw["wq"], pw["bq"], T[pb.Np:], pb.Nq, Dl)
        pp.act = pq.act = 2                                   # T = tanh(lin)
        arr = (P_ * 2)(pp, pq)
        steps.append(("lin", arr, 2, W))
        arr2 = (P_ * 1)(prob(T, pt["wnc"], None, AC, pb.Np + pb.Nq, 2 * D))      # [A1 | C1]
        steps.append(("lin", arr2, 1, pt["KT"]))
        ws["calls"][key] = steps
        return steps

    def _forward_fused(self, pb, get_node, query_node_mask, product_node_mask, l2_normalize):
        cfg, L, dev = self.cfg, _lib.lib(), self.device
        h, D, P, W = cfg.h, cfg.d_out, cfg.max_seq_len, cfg.node_width
        ws = self._workspace(pb, get_node)
        NQ, NP = ws["NQ"], ws["NP"]
        masked = query_node_mask is not None or product_node_mask is not None
        gather = pb.q_feat is None and pb.p_feat is None and not masked
        if not gather:      # features given / masked inputs: materialise slice 0 first (per-op kernels)
            self._features(pb.q_ids, pb.q_feat, self.query_table, pb.Nq, NQ)
            self._features(pb.p_ids, pb.p_feat, self.item_table, pb.Np, NP, product=True)
            if query_node_mask is not None:
                NQ[:, :cfg.d_in] *= query_node_mask.to(dev, torch.float32).view(-1, 1)
            if product_node_mask is not None:
                NP[:, :cfg.d_in] *= product_node_mask.to(dev, torch.float32).view(-1, 1)
        elif self.item_table is None or self.query_table is None:
            raise _lib.SssError("no feature table in the weights and no .feat on the batch")
        st = self._st()
        pw = self.pool
        for step in self._fused_calls(pb, ws, gather):
            if step[0] == "lin":
                _lib.check(L.sss_linear_grouped(step[1], step[2], step[3], st), "sss_linear_grouped")
            else:
                _lib.check(L.sss_hetero_layer_update(ctypes.byref(step[1]), st), "sss_hetero_layer_update")
        out = torch.empty((pb.B, D), dtype=torch.float32, device=dev)
        pt, T, AC = self.pool_tab, ws["T"], ws["AC"]
        rc = L.sss_pool_attention_tab(T.data_ptr(), T.stride(0), AC.data_ptr(), AC.stride(0), pt["tanhpos"].data_ptr(),
                                      pt["a2"].data_ptr(), pt["c2"].data_ptr(), pw["watt"].data_ptr(), pb.src_row.data_ptr(),
                                      pb.pos_id.data_ptr(), pb.pptr.data_ptr(), pb.qptr.data_ptr(), pb.n_clicks, pb.Np, pb.B,
                                      D - P, P, 1 if l2_normalize else 0, 1e-6, out.data_ptr(), D, st)
        _lib.check(rc, "sss_pool_attention_tab")
        return out, NQ, NP

    @torch.no_grad()
    def forward(self, data, query_node_mask=None, product_node_mask=None, get_node=False, get_token=False,
                l2_normalize=False):
        """``l2_normalize=True`` (extension) returns ``normalize(forward(data))`` -- the reference's
        util_amazon_filtered.normalize -- fused into the last pooling kernel."""
        cfg, L, dev = self.cfg, _lib.lib(), self.device
        h, D, P, W = cfg.h, cfg.d_out, cfg.max_seq_len, cfg.node_width
        pb = self.prepare(data)
        Nq, Np, B = pb.Nq, pb.Np, pb.B
        if self.fused_ok():
            out, NQ, NP = self._forward_fused(pb, get_node, query_node_mask, product_node_mask, l2_normalize)
            if self.debug_nan_checks and (torch.isnan(NQ).any() or torch.isnan(NP).any()):
                raise RuntimeError("nan in node embedding")
            return self._pack(out, NQ, NP, get_node, get_token)

        NQ = self._features(pb.q_ids, pb.q_feat, self.query_table, Nq)   # [Nq, W]; slice 0 = input features
        NP = self._features(pb.p_ids, pb.p_feat, self.item_table, Np, product=True)    # embedding lookup (NodeAsinEmbedding)
        if query_node_mask is not None:                   # model/model.py:293-296 (None at inference)
            NQ[:, :cfg.d_in] *= query_node_mask.to(dev, torch.float32).view(-1, 1)
        if product_node_mask is not None:
            NP[:, :cfg.d_in] *= product_node_mask.to(dev, torch.float32).view(-1, 1)
        if self.debug_nan_checks and (torch.isnan(NQ[:, :cfg.d_in]).any() or torch.isnan(NP[:, :cfg.d_in]).any()):
            raise RuntimeError("nan in embedding[query]")   # model/model.py:312

        csr_qp, csr_pq, csr_pp = pb.csr_qp, pb.csr_pq, pb.csr_pp
        mp, mq = 5 * h + ALPHA_PAD, h + ALPHA_PAD
        Yp = torch.empty((Np, mp), dtype=torch.float32, device=dev)
        Yq = torch.empty((Nq, mq), dtype=torch.float32, device=dev)
        T1 = torch.empty((Np, h), dtype=torch.float32, device=dev)
        T2 = torch.empty((Np, h), dtype=torch.float32, device=dev)
        T3 = torch.empty((Np, 3 * h), dtype=torch.float32, device=dev)
        for l, lw in enumerate(self.layers):
            off = 0 if l == 0 else cfg.d_in + (l - 1) * h
            din = lw["din"]
            xin_p, xin_q = NP[:, off:off + din], NQ[:, off:off + din]
            out_p = NP[:, cfg.d_in + l * h: cfg.d_in + (l + 1) * h]
            out_q = NQ[:, cfg.d_in + l * h: cfg.d_in + (l + 1) * h]
            self._linear(xin_p, lw["wp"], lw["bp"], Np, mp, din, Yp)
            self._linear(xin_q, lw["wq"], None, Nq, mq, din, Yq)
            # products <- queries (GAT) ; products <- products (GGC) ; GRU + sum + relu
            self._gat(Yq[:, :h], Yq[:, h], Yp[:, 5 * h + 1], csr_qp, Np, lw["bias_qp"], 0, T1, pb.n_self_loop)
            rowptr, col, wv = csr_pp
            rc = L.sss_csr_weighted_sum(Yp[:, h:2 * h].data_ptr(), Yp.stride(0), rowptr.data_ptr(), col.data_ptr(),
                                        0 if wv is None else wv.data_ptr(), Np, h, T2.data_ptr(), T2.stride(0),
                                        self._st())
            _lib.check(rc, "sss_csr_weighted_sum")
            self._linear(T2, lw["w_ih"], lw["b_ih"], Np, 3 * h, h, T3)
            rc = L.sss_gru_combine(T3.data_ptr(), T3.stride(0), Yp[:, 2 * h:].data_ptr(), Yp.stride(0),
                                   xin_p.data_ptr(), NP.stride(0), din, T1.data_ptr(), T1.stride(0), Np, h,
                                   out_p.data_ptr(), NP.stride(0), self._st())
            _lib.check(rc, "sss_gru_combine")
            # queries <- products (GAT) + relu
            self._gat(Yp[:, :h], Yp[:, 5 * h], Yq[:, h + 1], csr_pq, Nq, lw["bias_pq"], 1, out_q, pb.n_self_loop)
        if self.debug_nan_checks and (torch.isnan(NQ).any() or torch.isnan(NP).any()):
            raise RuntimeError("nan in node embedding")

        # ---- PositionalAttentionPooling (model/gnn.py:193-217)
        pw = self.pool
        Dl = D - P
        ldl = (Dl + 3) // 4 * 4
        lin_q = torch.empty((Nq, ldl), dtype=torch.float32, device=dev)
        lin_p = torch.empty((Np, ldl), dtype=torch.float32, device=dev)
        self._linear(NQ, pw["wq"], pw["bq"], Nq, Dl, W, lin_q)
        self._linear(NP, pw["wp"], pw["bp"], Np, Dl, W, lin_p)
        n_clicks, n_exp = pb.n_clicks, pb.n_clicks + Nq
        node = torch.empty((n_exp, D), dtype=torch.float32, device=dev)
        rc = L.sss_pool_expand(lin_p.data_ptr(), lin_q.data_ptr(), ldl, pb.src_row.data_ptr(), pb.pos_id.data_ptr(),
                               n_clicks, n_exp, Dl, P, pw["pos"].data_ptr(), node.data_ptr(), node.stride(0),
                               self._st())
        _lib.check(rc, "sss_pool_expand")
        A = self._linear(node, pw["wn"], pw["bn"], n_exp, D, D)
        pptr, qptr = pb.pptr, pb.qptr
        coarse = torch.empty((B, D), dtype=torch.float32, device=dev)
        rc = L.sss_segment_pool(node.data_ptr(), node.stride(0), pptr.data_ptr(), qptr.data_ptr(), n_clicks, B, D,
                                0, 0, 0, 0, 0, coarse.data_ptr(), coarse.stride(0), self._st())
        _lib.check(rc, "sss_segment_pool(mean)")
        Bc = self._linear(coarse, pw["wc"], None, B, D, D)
        out = torch.empty((B, D), dtype=torch.float32, device=dev)
        rc = L.sss_segment_pool(node.data_ptr(), node.stride(0), pptr.data_ptr(), qptr.data_ptr(), n_clicks, B, D,
                                A.data_ptr(), A.stride(0), Bc.data_ptr(), Bc.stride(0), pw["watt"].data_ptr(),
                                out.data_ptr(), out.stride(0), self._st())
        _lib.check(rc, "sss_segment_pool(att)")

        if l2_normalize:
            from .index import normalize_
            normalize_(out)
        return self._pack(out, NQ, NP, get_node, get_token)

    def _pack(self, out, NQ, NP, get_node, get_token):
        if self.d_id and get_node:      # the reference's query node rows are d_in + L h wide: drop the internal pad columns
            NQ = torch.cat([NQ[:, :self.d_feat], NQ[:, self.d_feat + self.d_id:]], dim=1)
        node_embedding = {"query": NQ, "product": NP}
        session_level_token_emb = {}            # the cross-attention branch is commented out upstream
        if not get_node and not get_token:
            return out
        if get_node and not get_token:
            return out, node_embedding
        if get_token and not get_node:
            return out, session_level_token_emb
        return out, node_embedding, session_level_token_emb

"""Oracle: session encoder (embedding lookup -> HeteroGGNN -> positional-attention pooling).
TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  PARITY UNPINNED (no reference
fixture exists for this path; PyG 2.0.4 is absent): the op semantics follow SURVEY.md
Appendix A.  ``torch.nn.GRUCell``, ``F.embedding``, ``F.linear`` and ``F.leaky_relu`` of the
installed torch are used directly.

Weights are a flat ``{name: tensor}`` dict (names: DESIGN.md "weight file"):

  item_table [V,d_in]   query_table [VQ,d_in]
  gat_qp.{l}.lin_src [h,d_q] .lin_dst [h,d_p] .att_src [h] .att_dst [h] .bias [h]   (query -> product)
  gat_pq.{l}.*                                                                  (product -> query)
  ggc.{l}.weight [h,h]  .w_ih [3h,h] .w_hh [3h,h] .b_ih [3h] .b_hh [3h]
  pool.query_lin.w [D-P,W] .b   pool.product_lin.w [D-P,W] .b   pool.pos_emb [P,P]
  pool.node_lin.w [D,D] .b      pool.coarse_lin.w [D,D]         pool.att_lin.w [D]
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

EDGE_QP = ("query", "clicks", "product")
EDGE_PQ = ("product", "clicked by", "query")
EDGE_PP = ("product", "to", "product")


def embedding_lookup(table, ids):
    """``NodeAsinEmbedding.forward`` (reference model/NodeEmbedding.py:137-138)."""
    return F.embedding(ids, table)


def rewrite_self_loops(edge_index, n_src, n_dst):
    """PyG ``GATConv(add_self_loops=True)`` edge rewrite, bipartite form (Appendix A.2):
    drop edges with src index == dst index, append (i -> i) for i < min(n_src, n_dst);
    indices are batch-global (hard part H3 of SURVEY.md)."""
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    loop = torch.arange(min(n_src, n_dst), dtype=src.dtype)
    return torch.stack([torch.cat([src[keep], loop]), torch.cat([dst[keep], loop])])


def gat_conv(x_src, x_dst, edge_index, lin_src, lin_dst, att_src, att_dst, bias,
             self_loops=True):
    """Bipartite ``GATConv((-1,-1), h)``, heads=1, negative_slope 0.2 (Appendix A.2);
    instantiated at reference model/gnn.py:54."""
    n_src, n_dst = x_src.shape[0], x_dst.shape[0]
    xs = F.linear(x_src, lin_src)
    xd = F.linear(x_dst, lin_dst)
    a_s = (xs * att_src).sum(-1)
    a_d = (xd * att_dst).sum(-1)
    if self_loops:
        edge_index = rewrite_self_loops(edge_index, n_src, n_dst)
    src, dst = edge_index[0], edge_index[1]
    e = F.leaky_relu(a_s[src] + a_d[dst], 0.2)
    emax = torch.full((n_dst,), float("-inf"), dtype=e.dtype)
    emax = emax.scatter_reduce(0, dst, e, reduce="amax", include_self=True)
    ex = torch.exp(e - emax[dst])
    den = torch.zeros(n_dst, dtype=e.dtype).index_add_(0, dst, ex)
    w = ex / (den[dst] + 1e-16)
    out = torch.zeros(n_dst, xs.shape[1], dtype=xs.dtype).index_add_(0, dst, w[:, None] * xs[src])
    return out + bias


def gated_graph_conv(x, edge_index, weight, w_ih, w_hh, b_ih, b_hh, edge_weight=None):
    """``GatedGraphConv(h, num_layers=1)`` (Appendix A.3); reference model/gnn.py:58."""
    h = weight.shape[0]
    if x.shape[1] > h:
        raise ValueError("The number of input channels is not allowed to be larger than "
                         "the number of output channels")
    if x.shape[1] < h:
        x = torch.cat([x, x.new_zeros(x.shape[0], h - x.shape[1])], dim=1)
    m = x @ weight
    src, dst = edge_index[0], edge_index[1]
    msg = m[src] if edge_weight is None else edge_weight[:, None].to(m.dtype) * m[src]
    magg = torch.zeros_like(m).index_add_(0, dst, msg)
    cell = torch.nn.GRUCell(h, h, bias=True)
    cell = cell.to(x.dtype)
    with torch.no_grad():
        cell.weight_ih.copy_(w_ih); cell.weight_hh.copy_(w_hh)
        cell.bias_ih.copy_(b_ih); cell.bias_hh.copy_(b_hh)
        return cell(magg, x)


def hetero_ggnn(x_q, x_p, edge_index_dict, w, n_layers, self_loops=True, edge_weight_dict=None,
                add_input_feat=True):
    """``HeteroGGNN.forward`` (reference model/gnn.py:64-81): per layer
    HeteroConv{GAT q->p, GAT p->q, GGC p->p} summed per dst type, ReLU, concat of layers."""
    outs_q, outs_p = [x_q], [x_p]
    cq, cp = x_q, x_p
    for l in range(n_layers):
        p_from_q = gat_conv(cq, cp, edge_index_dict[EDGE_QP], w[f"gat_qp.{l}.lin_src"],
                            w[f"gat_qp.{l}.lin_dst"], w[f"gat_qp.{l}.att_src"],
                            w[f"gat_qp.{l}.att_dst"], w[f"gat_qp.{l}.bias"], self_loops)
        q_from_p = gat_conv(cp, cq, edge_index_dict[EDGE_PQ], w[f"gat_pq.{l}.lin_src"],
                            w[f"gat_pq.{l}.lin_dst"], w[f"gat_pq.{l}.att_src"],
                            w[f"gat_pq.{l}.att_dst"], w[f"gat_pq.{l}.bias"], self_loops)
        ew = None if edge_weight_dict is None else edge_weight_dict.get(EDGE_PP)
        p_from_p = gated_graph_conv(cp, edge_index_dict[EDGE_PP], w[f"ggc.{l}.weight"],
                                    w[f"ggc.{l}.w_ih"], w[f"ggc.{l}.w_hh"],
                                    w[f"ggc.{l}.b_ih"], w[f"ggc.{l}.b_hh"], ew)
        # HeteroConv(aggr='sum'): stack(...).sum(0) over edge types per dst (Appendix A.1)
        cp = torch.relu(torch.stack([p_from_q, p_from_p]).sum(0))
        cq = torch.relu(q_from_p)
        outs_q.append(cq); outs_p.append(cp)
    if not add_input_feat:
        outs_q, outs_p = outs_q[1:], outs_p[1:]
    return torch.cat(outs_q, dim=1), torch.cat(outs_p, dim=1)


def global_mean_pool(x, batch, size):
    """PyG ``global_mean_pool`` (Appendix A.4)."""
    s = torch.zeros(size, x.shape[1], dtype=x.dtype).index_add_(0, batch, x)
    c = torch.zeros(size, dtype=x.dtype).index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return s / c.clamp(min=1)[:, None]


def pos_att_pool(node_q, node_p, q_pos, q_batch, p_cnt, p_pos, p_batch, num_graphs, w):
    """``PositionalAttentionPooling.forward`` (reference model/gnn.py:193-217)."""
    q = F.linear(node_q, w["pool.query_lin.w"], w["pool.query_lin.b"])
    p = F.linear(node_p, w["pool.product_lin.w"], w["pool.product_lin.b"])
    q = torch.tanh(torch.cat([q, F.embedding(q_pos, w["pool.pos_emb"])], dim=1))
    p = torch.repeat_interleave(p, p_cnt, dim=0)
    p = torch.tanh(torch.cat([p, F.embedding(p_pos, w["pool.pos_emb"])], dim=1))
    pb = torch.repeat_interleave(p_batch, p_cnt, dim=0)
    node = torch.cat([p, q], dim=0)
    nb = torch.cat([pb, q_batch], dim=0)
    coarse = global_mean_pool(node, nb, num_graphs)[nb]
    a = F.linear(node, w["pool.node_lin.w"], w["pool.node_lin.b"])
    b = F.linear(coarse, w["pool.coarse_lin.w"])
    att = F.linear(torch.sigmoid(a + b), w["pool.att_lin.w"].view(1, -1))
    return global_mean_pool(node * att, nb, num_graphs)


def encoder_forward(batch, w, n_layers, self_loops=True, dtype=torch.float32, get_node=False,
                    query_node_mask=None, product_node_mask=None, use_id_embedding=False):
    """``UnifyPoolingGraphLevelEncoder.forward`` (reference model/model.py:279-351) with the
    text encoder replaced by a feature-table lookup or by precomputed ``.feat`` rows on the node stores
    (out-of-scope boundary).  ``batch`` is a SessionBatch of torch CPU tensors.
    ``use_id_embedding`` (model/model.py:288-291): ``embedding['product'] = torch.concat((a, b), dim=1)`` with
    ``a = product_node_embedder(x)`` (``item_table``, NodeAsinEmbedding) and ``b`` the product text features
    (``.feat`` or the ``item_text_table`` stand-in); False: ``embedding['product'] = b``."""
    w = {k: v.to(dtype) if v.is_floating_point() else v for k, v in w.items()}
    q_feat, p_feat = getattr(batch["query"], "feat", None), getattr(batch["product"], "feat", None)
    xq = q_feat.to(dtype) if q_feat is not None else embedding_lookup(w["query_table"], batch["query"].x)
    if use_id_embedding:
        a = embedding_lookup(w["item_table"], batch["product"].x)
        b = p_feat.to(dtype) if p_feat is not None else embedding_lookup(w["item_text_table"], batch["product"].x)
        xp = torch.concat((a, b), dim=1)
    else:
        xp = p_feat.to(dtype) if p_feat is not None else embedding_lookup(w["item_table"], batch["product"].x)
    if query_node_mask is not None:
        xq = xq * query_node_mask.view(-1, 1).to(dtype)
    if product_node_mask is not None:
        xp = xp * product_node_mask.view(-1, 1).to(dtype)
    nq, np_ = hetero_ggnn(xq, xp, batch.edge_index_dict, w, n_layers, self_loops)
    out = pos_att_pool(nq, np_, batch["query"].pos_emb_id, batch["query"].batch,
                       batch["product"].cnt, batch["product"].pos_emb_id,
                       batch["product"].batch, batch.num_graphs, w)
    if get_node:
        return out, {"query": nq, "product": np_}
    return out

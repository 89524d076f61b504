"""Oracle: the reference's other conv / pool variants, restated with torch CPU ops.
TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED: PyG is absent; SAGEConv
semantics follow PyG 2.0.4 (``out = lin_l(mean_j x_j) + lin_r(x_i)``, lin_l with bias, lin_r without,
``to_hetero`` summing the per-edge-type outputs per destination type).

* ``hetero_sage``      <- model/gnn.py:83-121 (GNN + to_hetero(aggr='sum'))
* ``graph_pooling``    <- model/gnn.py:123-143
* ``attention_pooling``<- model/gnn.py:145-161 (incl. its dense [n_nodes, n_graphs] matrix)
* ``srgnn_pooling``    <- model/gnn.py:164-181
* ``mlp``              <- model/model.py:40-73 (eval mode)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .gnn_ref import EDGE_PP, EDGE_PQ, EDGE_QP, global_mean_pool


def _scatter_mean(x_src, edge_index, n_dst):
    src, dst = edge_index[0], edge_index[1]
    s = torch.zeros(n_dst, x_src.shape[1], dtype=x_src.dtype).index_add_(0, dst, x_src[src])
    c = torch.zeros(n_dst, dtype=x_src.dtype).index_add_(0, dst, torch.ones_like(dst, dtype=x_src.dtype))
    return s / c.clamp(min=1)[:, None]


def sage_conv(x_src, x_dst, edge_index, lin_l_w, lin_l_b, lin_r_w):
    return F.linear(_scatter_mean(x_src, edge_index, x_dst.shape[0]), lin_l_w, lin_l_b) + F.linear(x_dst, lin_r_w)


def hetero_sage(x_q, x_p, edge_index_dict, w, n_layers=3):
    for l in range(n_layers):
        g = lambda e, n: w[f"sage.{l}.{e}.{n}"]
        p = sage_conv(x_q, x_p, edge_index_dict[EDGE_QP], g("qp", "lin_l.w"), g("qp", "lin_l.b"), g("qp", "lin_r.w")) + \
            sage_conv(x_p, x_p, edge_index_dict[EDGE_PP], g("pp", "lin_l.w"), g("pp", "lin_l.b"), g("pp", "lin_r.w"))
        q = sage_conv(x_p, x_q, edge_index_dict[EDGE_PQ], g("pq", "lin_l.w"), g("pq", "lin_l.b"), g("pq", "lin_r.w"))
        x_q, x_p = torch.relu(q), torch.relu(p)
    return {"query": x_q, "product": x_p}


def graph_pooling(x, batch, size, key, w):
    if key == "mean":
        p = global_mean_pool(x, batch, size)
    elif key == "add":
        p = torch.zeros(size, x.shape[1], dtype=x.dtype).index_add_(0, batch, x)
    elif key == "max":
        p = torch.full((size, x.shape[1]), float("-inf"), dtype=x.dtype).scatter_reduce(
            0, batch[:, None].expand_as(x), x, reduce="amax", include_self=True)
        p = torch.where(torch.isinf(p), torch.zeros_like(p), p)
    else:
        raise Exception("Unrecognized pooling key: " + key)
    return F.linear(p, w["lin.w"], w["lin.b"])


def attention_pooling(x, batch, size, w):
    coarse = global_mean_pool(x, batch, size)
    att = x @ coarse.T                                  # num_nodes x num_graphs, as the reference builds it
    att = att[torch.arange(att.size(0)), batch]
    return F.linear(global_mean_pool(x * att.view(-1, 1), batch, size), w["lin.w"], w["lin.b"])


def srgnn_pooling(x, batch, size, last_click_mask, w):
    add = lambda v: torch.zeros(size, v.shape[1], dtype=v.dtype).index_add_(0, batch, v)
    local = add(x * last_click_mask.view(-1, 1))
    att = F.linear(torch.sigmoid(F.linear(local[batch], w["lin1.w"], w["lin1.b"]) + F.linear(x, w["lin2.w"], w["lin2.b"])),
                   w["lin3.w"])
    return F.linear(torch.cat([local, add(x * att)], dim=1), w["lin4.w"], w["lin4.b"])


def mlp(x, w, n_hidden_layers, last_act=True, jump=False):
    inp = x
    for i in range(n_hidden_layers + 1):
        x = F.relu(F.linear(x, w[f"layers.{i}.w"], w[f"layers.{i}.b"]))
        x = F.relu(F.batch_norm(x, w[f"bn.{i}.mean"], w[f"bn.{i}.var"], w[f"bn.{i}.gamma"], w[f"bn.{i}.beta"], False, 0.0, 1e-5))
    if jump:
        x = torch.cat([inp, x], dim=1)
    last = n_hidden_layers + 1
    y = F.linear(x, w[f"layers.{last}.w"], w[f"layers.{last}.b"])
    return torch.tanh(y) if last_act else y


def binarize_head(x, w, mlp_w=None, n_hidden_layers=0, mlp_last_act=True, jump=False, pre_sign=False):
    """``BinarizeHead.forward`` in eval mode, statement by statement (model/model.py:117-135): ``lin1(x)`` without an
    mlp, else ``lin1(tanh(mlp(x)))`` (``jump``: ``lin1(cat([tanh(mlp(x)), x]))``), then
    ``(sign(out) - tanh(out)) + tanh(out)`` -- the reference's straight-through expression, kept as written."""
    if mlp_w is None:
        out = F.linear(x, w["lin1.w"], w["lin1.b"])
    else:
        out = torch.tanh(mlp(x, mlp_w, n_hidden_layers, mlp_last_act, False))
        if jump:
            out = torch.cat([out, x], dim=1)
        out = F.linear(out, w["lin1.w"], w["lin1.b"])
    if pre_sign:
        return out
    return (torch.sign(out) - torch.tanh(out)) + torch.tanh(out)

"""Second, INDEPENDENT oracle of the session encoder: plain numpy float64, explicit per-edge /
per-node / per-graph Python loops, written from the formulas of SURVEY.md Appendix A.1-A.4 and the
reference's own ``PositionalAttentionPooling.forward`` (model/gnn.py:193-217) -- no
``torch.nn.GRUCell``, no ``F.linear``, no ``index_add_``/``scatter_reduce``.
TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

PARITY UNPINNED, like ``gnn_ref``: PyG 2.0.4 is absent and the reference holds no fixture for this
arithmetic.  This file exists because a second restatement that shares no code with the first is
the only cross-check available: ``tests/test_oracle_cpu.py`` requires the two
to agree to 1e-12 and pins both with ``tests/golden/encoder_tiny.npz``.

Weight names and shapes: see ``gnn_ref`` (a flat ``{name: array}`` dict).
"""
from __future__ import annotations

import math

import numpy as np


def _f64(a):
    return np.asarray(a.detach().cpu().numpy() if hasattr(a, "detach") else a, dtype=np.float64)


def _i64(a):
    return np.asarray(a.detach().cpu().numpy() if hasattr(a, "detach") else a, dtype=np.int64)


def _sigmoid(v):
    return 1.0 / (1.0 + math.exp(-v))


def _matvec(W, x):
    """y[o] = sum_i W[o, i] * x[i], accumulated left to right."""
    out = np.zeros(W.shape[0])
    for o in range(W.shape[0]):
        acc = 0.0
        for i in range(W.shape[1]):
            acc += W[o, i] * x[i]
        out[o] = acc
    return out


def gat_conv(x_src, x_dst, edges, lin_src, lin_dst, att_src, att_dst, bias, self_loops):
    """Appendix A.2 (PyG GATConv((-1,-1), h), heads=1, slope 0.2, add_self_loops)."""
    ns, nd, h = x_src.shape[0], x_dst.shape[0], lin_src.shape[0]
    xs = np.stack([_matvec(lin_src, x_src[j]) for j in range(ns)]) if ns else np.zeros((0, h))
    xd = np.stack([_matvec(lin_dst, x_dst[i]) for i in range(nd)]) if nd else np.zeros((0, h))
    a_s = [float(sum(xs[j, c] * att_src[c] for c in range(h))) for j in range(ns)]
    a_d = [float(sum(xd[i, c] * att_dst[c] for c in range(h))) for i in range(nd)]
    pairs = [(int(j), int(i)) for j, i in zip(edges[0], edges[1])]
    if self_loops:                      # drop src idx == dst idx, append i -> i for i < min(ns, nd)
        pairs = [(j, i) for (j, i) in pairs if j != i] + [(i, i) for i in range(min(ns, nd))]
    incoming = [[] for _ in range(nd)]
    for j, i in pairs:
        incoming[i].append(j)
    out = np.zeros((nd, h))
    for i in range(nd):
        if incoming[i]:
            e = []
            for j in incoming[i]:
                v = a_s[j] + a_d[i]
                e.append(v if v > 0 else 0.2 * v)
            m = max(e)
            ex = [math.exp(v - m) for v in e]
            den = sum(ex) + 1e-16
            for j, w in zip(incoming[i], ex):
                out[i] += (w / den) * xs[j]
        out[i] += bias
    return out


def gated_graph_conv(x, edges, weight, w_ih, w_hh, b_ih, b_hh, edge_weight=None):
    """Appendix A.3 (PyG GatedGraphConv(h, 1) + the GRUCell gate equations written out)."""
    n, h = x.shape[0], weight.shape[0]
    if x.shape[1] > h:
        raise ValueError("input wider than output")
    xp = np.zeros((n, h))
    xp[:, :x.shape[1]] = x
    m = np.stack([_matvec(weight.T, xp[i]) for i in range(n)]) if n else np.zeros((0, h))   # x @ W
    agg = np.zeros((n, h))
    for e, (j, i) in enumerate(zip(edges[0], edges[1])):
        w = 1.0 if edge_weight is None else float(edge_weight[e])
        agg[int(i)] += w * m[int(j)]
    out = np.zeros((n, h))
    for i in range(n):
        gi = _matvec(w_ih, agg[i]) + b_ih
        gh = _matvec(w_hh, xp[i]) + b_hh
        for c in range(h):
            r = _sigmoid(gi[c] + gh[c])
            z = _sigmoid(gi[h + c] + gh[h + c])
            nn = math.tanh(gi[2 * h + c] + r * gh[2 * h + c])
            out[i, c] = (1.0 - z) * nn + z * xp[i, c]
    return out


def hetero_ggnn(x_q, x_p, ei_qp, ei_pq, ei_pp, w, n_layers, self_loops, ew_pp=None):
    """model/gnn.py:64-81 -- HeteroConv sum per destination type, relu, concat of all layers."""
    outs_q, outs_p = [x_q], [x_p]
    cq, cp = x_q, x_p
    for l in range(n_layers):
        g = lambda name: _f64(w[name])
        p_from_q = gat_conv(cq, cp, ei_qp, g(f"gat_qp.{l}.lin_src"), g(f"gat_qp.{l}.lin_dst"),
                            g(f"gat_qp.{l}.att_src"), g(f"gat_qp.{l}.att_dst"), g(f"gat_qp.{l}.bias"), self_loops)
        q_from_p = gat_conv(cp, cq, ei_pq, g(f"gat_pq.{l}.lin_src"), g(f"gat_pq.{l}.lin_dst"),
                            g(f"gat_pq.{l}.att_src"), g(f"gat_pq.{l}.att_dst"), g(f"gat_pq.{l}.bias"), self_loops)
        p_from_p = gated_graph_conv(cp, ei_pp, g(f"ggc.{l}.weight"), g(f"ggc.{l}.w_ih"), g(f"ggc.{l}.w_hh"),
                                    g(f"ggc.{l}.b_ih"), g(f"ggc.{l}.b_hh"), ew_pp)
        cp = np.maximum(p_from_q + p_from_p, 0.0)
        cq = np.maximum(q_from_p, 0.0)
        outs_q.append(cq)
        outs_p.append(cp)
    return np.concatenate(outs_q, axis=1), np.concatenate(outs_p, axis=1)


def pos_att_pool(node_q, node_p, q_pos, q_batch, p_cnt, p_pos, p_batch, num_graphs, w):
    """model/gnn.py:193-217, one expanded node at a time."""
    g = lambda name: _f64(w[name])
    Wq, bq, Wp, bp = g("pool.query_lin.w"), g("pool.query_lin.b"), g("pool.product_lin.w"), g("pool.product_lin.b")
    P = g("pool.pos_emb")
    Wn, bn, Wc, wa = g("pool.node_lin.w"), g("pool.node_lin.b"), g("pool.coarse_lin.w"), g("pool.att_lin.w")
    nodes, nb = [], []
    t = 0
    for i in range(node_p.shape[0]):                    # repeat_interleave(product rows, cnt)
        lin = _matvec(Wp, node_p[i]) + bp
        for _ in range(int(p_cnt[i])):
            nodes.append(np.tanh(np.concatenate([lin, P[int(p_pos[t])]])))
            nb.append(int(p_batch[i]))
            t += 1
    assert t == len(p_pos)
    for i in range(node_q.shape[0]):
        lin = _matvec(Wq, node_q[i]) + bq
        nodes.append(np.tanh(np.concatenate([lin, P[int(q_pos[i])]])))
        nb.append(int(q_batch[i]))
    D = Wn.shape[0]
    coarse = np.zeros((num_graphs, D))
    count = np.zeros(num_graphs)
    for v, b in zip(nodes, nb):
        coarse[b] += v
        count[b] += 1
    for b in range(num_graphs):
        coarse[b] /= max(count[b], 1.0)
    out = np.zeros((num_graphs, D))
    for v, b in zip(nodes, nb):
        a = _matvec(Wn, v) + bn
        c = _matvec(Wc, coarse[b])
        att = 0.0
        for o in range(D):
            att += wa[o] * _sigmoid(a[o] + c[o])
        out[b] += v * att
    for b in range(num_graphs):
        out[b] /= max(count[b], 1.0)
    return out


def encoder_forward(batch, w, n_layers, self_loops=True, get_node=False):
    """model/model.py:279-351 with table-lookup node features (DESIGN.md boundary)."""
    q, p = batch["query"], batch["product"]
    xq = _f64(w["query_table"])[_i64(q.x)]
    xp = _f64(w["item_table"])[_i64(p.x)]
    ei = batch.edge_index_dict
    from .gnn_ref import EDGE_PP, EDGE_PQ, EDGE_QP           # the three key tuples only
    nq, np_ = hetero_ggnn(xq, xp, _i64(ei[EDGE_QP]), _i64(ei[EDGE_PQ]), _i64(ei[EDGE_PP]), w, n_layers, self_loops)
    out = pos_att_pool(nq, np_, _i64(q.pos_emb_id), _i64(q.batch), _i64(p.cnt), _i64(p.pos_emb_id), _i64(p.batch),
                       int(batch.num_graphs), w)
    if get_node:
        return out, {"query": nq, "product": np_}
    return out

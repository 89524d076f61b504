"""Session action tables -> batched heterogeneous session graphs (host side, numpy).

This is the input contract of the hot path.  In the
reference a session (a list of raw action tuples) is turned into one PyG ``HeteroData`` by
``sequence_to_graph`` (reference ``util_amazon_filtered.py:98-230``) and a list of those is
collated by PyG ``Batch.from_data_list`` (through ``DataLoader``,
``test_amazon_filterd.py:488``).  Here the same *structure* is produced for a whole range
of sessions at once from a flat ``(session, action_type, item_id, query_token)`` table --
the CSV schema of the reference's ``decompose_data.py:13,30,42`` -- with vectorised numpy
group operations (no per-session Python loop), and it lands directly in the batched,
offset form the encoder reads:

* ``data['product'].x``          int64 [Np]   item ids, distinct per session
* ``data['product'].batch``      int64 [Np]   graph id of every product node
* ``data['product'].cnt``        int64 [Np]   clicks per distinct item
* ``data['product'].pos_emb_id`` int64 [sum(cnt)]  ``len(seq) - j`` per click, grouped by item
* ``data['query'].x``            int64 [Nq]   query-feature row (0 = the empty root query)
* ``data['query'].pos_emb_id``   int64 [Nq]   ``len(seq) - query_pos``
* ``data['query'].batch``        int64 [Nq]
* ``data.edge_index_dict``       COO int64 [2, E] per edge type, batch-global node ids

Text tokenisation (the reference's BERT tokenizer calls) is out of scope: query nodes carry
a feature-row id instead of token ids, see DESIGN.md.

Deliberate, documented difference from the reference: distinct items are ordered by *first
occurrence* in the session.  The reference uses ``list(set(...))`` (hash order,
``util_amazon_filtered.py:128``) which is not reproducible across processes.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Tuple

import numpy as np

EDGE_QP = ("query", "clicks", "product")
EDGE_PQ = ("product", "clicked by", "query")
EDGE_PP = ("product", "to", "product")

ASIN_NUM = 391572          # reference test_amazon_filterd.py:457
QUERY_VOCAB = 4097         # row 0 = root/empty query + 4096 synthetic query features
MAX_SEQ_LEN = 20           # reference config.py:5


@dataclass
class ActionTable:
    """Flat table of actions, sessions stored contiguously in order.

    ``sess_ptr[s]:sess_ptr[s+1]`` are the actions of session ``s`` in time order.
    ``is_search[t]`` is True for a search action ('s' in the reference's raw tuples,
    ``util_amazon_filtered.py:11``), otherwise the action is an item click.
    """
    sess_ptr: np.ndarray    # int64 [S+1]
    is_search: np.ndarray   # bool  [T]
    item_id: np.ndarray     # int64 [T]  (0 where is_search)
    query_tok: np.ndarray   # int64 [T]  (0 where not is_search)

    @property
    def num_sessions(self) -> int:
        return int(self.sess_ptr.shape[0] - 1)

    def slice(self, lo: int, hi: int) -> "ActionTable":
        a, b = int(self.sess_ptr[lo]), int(self.sess_ptr[hi])
        return ActionTable(self.sess_ptr[lo:hi + 1] - a, self.is_search[a:b],
                           self.item_id[a:b], self.query_tok[a:b])

    def prefix(self, frac_num: int, frac_den: int) -> "ActionTable":
        """Prefix sub-sessions: keep the first ceil(len*num/den) (at least 1) actions.

        Deterministic stand-in for the random cut of the reference's ``to_subsession``
        (``train_subsession_embedding.py:41``); in the retrieval pipeline a sub-session is
        the prefix ``seq`` of a ``(seq, tar)`` split (``test_amazon_filterd.py:546``).
        """
        ln = np.diff(self.sess_ptr)
        keep = np.maximum(1, -(-ln * frac_num // frac_den))
        keep = np.minimum(keep, ln)
        t_sess = np.repeat(np.arange(self.num_sessions), ln)
        j = np.arange(self.is_search.shape[0]) - self.sess_ptr[:-1][t_sess]
        m = j < keep[t_sess]
        ptr = np.zeros(self.num_sessions + 1, np.int64)
        np.cumsum(keep, out=ptr[1:])
        return ActionTable(ptr, self.is_search[m], self.item_id[m], self.query_tok[m])


def synthetic_actions(n_sessions: int, seed: int, n_items: int = ASIN_NUM,
                      n_query: int = QUERY_VOCAB) -> ActionTable:
    """Seeded synthetic sessions of the shape SURVEY.md section 8(d) names.

    actions per session  A = clip(2 + Poisson(6), 2, 19)  (<= 19 so every pos_emb_id < 20);
    each action is a search with p = 0.3, otherwise a click on item
    ``1 + (Zipf(1.2) mod (V-1))`` (id 0 is the reserved "unknown" item,
    ``util_amazon_filtered.py:133``).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    ln = np.clip(2 + rng.poisson(6.0, n_sessions), 2, MAX_SEQ_LEN - 1).astype(np.int64)
    ptr = np.zeros(n_sessions + 1, np.int64)
    np.cumsum(ln, out=ptr[1:])
    T = int(ptr[-1])
    is_search = rng.random(T) < 0.3
    item = 1 + (rng.zipf(1.2, T).astype(np.int64) % (n_items - 1))
    item[is_search] = 0
    qtok = 1 + rng.integers(0, n_query - 1, T, dtype=np.int64)
    qtok[~is_search] = 0
    return ActionTable(ptr, is_search, item, qtok)


class NodeStore:
    """Attribute bag for one node type (stands in for a PyG node storage)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def keys(self):
        return list(self.__dict__.keys())


@dataclass
class SessionBatch:
    """Batched session graphs; duck-types the attributes of a PyG hetero ``Batch`` that the
    reference encoder reads (``model/model.py:279-351``, ``model/gnn.py:193-217``)."""
    nodes: Dict[str, NodeStore]
    edge_index_dict: Dict[Tuple[str, str, str], object]
    edge_weight_dict: Dict[Tuple[str, str, str], object]
    num_graphs: int
    extras: dict = field(default_factory=dict)

    def __getitem__(self, key):
        return self.nodes[key]

    def _map(self, fn):
        nodes = {k: NodeStore(**{a: fn(v) for a, v in s.__dict__.items()})
                 for k, s in self.nodes.items()}
        ei = {k: fn(v) for k, v in self.edge_index_dict.items()}
        ew = {k: (None if v is None else fn(v)) for k, v in self.edge_weight_dict.items()}
        return SessionBatch(nodes, ei, ew, self.num_graphs, dict(self.extras))

    def to_torch(self, device="cpu"):
        import torch

        def f(v):
            if isinstance(v, np.ndarray):
                return torch.from_numpy(np.ascontiguousarray(v)).to(device)
            if isinstance(v, torch.Tensor):
                return v.to(device)
            return v
        return self._map(f)

    def to(self, device):
        return self.to_torch(device)

    def to_numpy(self):
        def f(v):
            if hasattr(v, "detach"):
                return v.detach().cpu().numpy()
            return v
        return self._map(f)


def _group_rank(sorted_group: np.ndarray) -> np.ndarray:
    """Rank 0,1,2,.. of each element inside its (already contiguous) group."""
    n = sorted_group.shape[0]
    if n == 0:
        return np.zeros(0, np.int64)
    start = np.r_[True, sorted_group[1:] != sorted_group[:-1]]
    first = np.flatnonzero(start)
    return np.arange(n, dtype=np.int64) - np.repeat(first, np.diff(np.r_[first, n]))


def build_batch(actions: ActionTable) -> SessionBatch:
    """All sessions of ``actions`` as one batch (structure of ``sequence_to_graph`` +
    ``Batch.from_data_list``; see the module docstring for the field list)."""
    S = actions.num_sessions
    ptr = actions.sess_ptr
    ln = np.diff(ptr)                                   # len(seq) per session
    T = int(ptr[-1])
    sess = np.repeat(np.arange(S, dtype=np.int64), ln)  # session of each action
    j = np.arange(T, dtype=np.int64) - ptr[:-1][sess]   # index inside the session
    srch = actions.is_search
    clk = ~srch

    # ---- query nodes: a root per session + one per search (util_amazon_filtered.py:7-22)
    n_search = np.bincount(sess[srch], minlength=S).astype(np.int64)
    nq_per = 1 + n_search
    q_off = np.zeros(S + 1, np.int64)
    np.cumsum(nq_per, out=q_off[1:])
    Nq = int(q_off[-1])
    q_x = np.zeros(Nq, np.int64)
    q_pos = np.zeros(Nq, np.int64)
    q_batch = np.repeat(np.arange(S, dtype=np.int64), nq_per)
    q_pos[q_off[:-1]] = ln                               # root: len(seq) - 0
    # searches so far (inclusive) inside each session == local index of that query node
    cs = np.cumsum(srch.astype(np.int64))
    base = np.r_[0, cs][ptr[:-1]]                        # searches before the session
    local_q = cs - base[sess]                            # for a search: its node id (1..)
    s_idx = np.flatnonzero(srch)
    gq = q_off[:-1][sess[s_idx]] + local_q[s_idx]
    q_x[gq] = actions.query_tok[s_idx]
    q_pos[gq] = ln[sess[s_idx]] - (j[s_idx] + 1)         # len(seq) - query_pos

    # ---- product nodes: distinct items per session, first-occurrence order
    c_idx = np.flatnonzero(clk)
    c_sess = sess[c_idx]
    c_item = actions.item_id[c_idx]
    n_click = np.bincount(c_sess, minlength=S).astype(np.int64)
    # group clicks by (session, item); stable so occurrences stay in time order
    order = np.lexsort((c_idx, c_item, c_sess))
    so_sess, so_item = c_sess[order], c_item[order]
    newgrp = np.r_[True, (so_sess[1:] != so_sess[:-1]) | (so_item[1:] != so_item[:-1])] \
        if order.size else np.zeros(0, bool)
    grp_of_sorted = np.cumsum(newgrp) - 1                 # group id per sorted click
    g_first = np.flatnonzero(newgrp)                      # first (earliest) click per group
    G = g_first.shape[0]
    g_sess = so_sess[g_first]
    g_item = so_item[g_first]
    g_firstpos = c_idx[order][g_first]                    # action index of first occurrence
    g_cnt = np.diff(np.r_[g_first, order.size]).astype(np.int64)
    # order the groups of each session by first occurrence -> local product index
    gord = np.lexsort((g_firstpos, g_sess))
    g_local = np.empty(G, np.int64)
    g_local[gord] = _group_rank(g_sess[gord])
    empty = n_click == 0                                  # sessions without any click
    np_per = np.bincount(g_sess, minlength=S).astype(np.int64) + empty
    p_off = np.zeros(S + 1, np.int64)
    np.cumsum(np_per, out=p_off[1:])
    Np = int(p_off[-1])
    p_x = np.zeros(Np, np.int64)                          # unknown item 0 for empty sessions
    p_cnt = np.ones(Np, np.int64)
    p_batch = np.repeat(np.arange(S, dtype=np.int64), np_per)
    g_node = p_off[:-1][g_sess] + g_local                 # batch-global product node id
    p_x[g_node] = g_item
    p_cnt[g_node] = g_cnt
    # pos_emb_id: per product node (in node order), per occurrence in time order
    click_node_sorted = g_node[grp_of_sorted]             # node of each sorted click
    c_j = j[c_idx][order]
    c_len = ln[so_sess]
    o2 = np.lexsort((c_j, click_node_sorted))
    pos_clicks = (c_len - c_j)[o2]
    pos_node = click_node_sorted[o2]
    # empty sessions contribute one pos id 0 at their single node (util_..:132-135)
    e_nodes = p_off[:-1][empty]
    if e_nodes.size:
        allnode = np.r_[pos_node, e_nodes]
        allpos = np.r_[pos_clicks, np.zeros(e_nodes.size, np.int64)]
        o3 = np.argsort(allnode, kind="stable")
        p_pos = allpos[o3]
    else:
        p_pos = pos_clicks

    # ---- click edges (util_amazon_filtered.py:180-195), in action order
    node_of_click = np.empty(c_idx.size, np.int64)
    node_of_click[order] = click_node_sorted
    e_from = q_off[:-1][c_sess] + local_q[c_idx]          # most recent query node
    e_to = node_of_click
    ei_qp = np.stack([e_from, e_to]).astype(np.int64)
    ei_pq = np.stack([e_to, e_from]).astype(np.int64)

    # ---- item->item transitions, de-duplicated with counts (util_..:199-218)
    if c_idx.size > 1:
        same = c_sess[1:] == c_sess[:-1]
        t_from = node_of_click[:-1][same]
        t_to = node_of_click[1:][same]
        key = t_from * np.int64(Np) + t_to
        uk, first, cnt = np.unique(key, return_index=True, return_counts=True)
        o = np.argsort(first, kind="stable")              # keep first-occurrence order
        ei_pp = np.stack([t_from[first[o]], t_to[first[o]]]).astype(np.int64)
        w_pp = cnt[o].astype(np.float32)
    else:
        ei_pp = np.zeros((2, 0), np.int64)
        w_pp = np.zeros(0, np.float32)

    nodes = {
        "product": NodeStore(x=p_x, batch=p_batch, cnt=p_cnt, pos_emb_id=p_pos.astype(np.int64)),
        "query": NodeStore(x=q_x, batch=q_batch, pos_emb_id=q_pos),
    }
    ei = {EDGE_QP: ei_qp, EDGE_PQ: ei_pq, EDGE_PP: ei_pp}
    ew = {EDGE_QP: None, EDGE_PQ: None, EDGE_PP: w_pp}
    return SessionBatch(nodes, ei, ew, S)

"""Oracle: per-session graph construction + batch collation.  TEST INFRASTRUCTURE ONLY.

Plain-Python restatement (one session at a time, loops as in the reference) of the
*structural* part of ``sequence_to_graph`` (reference ``util_amazon_filtered.py:98-230``) and
of PyG ``Batch.from_data_list``.  It checks the vectorised builder
``sessionsimilaritysearch_amd.sessions.build_batch``.

A session is a list of ``(is_search: bool, item_id: int, query_tok: int)``.

Difference from the reference, on purpose: distinct items are kept in first-occurrence order
(the reference's ``list(set(...))`` at ``util_amazon_filtered.py:128`` is hash-ordered).
"""
from __future__ import annotations

import numpy as np


def session_to_graph(seq):
    n = len(seq)
    # query nodes -- get_query_node_tokens, util_amazon_filtered.py:7-22
    q_x, q_pos = [0], [0]
    for i, (is_s, _item, tok) in enumerate(seq):
        if not is_s:
            continue
        q_x.append(tok)
        q_pos.append(i + 1)
    q_pos_emb = [n - p for p in q_pos]
    # distinct items, first-occurrence order -- :128 (see module docstring)
    distinct = []
    for is_s, item, _ in seq:
        if not is_s and item not in distinct:
            distinct.append(item)
    # get_item_pos_cnt, :77-85
    pos_ids, cnt = [], [0] * len(distinct)
    for i, item in enumerate(distinct):
        for j, (is_s, it, _) in enumerate(seq):
            if not is_s and it == item:
                cnt[i] += 1
                pos_ids.append(n - j)
    if len(distinct) == 0:                  # :132-135
        distinct, cnt, pos_ids = [0], [1], [0]
    pos = {it: i for i, it in enumerate(distinct)}
    # click edges -- :180-195
    last_q, e_from, e_to = 0, [], []
    for is_s, item, _ in seq:
        if is_s:
            last_q += 1
            continue
        e_from.append(last_q)
        e_to.append(pos[item])
    # transitions -- :199-218
    item_seq = [it for is_s, it, _ in seq if not is_s]
    t_from, t_to, w, seen = [], [], [], {}
    for i in range(len(item_seq) - 1):
        k = (pos[item_seq[i]], pos[item_seq[i + 1]])
        if k not in seen:
            seen[k] = len(t_from)
            t_from.append(k[0]); t_to.append(k[1]); w.append(1)
        else:
            w[seen[k]] += 1
    return dict(q_x=q_x, q_pos=q_pos_emb, p_x=distinct, p_cnt=cnt, p_pos=pos_ids,
                qp=(e_from, e_to), pp=(t_from, t_to), pp_w=w)


def collate(graphs):
    """``Batch.from_data_list`` semantics: concatenate per node type, offset edge indices by
    the cumulative node count of the source (row 0) / target (row 1) type, add ``batch``."""
    out = {k: [] for k in ("q_x", "q_pos", "q_batch", "p_x", "p_cnt", "p_pos", "p_batch",
                           "qp0", "qp1", "pp0", "pp1", "pp_w")}
    qo = po = 0
    for g, d in enumerate(graphs):
        out["q_x"] += d["q_x"]; out["q_pos"] += d["q_pos"]; out["q_batch"] += [g] * len(d["q_x"])
        out["p_x"] += d["p_x"]; out["p_cnt"] += d["p_cnt"]; out["p_pos"] += d["p_pos"]
        out["p_batch"] += [g] * len(d["p_x"])
        out["qp0"] += [qo + a for a in d["qp"][0]]; out["qp1"] += [po + b for b in d["qp"][1]]
        out["pp0"] += [po + a for a in d["pp"][0]]; out["pp1"] += [po + b for b in d["pp"][1]]
        out["pp_w"] += d["pp_w"]
        qo += len(d["q_x"]); po += len(d["p_x"])
    r = {k: np.asarray(v, dtype=np.int64) for k, v in out.items() if k != "pp_w"}
    r["pp_w"] = np.asarray(out["pp_w"], dtype=np.float32)
    return r


def actions_to_sessions(actions):
    """ActionTable -> list of python sessions (for the small oracle cases)."""
    res = []
    for s in range(actions.num_sessions):
        a, b = int(actions.sess_ptr[s]), int(actions.sess_ptr[s + 1])
        res.append([(bool(actions.is_search[t]), int(actions.item_id[t]), int(actions.query_tok[t]))
                    for t in range(a, b)])
    return res

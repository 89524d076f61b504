"""CPU tests (no GPU): the oracle against the reference's known answers, the host-side batch
builder against the oracle's per-session restatement, and the C-ABI library's symbol table."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import graph_ref, gnn_ref, search_ref as sr
from sessionsimilaritysearch_amd import sessions as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_normalize_known_answer():
    # the one known-answer value the reference holds: print(normalize(np.ones(4))) -> [0.5 0.5 0.5 0.5]
    # (test_amazon_filterd.py:866)
    assert np.array_equal(sr.normalize(np.ones(4)), np.full(4, 0.5))
    z = sr.normalize(np.zeros((2, 8), np.float32))
    assert np.isfinite(z).all() and (z == 0).all()
    x = np.random.default_rng(0).standard_normal((5, 16)).astype(np.float32)
    assert sr.normalize(x).dtype == np.float32                      # float32 in -> float32 out
    np.testing.assert_allclose(np.linalg.norm(sr.normalize(x), axis=1), 1.0, rtol=1e-6)


def test_golden_node_asin_embedding():
    """NodeAsinEmbedding outputs generated from the reference's own model/NodeEmbedding.py
    (tests/golden/make_golden.py) pin the oracle's lookup."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "node_asin_embedding.npz"), allow_pickle=False)
    out = gnn_ref.embedding_lookup(torch.from_numpy(z["table"]), torch.from_numpy(z["ids"]))
    assert np.array_equal(out.numpy(), z["out"])


def test_exact_search_c_matches_numpy_and_tie_rule():
    rng = np.random.default_rng(1)
    q = sr.normalize(rng.standard_normal((17, 64)).astype(np.float32))
    c = sr.normalize(rng.standard_normal((500, 64)).astype(np.float32))
    c[100] = c[7]; c[300] = c[7]                     # exact duplicates tie -> ascending id
    D1, I1 = sr.search_exact(q, c, 12)
    D2, I2 = sr.search_exact_numpy(q, c, 12)
    assert np.array_equal(I1, I2) and np.array_equal(D1, D2)
    for row in I1:
        pos = {int(v): i for i, v in enumerate(row)}
        if 7 in pos and 100 in pos:
            assert pos[7] < pos[100]
    # faiss-shaped fp32 search agrees up to near-ties
    D3, I3 = sr.search_fp32_blocked(q, c, 12, block=128)
    assert sr.recall_at_k(I3, I1, 12) > 0.99 and np.abs(D3 - D1).max() < 1e-5
    # padding when n < k
    D4, I4 = sr.search_exact(q[:2], c[:5], 8)
    assert (I4[:, 5:] == -1).all() and (D4[:, 5:] == sr.NEG_SENTINEL).all()


def test_merge_topk_equals_unsharded():
    rng = np.random.default_rng(2)
    q = rng.standard_normal((9, 32)).astype(np.float32)
    c = rng.standard_normal((400, 32)).astype(np.float32)
    D, I = sr.search_exact(q, c, 10)
    parts = [sr.search_exact(q, c[lo:lo + 100], 10, id_offset=lo) for lo in range(0, 400, 100)]
    Dm, Im = sr.merge_topk([p[0] for p in parts], [p[1] for p in parts], 10)
    assert np.array_equal(Im, I) and np.array_equal(Dm, D)


def test_build_index_metrics():
    rng = np.random.default_rng(3)
    emb = rng.standard_normal((50, 16)).astype(np.float32)
    for m in ("cos", "ip", "l2"):
        idx = sr.build_index(emb, m)
        D, I = idx.search(emb[:4], 3)
        assert (I[:, 0] == np.arange(4)).all()        # every vector's nearest neighbour is itself
    with pytest.raises(RuntimeError):
        sr.build_index(emb, "nope")


def test_knn_vote_and_pr():
    D = np.array([0.9, 0.5, 0.4], np.float32)
    I = np.array([2, 0, 1])
    items = [np.array([10, 11]), np.array([11]), np.array([12, 10])]
    # item 10: 0.9 + 0.5, item 12: 0.9, item 11: 0.5 + 0.4
    assert sr.knn_item_vote(D, I, items, 2) == [10, 12] or sr.knn_item_vote(D, I, items, 3)[0] == 10
    assert sr.knn_item_vote(D, I, items, 3) == [10, 12, 11] or sr.knn_item_vote(D, I, items, 3) == [10, 11, 12]
    p, r = sr.get_p_r({10, 99}, [10, 12, 11], 2)
    assert p == 0.5 and r == 0.5


@pytest.mark.parametrize("seed,n", [(0, 1), (1, 7), (2, 200)])
def test_build_batch_matches_per_session_oracle(seed, n):
    acts = S.synthetic_actions(n, seed, n_items=50, n_query=9)    # small vocab -> many repeats
    b = S.build_batch(acts)
    ref = graph_ref.collate([graph_ref.session_to_graph(s) for s in graph_ref.actions_to_sessions(acts)])
    assert np.array_equal(b["query"].x, ref["q_x"])
    assert np.array_equal(b["query"].pos_emb_id, ref["q_pos"])
    assert np.array_equal(b["query"].batch, ref["q_batch"])
    assert np.array_equal(b["product"].x, ref["p_x"])
    assert np.array_equal(b["product"].cnt, ref["p_cnt"])
    assert np.array_equal(b["product"].pos_emb_id, ref["p_pos"])
    assert np.array_equal(b["product"].batch, ref["p_batch"])
    assert np.array_equal(b.edge_index_dict[S.EDGE_QP], np.stack([ref["qp0"], ref["qp1"]]))
    assert np.array_equal(b.edge_index_dict[S.EDGE_PQ], np.stack([ref["qp1"], ref["qp0"]]))
    assert np.array_equal(b.edge_index_dict[S.EDGE_PP], np.stack([ref["pp0"], ref["pp1"]]))
    assert np.array_equal(b.edge_weight_dict[S.EDGE_PP], ref["pp_w"])
    assert b["product"].pos_emb_id.max() < S.MAX_SEQ_LEN and b["query"].pos_emb_id.max() < S.MAX_SEQ_LEN
    assert b["product"].cnt.sum() == b["product"].pos_emb_id.shape[0]


def test_build_batch_edge_cases():
    # a session of searches only gets the single "unknown item" node (util_amazon_filtered.py:132-135)
    acts = S.ActionTable(np.array([0, 2, 5]), np.array([1, 1, 0, 0, 1], bool),
                         np.array([0, 0, 5, 5, 0]), np.array([3, 4, 0, 0, 2]))
    b = S.build_batch(acts)
    assert b["product"].x.tolist() == [0, 5] and b["product"].cnt.tolist() == [1, 2]
    assert b["product"].pos_emb_id.tolist() == [0, 3, 2]
    assert b["query"].x.tolist() == [0, 3, 4, 0, 2]
    assert b.edge_index_dict[S.EDGE_PP].tolist() == [[1], [1]]     # 5 -> 5 self transition, weight 1
    ref = graph_ref.collate([graph_ref.session_to_graph(s) for s in graph_ref.actions_to_sessions(acts)])
    assert np.array_equal(b.edge_index_dict[S.EDGE_QP], np.stack([ref["qp0"], ref["qp1"]]))


def test_prefix_subsessions():
    acts = S.synthetic_actions(50, 4)
    half = acts.prefix(1, 2)
    ln, lh = np.diff(acts.sess_ptr), np.diff(half.sess_ptr)
    assert (lh == np.maximum(1, -(-ln // 2))).all()
    a0, h0 = int(acts.sess_ptr[3]), int(half.sess_ptr[3])
    assert np.array_equal(acts.item_id[a0:a0 + lh[3]], half.item_id[h0:h0 + lh[3]])
    full = acts.prefix(1, 1)
    assert np.array_equal(full.item_id, acts.item_id)


def test_gnn_oracle_invariants():
    """Structural properties of the restatement: softmax weights of a target sum to 1, a target
    with no incoming edge gets the bias, the GRU part equals torch.nn.GRUCell, pooling is
    invariant to the order of a graph's nodes."""
    torch.manual_seed(0)
    xs, xd = torch.randn(5, 8), torch.randn(4, 8)
    ls, ld = torch.randn(6, 8), torch.randn(6, 8)
    a_s, a_d, b = torch.randn(6), torch.randn(6), torch.randn(6)
    ei = torch.tensor([[0, 1, 1, 4], [0, 0, 2, 2]])
    out = gnn_ref.gat_conv(xs, xd, ei, ls, ld, a_s, a_d, b, self_loops=False)
    assert torch.allclose(out[1], b) and torch.allclose(out[3], b)
    # with the bipartite rewrite: edge (0 -> 0) is replaced by the synthetic loop, loops 0..3 added
    rw = gnn_ref.rewrite_self_loops(ei, 5, 4)
    assert rw.tolist() == [[1, 1, 4, 0, 1, 2, 3], [0, 2, 2, 0, 1, 2, 3]]
    x = torch.randn(7, 8)
    W, wih, whh, bih, bhh = torch.randn(8, 8), torch.randn(24, 8), torch.randn(24, 8), torch.randn(24), torch.randn(24)
    o = gnn_ref.gated_graph_conv(x, torch.zeros((2, 0), dtype=torch.long), W, wih, whh, bih, bhh)
    cell = torch.nn.GRUCell(8, 8)
    with torch.no_grad():
        cell.weight_ih.copy_(wih); cell.weight_hh.copy_(whh); cell.bias_ih.copy_(bih); cell.bias_hh.copy_(bhh)
        assert torch.allclose(o, cell(torch.zeros(7, 8), x))
    with pytest.raises(ValueError):
        gnn_ref.gated_graph_conv(torch.randn(3, 9), torch.zeros((2, 0), dtype=torch.long), W, wih, whh, bih, bhh)


def test_c_abi_library_exports_every_declared_symbol():
    """include/sss.h <-> libsss.so <-> the ctypes table agree (no compute call: no GPU here)."""
    import sessionsimilaritysearch_amd as pkg
    hdr = open(os.path.join(ROOT, "include", "sss.h")).read()
    declared = set(re.findall(r"\b(sss_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(pkg.exported_symbols())
    L = pkg.lib()                                   # raises if the .so is missing or lacks a symbol
    assert L.sss_version() >= 100
    for name in declared:
        assert hasattr(L, name)


def test_product_path_never_imports_the_oracle():
    pk = os.path.join(ROOT, "sessionsimilaritysearch_amd")
    for fn in os.listdir(pk):
        if fn.endswith(".py"):
            src = open(os.path.join(pk, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn

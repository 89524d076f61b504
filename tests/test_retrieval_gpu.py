"""GPU parity: neighbour-weighted item vote (sss_knn_item_vote) and the prefix sub-session
pipeline of BASELINE config C3, against the oracle's restatement of the reference's
``get_prediction_by_knn`` (test_amazon_filterd.py:59-78).  Integer results (item ids) are compared
bit for bit; the float64 weights too (same additions in the same order)."""
import numpy as np
import pytest
import torch

from oracle import search_ref as sr
from sessionsimilaritysearch_amd import sessions as S

pytestmark = pytest.mark.gpu


def _items_of(batch):
    x, gb = np.asarray(batch["product"].x), np.asarray(batch["product"].batch)
    return [x[gb == g] for g in range(batch.num_graphs)]


def _check_vote(cuda, D, I, batch, K):
    from sessionsimilaritysearch_amd.retrieval import SessionItems, knn_item_vote
    ds = SessionItems.from_batch(batch, cuda)
    out, wts, status = knn_item_vote(torch.from_numpy(D).to(cuda), torch.from_numpy(I).to(cuda), ds, K, return_weights=True)
    assert int(status.sum().item()) == 0
    out, wts = out.cpu().numpy(), wts.cpu().numpy()
    items = _items_of(batch)
    for r in range(D.shape[0]):
        ref_i, ref_w = sr.knn_item_vote_weights(D[r], I[r], items, K)
        got = [int(v) for v in out[r] if v >= 0]
        assert got == ref_i, (r, got, ref_i)
        assert wts[r, :len(ref_w)].tolist() == ref_w
        assert (out[r, len(ref_i):] == -1).all()


def test_vote_matches_oracle_small_vocab_many_ties(cuda):
    """Tiny vocabulary -> many items shared between neighbours and many exactly tied weights
    (equal D values): exercises the float64 order of additions and the first-seen tie rule."""
    rng = np.random.default_rng(50)
    batch = S.build_batch(S.synthetic_actions(300, 50, n_items=40, n_query=9))
    nq, Sn = 33, 64
    I = rng.integers(0, 300, (nq, Sn)).astype(np.int64)
    D = np.round(rng.uniform(0.2, 1.0, (nq, Sn)), 1).astype(np.float32)     # few distinct values -> ties
    D = -np.sort(-D, axis=1)
    I[3, 10:] = -1                                                          # padded result rows are skipped
    I[4, :] = -1
    _check_vote(cuda, D, I, batch, 10)
    _check_vote(cuda, D, I, batch, 20)                                      # reference main() uses K = 20


def test_vote_sample_size_500(cuda):
    rng = np.random.default_rng(51)
    batch = S.build_batch(S.synthetic_actions(5000, 51, n_items=3000, n_query=65))
    nq, Sn = 40, 500
    I = np.stack([rng.permutation(5000)[:Sn] for _ in range(nq)]).astype(np.int64)
    D = -np.sort(-rng.uniform(0.3, 0.99, (nq, Sn)).astype(np.float32), axis=1)
    _check_vote(cuda, D, I, batch, 10)


def test_vote_large_samples_take_the_second_launch(cuda):
    """More expanded (neighbour, item) pairs than the first launch's 4096-entry footprint (sample size 1500, ~5 items a
    session: ~7500 pairs) next to queries that fit it (most of their neighbours padded away): both launches, one result;
    and a sample too large for the full 16384-entry capacity reports status 1."""
    from sessionsimilaritysearch_amd.retrieval import SessionItems, knn_item_vote
    rng = np.random.default_rng(52)
    batch = S.build_batch(S.synthetic_actions(6000, 52, n_items=2000, n_query=65))
    nq, Sn = 24, 1500
    I = np.stack([rng.permutation(6000)[:Sn] for _ in range(nq)]).astype(np.int64)
    D = -np.sort(-rng.uniform(0.3, 0.99, (nq, Sn)).astype(np.float32), axis=1)
    I[::3, 400:] = -1                                                       # every third query: 400 neighbours (~2000 pairs)
    _check_vote(cuda, D, I, batch, 10)
    Sn = 6000                                                               # ~30 000 pairs: beyond the capacity
    I = np.stack([rng.permutation(6000)[:Sn] for _ in range(4)]).astype(np.int64)
    D = -np.sort(-rng.uniform(0.3, 0.99, (4, Sn)).astype(np.float32), axis=1)
    I[1, 300:] = -1
    ds = SessionItems.from_batch(batch, cuda)
    res = knn_item_vote(torch.from_numpy(D).to(cuda), torch.from_numpy(I).to(cuda), ds, 10, return_weights=True)
    out, status = res[0].cpu().numpy(), res[2].cpu().numpy()
    assert status.tolist() == [1, 0, 1, 1] and (out[0] == -1).all()
    ref_i, _ = sr.knn_item_vote_weights(D[1], I[1], _items_of(batch), 10)
    assert [int(v) for v in out[1]] == ref_i


def test_config_c3_pipeline_small(cuda):
    """Config C3 in miniature: corpus = 4 prefix sub-sessions (25/50/75/100 % of the actions) of
    every session, embedded by the HIP encoder; query = the 50 % prefix; top-500 neighbours ->
    item vote -> top-10 items.  Checked against the oracle run on the same embeddings."""
    from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, init_weights
    from sessionsimilaritysearch_amd.index import FlatIndex
    from sessionsimilaritysearch_amd.retrieval import SessionItems, get_p_r, get_prediction_by_knn
    cfg = EncoderConfig(d_in=64, h=64, n_layers=2, d_out=64, n_items=4000, n_query=129, self_loop_rule="none")
    enc = SessionEncoder(cfg, init_weights(cfg, 52), cuda)
    acts = S.synthetic_actions(1500, 52, cfg.n_items, cfg.n_query)
    batches = [S.build_batch(acts.prefix(f, 4)) for f in (1, 2, 3, 4)]
    index = FlatIndex(64, "ip", cuda)
    for b in batches:
        index.add(enc(b.to(cuda), l2_normalize=True))
    assert index.ntotal == 6000
    ds = SessionItems.from_batch(batches, cuda)
    q = enc(S.build_batch(acts.slice(0, 64).prefix(1, 2)).to(cuda), l2_normalize=True)
    pred = get_prediction_by_knn(q, index, ds, 500, 10).cpu().numpy()
    # oracle: exact search on the same stored vectors + the reference's vote
    Dr, Ir = sr.search_exact(q.cpu().numpy(), index._xb.cpu().numpy(), 500)
    items = sum((_items_of(b) for b in batches), [])
    for r in range(64):
        ref = sr.knn_item_vote(Dr[r], Ir[r], items, 10)
        assert [int(v) for v in pred[r] if v >= 0] == ref
    # single 1-D query form returns a python list, as the reference does
    one = get_prediction_by_knn(q[0], index, ds, 500, 10)
    assert one == [int(v) for v in pred[0] if v >= 0]
    gt = set(int(v) for v in items[4500 + 0])                             # the full session's items
    p, r = get_p_r(gt, one, 10)
    assert 0.0 <= p <= 1.0 and 0.0 <= r <= 1.0


def test_vote_and_p_r_equal_the_reference_run_vectors(cuda):
    """`sss_knn_item_vote` and `retrieval.get_p_r` against tests/golden/reference_pure.npz: item lists the reference's
    OWN `get_prediction_by_knn` (test_amazon_filterd.py:59-78) produced for these (D, I, session items) inputs in the
    build container (tests/golden/make_golden_pure.py), incl. exact weight ties and fewer items than K."""
    import os
    from sessionsimilaritysearch_amd.retrieval import SessionItems, get_p_r, knn_item_vote
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_pure.npz"), allow_pickle=False)
    for tag in z["vote_tags"]:
        ds = SessionItems(torch.from_numpy(z[f"vote_{tag}_ptr"]).to(cuda), torch.from_numpy(z[f"vote_{tag}_items"].astype(np.int32)).to(cuda))
        D = torch.from_numpy(z[f"vote_{tag}_D"][None, :]).to(cuda)
        I = torch.from_numpy(z[f"vote_{tag}_I"][None, :]).to(cuda)
        K = int(z[f"vote_{tag}_K"])
        out, status = knn_item_vote(D, I, ds, K)
        assert int(status.sum().item()) == 0
        got = [int(v) for v in out[0].cpu().numpy() if v >= 0]
        assert got == z[f"vote_{tag}_pred"].tolist(), tag
        assert (out[0, len(got):] == -1).all()
    for i in range(int(z["pr_cases"])):
        gt, pred, K = set(z[f"pr{i}_gt"].tolist()), z[f"pr{i}_pred"].tolist(), int(z[f"pr{i}_K"])
        assert list(get_p_r(gt, pred, K)) == z[f"pr{i}_out"].tolist()

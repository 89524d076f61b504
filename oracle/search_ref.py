"""Oracle: normalise, flat inner-product index/search, shard merge, neighbour item vote.
TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  PARITY UNPINNED except
``normalize(np.ones(4))`` (reference test_amazon_filterd.py:866); faiss is absent, its
``IndexFlatIP`` semantics follow SURVEY.md Appendix A.5.

The canonical result contract (DESIGN.md "exactness"): the score of (query a, corpus row b)
is the dot product accumulated sequentially over k = 0..d-1 in float64 (every float32
product is exact in float64) and rounded once to float32; the top-k is ordered by
(score descending, id ascending).  faiss' own tie order is implementation-defined; this is
the rule the build fixes (SURVEY.md hard part H1).
"""
from __future__ import annotations

import ctypes
import os
from collections import defaultdict

import numpy as np

NEG_SENTINEL = np.float32(-3.4028234663852886e38)   # faiss heap neutral for IP (-FLT_MAX)
_HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- normalize
def normalize(vec):
    """Reference ``normalize`` verbatim in behaviour (util_amazon_filtered.py:28-31)."""
    if len(vec.shape) == 1:
        return vec / np.sqrt(np.clip(np.sum(vec ** 2), 1e-6, None))
    return vec / np.sqrt(np.clip(np.sum(vec ** 2, axis=1), 1e-6, None)).reshape(-1, 1)


def normalize_norm_eps(v):
    """The fine-tune scripts' local variant (fine_tune_ours.py:38-40): v / (||v|| + 1e-4)."""
    norm = np.linalg.norm(v, axis=1) + 1e-4
    return v / np.expand_dims(norm, -1)


# ------------------------------------------------------------------- canonical exact search
def canonical_scores(q, c):
    """float32 [nq, n]: sequential-k float64 dot product, rounded once (numpy, small sizes)."""
    q64 = np.asarray(q, np.float32).astype(np.float64)
    c64 = np.asarray(c, np.float32).astype(np.float64)
    acc = np.zeros((q64.shape[0], c64.shape[0]), np.float64)
    for k in range(q64.shape[1]):
        acc += q64[:, k:k + 1] * c64[:, k][None, :]
    return acc.astype(np.float32)


def canonical_l2(q, c):
    """float32 [nq, n]: sum_k (q_k - c_k)^2 sequentially in float64 (IndexFlatL2 = squared L2)."""
    q64 = np.asarray(q, np.float32).astype(np.float64)
    c64 = np.asarray(c, np.float32).astype(np.float64)
    acc = np.zeros((q64.shape[0], c64.shape[0]), np.float64)
    for k in range(q64.shape[1]):
        dlt = q64[:, k:k + 1] - c64[:, k][None, :]
        acc += dlt * dlt
    return acc.astype(np.float32)


def topk_from_scores(scores, k, id_offset=0, largest=True):
    """Rows of ``scores`` -> (D float32 [nq,k], I int64 [nq,k]); (score desc, id asc), or
    (distance asc, id asc) when ``largest`` is False.  Missing results: I = -1, D = sentinel
    (faiss IndexFlat semantics, Appendix A.5)."""
    nq, n = scores.shape
    key = -scores if largest else scores
    ids = np.broadcast_to(np.arange(n, dtype=np.int64), (nq, n))
    order = np.lexsort((ids, key), axis=1)[:, :k]
    kk = order.shape[1]
    D = np.full((nq, k), NEG_SENTINEL if largest else -NEG_SENTINEL, np.float32)
    I = np.full((nq, k), -1, np.int64)
    D[:, :kk] = np.take_along_axis(scores, order, axis=1)
    I[:, :kk] = order + id_offset
    return D, I


def search_exact_numpy(q, c, k, id_offset=0):
    return topk_from_scores(canonical_scores(q, c), k, id_offset)


_lib = None


def _load_c():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libsss_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle C library missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = ctypes.CDLL(path)
        _lib.oracle_search_exact.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                             ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_int]
        _lib.oracle_search_exact.restype = ctypes.c_int
        _lib.oracle_search_fp32.argtypes = _lib.oracle_search_exact.argtypes
        _lib.oracle_search_fp32.restype = ctypes.c_int
    return _lib


def search_exact(q, c, k, id_offset=0, threads=0):
    """Canonical exact search through the C restatement (``oracle/search_exact.c``);
    same contract as ``search_exact_numpy``, usable up to ~1e11 pairs."""
    lib = _load_c()
    q = np.ascontiguousarray(q, np.float32)
    c = np.ascontiguousarray(c, np.float32)
    nq, d = q.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    rc = lib.oracle_search_exact(q.ctypes.data, nq, c.ctypes.data, c.shape[0], d, k,
                                 id_offset, D.ctypes.data, I.ctypes.data, threads)
    assert rc == 0
    return D, I


# ------------------------------------------------------- faiss-shaped fp32 search (CPU baseline)
def search_fp32_blocked(q, c, k, block=16384, threads=None):
    """What the reference's CPU path does inside ``IndexFlatIP.search``
    (test_amazon_filterd.py:578; Appendix A.5): blocked float32 SGEMM over corpus chunks,
    per-query top-k per chunk, running merge.  float32 BLAS summation order, so near-ties
    may order differently from the canonical contract -- this is the timed CPU baseline, not
    the parity checker."""
    import torch
    if threads:
        torch.set_num_threads(threads)
    tq = torch.from_numpy(np.ascontiguousarray(q, np.float32))
    tc = torch.from_numpy(np.ascontiguousarray(c, np.float32))
    n = tc.shape[0]
    best_d = torch.full((tq.shape[0], k), float(NEG_SENTINEL))
    best_i = torch.full((tq.shape[0], k), -1, dtype=torch.int64)
    for lo in range(0, n, block):
        s = tq @ tc[lo:lo + block].T
        kk = min(k, s.shape[1])
        d_, i_ = torch.topk(s, kk, dim=1)
        cd = torch.cat([best_d, d_], dim=1)
        ci = torch.cat([best_i, i_ + lo], dim=1)
        d2, sel = torch.topk(cd, k, dim=1)
        best_d, best_i = d2, torch.gather(ci, 1, sel)
    return best_d.numpy(), best_i.numpy()


# ----------------------------------------------------------------------------- shard merge
def merge_topk(D_parts, I_parts, k):
    """k-way merge of per-shard results by (score desc, id asc); -1 ids sort last."""
    D = np.concatenate(D_parts, axis=1)
    I = np.concatenate(I_parts, axis=1)
    big = np.where(I < 0, np.iinfo(np.int64).max, I)
    order = np.lexsort((big, -D.astype(np.float64)), axis=1)[:, :k]
    return np.take_along_axis(D, order, axis=1), np.take_along_axis(I, order, axis=1)


# ------------------------------------------------------------------- flat index (faiss-shaped)
class FlatIndexRef:
    """``faiss.IndexFlatIP`` / ``IndexFlatL2`` stand-in with the canonical contract."""

    def __init__(self, d, metric="ip"):
        self.d, self.metric = d, metric
        self.xb = np.zeros((0, d), np.float32)

    @property
    def ntotal(self):
        return self.xb.shape[0]

    def add(self, x):
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape[1] == self.d
        self.xb = np.concatenate([self.xb, x], axis=0)

    def search(self, q, k):
        q = np.ascontiguousarray(q, np.float32)
        if self.metric == "l2":
            return topk_from_scores(canonical_l2(q, self.xb), k, largest=False)
        try:
            return search_exact(q, self.xb, k)
        except RuntimeError:
            return search_exact_numpy(q, self.xb, k)


def build_index(emb, metric):
    """Reference ``build_index`` (test_amazon_filterd.py:207-223)."""
    if metric == "cos":
        index = FlatIndexRef(emb.shape[1], "ip")
        index.add(normalize(emb))
    elif metric == "l2":
        index = FlatIndexRef(emb.shape[1], "l2")
        index.add(emb)
    elif metric == "ip":
        index = FlatIndexRef(emb.shape[1], "ip")
        index.add(emb)
    else:
        raise RuntimeError("Unregnozed metric", metric)
    return index


# -------------------------------------------------------------------- neighbour vote, p / r
def knn_item_vote(D_row, I_row, session_items, K):
    """``get_prediction_by_knn`` after the search (test_amazon_filterd.py:64-78).  Every item of
    neighbour session i gets weight D[i]: the reference multiplies an int64 ``ones_like`` array by
    the float32 scalar ``D[i]`` (-> a float64 array, :69) and adds those float64 values to a
    Python accumulator that starts at int 0 (:71-73), so weights are summed IN FLOAT64, in
    neighbour order; the K heaviest items are returned (python's stable sort, :74: ties keep
    first-seen order)."""
    aw = defaultdict(lambda: 0)
    for dist, sid in zip(D_row, I_row):
        if sid < 0:
            continue
        wgt = np.float64(np.float32(dist))
        for item in session_items[int(sid)]:
            aw[int(item)] += wgt
    sorted_aw = sorted(aw.items(), key=lambda x: x[1], reverse=True)
    return [p[0] for p in sorted_aw[:K]]


def knn_item_vote_weights(D_row, I_row, session_items, K):
    """Same vote, also returning the float64 weights of the K items (checker for the device kernel)."""
    aw = defaultdict(lambda: 0)
    for dist, sid in zip(D_row, I_row):
        if sid < 0:
            continue
        wgt = np.float64(np.float32(dist))
        for item in session_items[int(sid)]:
            aw[int(item)] += wgt
    sorted_aw = sorted(aw.items(), key=lambda x: x[1], reverse=True)[:K]
    return [p[0] for p in sorted_aw], [float(p[1]) for p in sorted_aw]


def get_p_r(gt, pred, K):
    """Reference ``get_p_r`` (test_amazon_filterd.py:80-85)."""
    pred = pred[:K]
    hit = float(len(gt & set(pred)))
    return hit / K, hit / len(gt)


def recall_at_k(I_got, I_ref, k):
    """Mean overlap of returned ids with the oracle's exact top-k (recall@k of BASELINE.json)."""
    hits = [len(set(a[:k].tolist()) & set(b[:k].tolist())) for a, b in zip(I_got, I_ref)]
    return float(np.mean(hits)) / k


# ------------------------------------------------------------------------- binary codes (Hamming)
def pack_sign_bits(emb):
    """The reference's packing of BinarizeHead outputs (fine_tune_ours.py:839-840):
    ``np.packbits(((emb + 1) / 2).astype(int), axis=1)``."""
    return np.packbits(((np.asarray(emb) + 1) / 2).astype(int), axis=1)


def hamming_search(q_codes, codes, k, id_offset=0):
    """``faiss.IndexBinaryFlat.search`` (Appendix A.5): Hamming distance ascending, int32 D; ties by
    ascending id (the build's rule); -1 / INT_MAX padding."""
    q_codes, codes = np.asarray(q_codes, np.uint8), np.asarray(codes, np.uint8)
    lut = np.array([bin(i).count("1") for i in range(256)], np.int32)
    nq, n = q_codes.shape[0], codes.shape[0]
    D = np.full((nq, k), 0x7fffffff, np.int32)
    I = np.full((nq, k), -1, np.int64)
    for a in range(nq):
        dist = lut[np.bitwise_xor(codes, q_codes[a][None, :])].sum(axis=1).astype(np.int64)
        order = np.lexsort((np.arange(n), dist))[:k]
        D[a, :order.size] = dist[order]
        I[a, :order.size] = order + id_offset
    return D, I

"""Retrieval harness pieces around the index: the neighbour-weighted item vote and the
prefix sub-session pipeline (BASELINE config C3).

Drop-ins (same names and argument meaning as the reference):

* ``get_prediction_by_knn(emb, index, dataset, sample_size, K)`` <- ``test_amazon_filterd.py:59-78``
  (``dataset`` here is a ``SessionItems`` -- the distinct items, ``product.x``, of every indexed
  graph -- instead of a list of PyG graphs; ``emb`` may hold many queries, the reference loops
  over them one by one with batch size 1, ``test_amazon_filterd.py:187-201``)
* ``get_p_r(gt, pred, K)``                                      <- ``test_amazon_filterd.py:80-85``

All arithmetic runs in the HIP kernels of ``libsss.so`` (``sss_ip_topk`` + ``sss_knn_item_vote``);
there is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


class SessionItems:
    """CSR session -> distinct item ids (the ``data['product'].x`` of each indexed graph), on device."""

    def __init__(self, ptr: torch.Tensor, items: torch.Tensor, id_offset: int = 0):
        self.ptr = _lib.require_cuda(ptr, "ptr", torch.int64)
        self.items = _lib.require_cuda(items, "items", torch.int32)
        self.id_offset = int(id_offset)

    @property
    def num_sessions(self) -> int:
        return int(self.ptr.shape[0] - 1)

    @classmethod
    def from_prepared(cls, pbs, id_offset: int = 0):
        """From natively built ``PreparedBatch``es (``SessionEncoder.prepare_actions``), in index
        order: the graph -> product-node pointer and the item ids are already on the device."""
        pbs = pbs if isinstance(pbs, (list, tuple)) else [pbs]
        ptrs, items, base = [], [], 0
        for pb in pbs:
            p = pb.p_ptr.to(torch.int64)
            ptrs.append(p[:-1] + base)
            items.append(pb.p_ids.to(torch.int32))
            base += int(pb.Np)
        ptr = torch.cat(ptrs + [torch.tensor([base], dtype=torch.int64, device=ptrs[0].device)])
        return cls(ptr.contiguous(), torch.cat(items).contiguous(), id_offset)

    @classmethod
    def from_batch(cls, batch, device, id_offset: int = 0):
        """From a ``SessionBatch`` (or several, concatenated in index order): product nodes are
        already grouped by graph, in node order = the order of ``product.x``."""
        batches = batch if isinstance(batch, (list, tuple)) else [batch]
        ptrs, items, base = [np.zeros(1, np.int64)], [], 0
        for b in batches:
            b = b.to_numpy()
            x, gb = np.asarray(b["product"].x), np.asarray(b["product"].batch)
            counts = np.bincount(gb, minlength=b.num_graphs).astype(np.int64)
            ptrs.append(base + np.cumsum(counts))
            items.append(x.astype(np.int32))
            base += int(counts.sum())
        return cls(torch.from_numpy(np.concatenate(ptrs)).to(device),
                   torch.from_numpy(np.concatenate(items) if items else np.zeros(0, np.int32)).to(device), id_offset)


def knn_item_vote(D: torch.Tensor, I: torch.Tensor, dataset: SessionItems, K: int, return_weights: bool = False):
    """Device vote over an existing search result (CUDA tensors in and out)."""
    L = _lib.lib()
    _lib.require_cuda(D, "D", torch.float32)
    _lib.require_cuda(I, "I", torch.int64)
    nq, S = D.shape
    out = torch.empty((nq, K), dtype=torch.int64, device=D.device)
    wts = torch.empty((nq, K), dtype=torch.float64, device=D.device) if return_weights else None
    status = torch.empty((nq,), dtype=torch.int32, device=D.device)
    rc = L.sss_knn_item_vote(D.data_ptr(), I.data_ptr(), nq, S, dataset.ptr.data_ptr(), dataset.items.data_ptr(),
                             dataset.id_offset, dataset.num_sessions, K, out.data_ptr(),
                             0 if wts is None else wts.data_ptr(), status.data_ptr(), _lib.stream_ptr(D.device))
    _lib.check(rc, "sss_knn_item_vote")
    return (out, wts, status) if return_weights else (out, status)


def get_prediction_by_knn(emb, index, dataset: SessionItems, sample_size: int, K: int):
    """Reference ``get_prediction_by_knn``: search ``sample_size`` neighbours, weight every item of
    a neighbour session by its similarity, return the K heaviest items per query
    (``[nq, K]`` int64 CUDA tensor, -1 padded; a python list of ids for a single 1-D query)."""
    one = emb.dim() == 1 if isinstance(emb, torch.Tensor) else np.ndim(emb) == 1
    q = emb.view(1, -1) if (one and isinstance(emb, torch.Tensor)) else (np.reshape(emb, (1, -1)) if one else emb)
    if isinstance(q, np.ndarray):
        q = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32))
    q = q.detach().to(index.device, torch.float32).contiguous()
    D, I = index.search_device(q, int(sample_size))
    out, status = knn_item_vote(D, I, dataset, int(K))
    if int(status.sum().item()) != 0:
        raise _lib.SssError("get_prediction_by_knn: a query has more than 16384 (neighbour, item) pairs")
    if one:
        return [int(v) for v in out[0].tolist() if v >= 0]
    return out


def get_p_r(gt, pred, K):
    """Reference ``get_p_r`` (test_amazon_filterd.py:80-85): precision and recall at K."""
    pred = list(pred)[:K]
    hit = float(len(set(gt) & set(pred)))
    return hit / K, hit / len(gt)

"""Row-sharded flat index over the GPUs of one node: one process per GPU, the normalised corpus
split by rows, per-shard exact top-k, ONE all-gather of the packed (ids | scores) block over
RCCL/xGMI, then a k-way merge on every rank.

The reference is single-process (no NCCL/MPI call site anywhere); this is the one parallel
strategy the path needs, and the only collective is that all-gather of ``12 * nq * k`` bytes per
rank -- latency-bound, so it is a single ``all_gather_into_tensor`` rather than a ring of
small messages.  The query batch is embedded cooperatively: every rank embeds nq / world of the
sessions and one all-gather of ``4 * nq * d`` bytes hands every rank the whole batch
(``gather_query_embeddings``) -- once the scan takes ~0.1 ms per shard, embedding the full batch on
every rank would be the Amdahl term of strong scaling.

The local search and the merge are injected (``engine``) so the sharding / packing / gather
logic can be exercised on CPU with the ``gloo`` backend in the tests; the default engine is the
HIP one and has no CPU fallback.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n: int, world: int, rank: int):
    """Contiguous row range of ``rank``: sizes differ by at most one row, earlier ranks larger."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def query_slice(nq: int, world: int, rank: int):
    """Rows of the query batch that ``rank`` embeds; equal sizes (the all-gather needs them), so
    (0, nq) -- every rank embeds everything -- when world does not divide nq."""
    if world <= 1 or nq % world:
        return 0, nq
    per = nq // world
    return rank * per, (rank + 1) * per


def gather_query_embeddings(emb_local: torch.Tensor, nq: int, out: torch.Tensor | None = None, group=None,
                            force_collective: bool = False):
    """All ranks' [nq / world, d] slices (``query_slice`` order) -> the full [nq, d] batch on every rank.
    ``force_collective`` issues the all-gather even where it moves nothing new (one rank, or every rank
    embedded the whole batch and world == 1): the 1-rank RCCL execution of tests / ``bench.py --force-collectives``."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if force_collective and dist.is_initialized() and emb_local.shape[0] * world == nq:
        pass
    elif world == 1 or emb_local.shape[0] == nq:
        return emb_local
    if out is None:
        out = torch.empty((nq, emb_local.shape[1]), dtype=emb_local.dtype, device=emb_local.device)
    dist.all_gather_into_tensor(out, emb_local.contiguous(), group=group)
    return out


class HipEngine:
    """Local fused search + merge through libsss (the product engine)."""

    def __init__(self, index):
        self.index = index
        # running count of queries the fused path could not prove exact (device side, no sync)
        self.unproven = torch.zeros(1, dtype=torch.int32, device=index.device)

    def local_search(self, q, k, D, I, status):
        self.index.search_fused(q, k, (D, I, status), self.unproven)

    def fix_unproven(self, q, k, D, I, status):
        return self.index.fix_unproven(q, k, D, I, status)

    def merge(self, pack_all, chunk, shards, nq, k, D_out, I_out):
        from . import _lib
        i_ptr = pack_all.data_ptr()
        d_ptr = i_ptr + nq * k * 8
        rc = _lib.lib().sss_topk_merge(d_ptr, 2 * chunk, i_ptr, chunk, shards, nq, k, D_out.data_ptr(),
                                       I_out.data_ptr(), _lib.stream_ptr(pack_all.device))
        _lib.check(rc, "sss_topk_merge")


class ShardedFlatIndex:
    """``index.search(q, k)`` over a corpus row-sharded across ``dist`` ranks.

    ``engine.local_search`` must write this rank's exact top-k with GLOBAL ids (id_offset =
    first row of the shard).  Every rank returns the full merged result.
    """

    def __init__(self, engine, device, group=None, force_collectives: bool = False):
        self.engine = engine
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # one rank normally returns its local result as is; with force_collectives (an initialised process
        # group required) it takes the all-gather + merge route of the multi-rank path all the same
        self.exchange = self.world > 1 or (force_collectives and dist.is_initialized())
        self._bufs = {}

    def _buffers(self, nq, k):
        key = (nq, k)
        if key not in self._bufs:
            nk = nq * k
            chunk = nk + (nk + 1) // 2                       # int64 words: ids, then float32 scores
            pack = torch.zeros(chunk, dtype=torch.int64, device=self.device)
            pack_all = torch.zeros(self.world * chunk, dtype=torch.int64, device=self.device)
            I = pack[:nk].view(nq, k)
            D = pack[nk:].view(torch.float32)[:nk].view(nq, k)
            status = torch.zeros(nq, dtype=torch.int32, device=self.device)
            Do = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            Io = torch.empty((nq, k), dtype=torch.int64, device=self.device)
            self._bufs[key] = (chunk, pack, pack_all, D, I, status, Do, Io)
        return self._bufs[key]

    def search_async(self, q, k):
        """Enqueue local search -> all-gather -> merge; no host sync.  Returns (D, I, status):
        status is this rank's per-query "proven exact" vector (0 = proven)."""
        nq = q.shape[0]
        chunk, pack, pack_all, D, I, status, Do, Io = self._buffers(nq, k)
        self.engine.local_search(q, k, D, I, status)
        if not self.exchange:
            return D, I, status
        dist.all_gather_into_tensor(pack_all, pack, group=self.group)
        self.engine.merge(pack_all, chunk, self.world, nq, k, Do, Io)
        return Do, Io, status

    def search(self, q, k):
        """Exact search: re-runs locally unproven queries exhaustively before the exchange."""
        nq = q.shape[0]
        chunk, pack, pack_all, D, I, status, Do, Io = self._buffers(nq, k)
        self.engine.local_search(q, k, D, I, status)
        self.engine.fix_unproven(q, k, D, I, status)
        if not self.exchange:
            return D, I
        dist.all_gather_into_tensor(pack_all, pack, group=self.group)
        self.engine.merge(pack_all, chunk, self.world, nq, k, Do, Io)
        return Do, Io

"""world_size-2 (and 4) gloo tests of the row-sharded search plumbing on CPU.

The HIP engine cannot run here, so the test injects an oracle-backed engine with the same
interface: what is exercised is the product's sharding plan, packed (ids | scores) buffer,
all-gather and shard ordering -- the merged result must equal the unsharded oracle result.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import search_ref as sr
from sessionsimilaritysearch_amd.distributed import (ShardedFlatIndex, gather_query_embeddings, query_slice,
                                                       shard_range)


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_query_slice_is_an_equal_partition_or_everything():
    for nq, w in ((1024, 8), (1024, 2), (12, 4), (1, 1)):
        spans = [query_slice(nq, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == nq and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert len({b - a for a, b in spans}) == 1
    assert all(query_slice(13, 4, r) == (0, 13) for r in range(4))      # not divisible: every rank embeds the batch


class OracleEngine:
    def __init__(self, shard, id_offset):
        self.shard, self.off = shard, id_offset

    def local_search(self, q, k, D, I, status):
        d, i = sr.search_exact(q.numpy(), self.shard, k, id_offset=self.off, threads=1)
        D.copy_(torch.from_numpy(d)); I.copy_(torch.from_numpy(i)); status.zero_()

    def fix_unproven(self, q, k, D, I, status):
        return 0

    def merge(self, pack_all, chunk, shards, nq, k, D_out, I_out):
        nk = nq * k
        Ds, Is = [], []
        for s in range(shards):
            blk = pack_all[s * chunk:(s + 1) * chunk]
            Is.append(blk[:nk].view(nq, k).numpy())
            Ds.append(blk[nk:].view(torch.float32)[:nk].view(nq, k).numpy())
        d, i = sr.merge_topk(Ds, Is, k)
        D_out.copy_(torch.from_numpy(np.ascontiguousarray(d))); I_out.copy_(torch.from_numpy(np.ascontiguousarray(i)))


def _worker(rank, world, port, n, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(99)
        q = sr.normalize(rng.standard_normal((13, 32)).astype(np.float32))
        c = sr.normalize(rng.standard_normal((n, 32)).astype(np.float32))
        c[1] = c[n - 2]                                    # a cross-shard exact tie
        lo, hi = shard_range(n, world, rank)
        idx = ShardedFlatIndex(OracleEngine(c[lo:hi], lo), torch.device("cpu"))
        tq = torch.from_numpy(q)
        D, I = idx.search(tq, k)
        D2, I2, st = idx.search_async(tq, k)
        Dr, Ir = sr.search_exact(q, c, k, threads=1)
        ok = np.array_equal(I.numpy(), Ir) and np.array_equal(D.numpy(), Dr)
        ok = ok and np.array_equal(I2.numpy(), Ir) and int(st.sum()) == 0
        # cooperative embedding: each rank holds its query_slice of a 12-row batch, everyone ends with all of it
        full = torch.arange(12 * 4, dtype=torch.float32).view(12, 4)
        a, b = query_slice(12, world, rank)
        ok = ok and torch.equal(gather_query_embeddings(full[a:b].clone(), 12), full)
        open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "MISMATCH")
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n,k", [(2, 1001, 10), (4, 403, 10), (2, 5, 10)])
def test_sharded_search_equals_unsharded(tmp_path, world, n, k):
    mp.spawn(_worker, args=(world, _free_port(), n, k, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def _worker_forced(rank, world, port, out_dir):
    """world size 1 with force_collectives: the one-rank form of the exchange route (what the GPU test and
    `bench.py --force-collectives` run over RCCL), here over gloo."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        q = sr.normalize(rng.standard_normal((9, 16)).astype(np.float32))
        c = sr.normalize(rng.standard_normal((300, 16)).astype(np.float32))
        idx = ShardedFlatIndex(OracleEngine(c, 70), torch.device("cpu"), force_collectives=True)
        plain = ShardedFlatIndex(OracleEngine(c, 70), torch.device("cpu"))
        tq = torch.from_numpy(q)
        D, I = idx.search(tq, 5)
        D2, I2, _ = idx.search_async(tq, 5)
        Dr, Ir = sr.search_exact(q, c, 5, id_offset=70, threads=1)
        ok = idx.exchange and not plain.exchange
        ok = ok and np.array_equal(I.numpy(), Ir) and np.array_equal(D.numpy(), Dr) and np.array_equal(I2.numpy(), Ir)
        full = torch.arange(12.0).view(6, 2)
        out = torch.zeros_like(full)
        got = gather_query_embeddings(full, 6, out, force_collective=True)
        ok = ok and got is out and torch.equal(out, full)
        ok = ok and gather_query_embeddings(full, 6, out) is full          # default: one rank returns its input as is
        open(os.path.join(out_dir, "forced.txt"), "w").write("ok" if ok else "MISMATCH")
    finally:
        dist.destroy_process_group()


def test_one_rank_forced_exchange_route(tmp_path):
    mp.spawn(_worker_forced, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert open(tmp_path / "forced.txt").read() == "ok"

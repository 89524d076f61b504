"""Randomised parity sweep of FlatIndex.search against the CPU oracle: shapes, candidate scans, k regimes,
value scales, duplicates and incremental adds drawn from a fixed seed (the long version with more and larger
cases is scripts/dev/fuzz_search.py).  Indices and scores must be bit-identical to the oracle in every case."""
import numpy as np
import pytest
import torch

from oracle import search_ref as sr

pytestmark = pytest.mark.gpu


def _case(rng):
    d = int(rng.choice([64, 128, 128, 256, 512, 96, 320, 1600]))        # 320 / 1600: the K-tiled long-row scan
    scans = ["auto", "f32", "split", "f16"] if d in (64, 128, 256) else ["auto", "f16"] if d == 512 else ["auto"]
    n = int(rng.choice([1, 37, 1000, 4097, 30000, 70001]))
    return dict(d=d, scan=str(rng.choice(scans)), n=n if d < 256 else min(n, 30000) if d < 1600 else min(n, 4097),
                nq=int(rng.choice([1, 31, 64, 257, 600])), k=int(rng.choice([1, 5, 10, 12, 13, 16, 17, 20, 21, 50, 100, 200])),
                flavour=str(rng.choice(["unit", "unit", "raw", "scaled", "dups", "adds"])),
                bf16=bool(d in (128, 256, 512, 320) and rng.random() < 0.2))


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_random_searches_match_oracle(cuda, seed):
    from sessionsimilaritysearch_amd.index import FlatIndex
    rng = np.random.default_rng(seed)
    for case in range(14):
        p = _case(rng)
        d, n, nq, k = p["d"], p["n"], p["nq"], p["k"]
        c = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal((nq, d)).astype(np.float32)
        if p["flavour"] in ("unit", "dups", "adds"):
            c, q = sr.normalize(c).astype(np.float32), sr.normalize(q).astype(np.float32)
        if p["flavour"] == "scaled":
            c *= np.float32(10.0 ** rng.uniform(-6, 6)); q *= np.float32(10.0 ** rng.uniform(-6, 6))
        if p["flavour"] == "dups" and n > 10:
            c[rng.integers(0, n, n // 3)] = c[rng.integers(0, n, n // 3)]
        if p["bf16"]:                                # bf16 index: the contract is defined on the rounded vectors
            c = torch.from_numpy(c).to(torch.bfloat16).float().numpy()
            q = torch.from_numpy(q).to(torch.bfloat16).float().numpy()
        idx = FlatIndex(d, "ip", cuda, dtype="bf16" if p["bf16"] else "f32", scan=None if p["bf16"] else p["scan"])
        if p["flavour"] == "adds" and n > 3:
            cuts = sorted(set(int(v) for v in rng.integers(1, n, 3)))
            for lo, hi in zip([0] + cuts, cuts + [n]):
                idx.add(c[lo:hi])
                if rng.random() < 0.5:
                    idx.search(q[:1], k)              # builds / extends the scan image between adds
        else:
            idx.add(c)
        D, I = idx.search(q, k)
        Dr, Ir = sr.search_exact(q, c, k)
        assert np.array_equal(I, Ir) and np.array_equal(D, Dr), (seed, case, p, idx.last_scan)

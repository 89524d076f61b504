"""Dev helper: randomised parity sweep of FlatIndex.search against the CPU oracle (shapes, scans, k, value
scales, duplicates, incremental adds).  Prints one line per case; exit code 1 on any mismatch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import search_ref as sr
from sessionsimilaritysearch_amd.index import FlatIndex

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
long_rows = len(sys.argv) > 3 and sys.argv[3] == "long"      # the K-tiled long-row search only: 2 - 5 levels, disjoint levels, ties
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
bad = 0
for case in range(cases):
    d = int(rng.choice([64, 128, 128, 256, 512, 96]))
    scans = ["auto", "f32", "split", "f16"] if d in (64, 128, 256) else ["auto", "f16"] if d == 512 else ["auto"]
    scan = str(rng.choice(scans))
    n = int(rng.choice([1, 37, 1000, 4097, 30000, 70001, 150000, 300000, 600000]))
    if d >= 256:
        n = min(n, 150000)
    nq = int(rng.choice([1, 31, 64, 257, 600, 1024]))
    k = int(rng.choice([1, 5, 10, 10, 12, 13, 16, 17, 20, 21, 50, 100, 200, 500]))
    bf16 = d in (128, 256, 512) and rng.random() < 0.2
    flavour = str(rng.choice(["unit", "unit", "raw", "scaled", "dups", "adds"]))
    if long_rows:
        d = int(rng.choice([320, 512, 1024, 1600, 2048]))
        n = int(rng.choice([9000, 70001, 170000, 300000, 450001, 800000]))
        n = min(n, 480_000_000 // d)                            # <= 1.9 GB of float32 rows
        nq = int(rng.choice([1, 31, 64, 200, 300]))
        k = int(rng.choice([1, 10, 50, 100, 100, 200, 500, 1000]))
        scan, bf16 = "auto", d <= 1024 and rng.random() < 0.2
        flavour = str(rng.choice(["unit", "unit", "dups", "hot", "sorted", "scaled"]))
    c = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    if flavour in ("unit", "dups", "adds"):
        c, q = sr.normalize(c).astype(np.float32), sr.normalize(q).astype(np.float32)
    if flavour == "scaled":
        c *= np.float32(10.0 ** rng.uniform(-6, 6)); q *= np.float32(10.0 ** rng.uniform(-6, 6))
    if flavour == "dups" and n > 10:
        c[rng.integers(0, n, n // 3)] = c[rng.integers(0, n, n // 3)]
    if flavour == "hot":                        # groups of exact ties at the top of some queries, scattered over the tiles
        c, q = sr.normalize(c).astype(np.float32), sr.normalize(q).astype(np.float32)
        for j in range(min(3, nq)):
            c[rng.choice(n, int(rng.choice([3, 150, 700])), replace=False)] = q[j]
    if flavour == "sorted":                     # the best rows of query 0 all at one end of the corpus
        c, q = sr.normalize(c).astype(np.float32), sr.normalize(q).astype(np.float32)
        c = np.ascontiguousarray(c[np.argsort((1 if rng.random() < 0.5 else -1) * (c @ q[0]))])
    if bf16:                                    # bf16 index: the contract is defined on the rounded vectors
        c = torch.from_numpy(c).to(torch.bfloat16).float().numpy()
        q = torch.from_numpy(q).to(torch.bfloat16).float().numpy()
        scan = "native"
    idx = FlatIndex(d, "ip", dev, dtype="bf16" if bf16 else "f32", scan=None if bf16 else scan)
    if flavour == "adds" and n > 3:
        cuts = sorted(set(int(v) for v in rng.integers(1, n, 3)))
        for lo, hi in zip([0] + cuts, cuts + [n]):
            idx.add(c[lo:hi])
            if rng.random() < 0.5:
                idx.search(q[:1], k)          # builds / extends the scan image between adds
    else:
        idx.add(c)
    t0 = time.time()
    D, I = idx.search(q, k)
    Dr, Ir = sr.search_exact(q, c, k)
    ok = np.array_equal(I, Ir) and np.array_equal(D, Dr)
    bad += 0 if ok else 1
    print(f"case {case:3d} d={d:4d} n={n:7d} nq={nq:5d} k={k:4d} scan={scan:6s}->{idx.last_scan or 'exhaustive':10s} {flavour:7s} "
          f"fallback={idx.last_fallback_queries:5d} {'ok' if ok else 'MISMATCH'} ({time.time() - t0:.1f}s)", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)

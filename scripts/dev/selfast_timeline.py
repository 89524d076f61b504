"""Dev helper: phase cycles of k_select_fast (one wave per query) from libsss_sftl.so (make_selectfast_tl.py)."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sessionsimilaritysearch_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsss_sftl.so")
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_
dev = torch.device("cuda", 0)
for spec in sys.argv[1:] or ["1024,1000000,128,10,f16"]:
    nq, n, d, k, scan = spec.split(","); nq, n, d, k = int(nq), int(n), int(d), int(k)
    g = torch.Generator(device=dev); g.manual_seed(1)
    c = torch.randn((n, d), device=dev, generator=g); normalize_(c)
    q = torch.randn((nq, d), device=dev, generator=g); normalize_(q)
    idx = FlatIndex(d, "ip", dev, scan=scan).adopt(c); idx.corpus_max_norm()
    out = idx.search_fused(q, k)
    for _ in range(3): idx.search_fused(q, k, out)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 8))()
    L = _lib.lib(); L.sss_debug_selfast.argtypes = [ctypes.c_void_p]; L.sss_debug_selfast(buf)
    a = np.array(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)[:min(nq, 1024)]
    ph = np.diff(a[:, :7], axis=1)
    names = ["count load", "column loads + sort", "K2 rounds", "re-score (rows + f64 chain)", "norms + rank + write", "threshold, clear, decide"]
    print(spec)
    for i, nm in enumerate(names):
        print(f"  {nm:30s} median {np.median(ph[:, i]):8.0f} cycles   p90 {np.percentile(ph[:, i], 90):8.0f}   max {ph[:, i].max():8.0f}")
    tot = a[:, 6] - a[:, 0]
    print(f"  wave total: median {np.median(tot):.0f}, max {tot.max()} cycles; first start -> last end {a[:, 6].max() - a[:, 0].min()} cycles; "
          f"start spread {a[:, 0].max() - a[:, 0].min()}; candidates median {np.median(a[:, 7]):.0f} max {a[:, 7].max()}")

"""Dev helper: phase cycles of k_select_all at the reference's shape (1M x 1600, K = 100) from libsss_satl.so (make_selectall_tl.py)."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sessionsimilaritysearch_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsss_satl.so")
import torch
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_
n, d, k, nq = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (1000000, 1600, 100, 1024)))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
c = torch.randn((n, d), device=dev, generator=g); normalize_(c)
q = torch.randn((nq, d), device=dev, generator=g); normalize_(q)
idx = FlatIndex(d, "ip", dev).adopt(c); idx.corpus_max_norm()
out = idx.search_fused(q, k)
for _ in range(3): idx.search_fused(q, k, out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 8))()
L = _lib.lib(); L.sss_debug_selall.argtypes = [ctypes.c_void_p]; L.sss_debug_selall(buf)
a = np.array(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)[:min(nq, 1024)]
ph = np.diff(a[:, :6], axis=1)
names = ["load keys + query row", "k-th largest (cut)", "compact survivors", "re-score", "sort + write"]
for i, nm in enumerate(names):
    print(f"{nm:24s} median {np.median(ph[:, i]):9.0f} cycles   max {ph[:, i].max():9.0f}")
print("kept rows M: median", np.median(a[:, 6]), "max", a[:, 6].max(), "| survivors: median", np.median(a[:, 7]), "max", a[:, 7].max())
print("workgroup total: median", np.median(a[:, 5] - a[:, 0]), "cycles; first start -> last end", a[:, 5].max() - a[:, 0].min())

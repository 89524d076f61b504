"""Dev helper: phase stamps of k_select_fast (scripts/dev/libsss_tl.so: see scripts/dev/README.md)."""
import sys, os, ctypes, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sessionsimilaritysearch_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsss_tl.so")
import torch
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_
nq, n, d, k = 1024, 1000000, 128, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
c = torch.randn((n, d), device=dev, generator=g); normalize_(c)
q = torch.randn((nq, d), device=dev, generator=g); normalize_(q)
idx = FlatIndex(d, "ip", dev).adopt(c)
out = idx.search_fused(q, k)
for _ in range(10):
    idx.search_fused(q, k, out)
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (1024 * 8))()
L.sss_debug_select_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.sss_debug_select_timeline(buf, 1024 * 8)
a = np.array(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
names = ["stage keys+query", "K2 argmax rounds", "f64 re-score", "norms/residual", "rank+write", "state/decide"]
dt = np.diff(a[:, :7], axis=1)
print(json.dumps({nm: float(np.median(dt[:, i])) for i, nm in enumerate(names)} | {"total_cycles_med": float(np.median(a[:, 6] - a[:, 0]))}))

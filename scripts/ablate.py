"""Dev helper: time only the scan (results are garbage in ablation builds)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_
dev = torch.device("cuda", 0)
for nq, n in ((1024, 1_000_000), (1024, 10_000_000)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    c = torch.randn((n, 128), device=dev, generator=g); normalize_(c)
    q = torch.randn((nq, 128), device=dev, generator=g); normalize_(q)
    idx = FlatIndex(128, "ip", dev).adopt(c); idx.corpus_max_norm()
    out = idx.search_fused(q, 10); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): idx.search_fused(q, 10, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(os.environ.get("SSS_LIB_PATH", "default"), n, round(ms, 3), "ms", round(2.0*nq*n*128/ms/1e9/157.3, 4), flush=True)
    del c, idx

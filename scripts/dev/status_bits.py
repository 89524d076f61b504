"""Dev helper: status bits the fused search leaves (1 full-list tail, 2 threshold, 4 near-tie window) per shape."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_
dev = torch.device("cuda", 0)
for spec in sys.argv[1:]:
    nq, n, d, k, scan = spec.split(",")
    nq, n, d, k = int(nq), int(n), int(d), int(k)
    tot = collections.Counter()
    for seed in range(6):
        g = torch.Generator(device=dev); g.manual_seed(seed + 1)
        c = torch.randn((n, d), device=dev, generator=g); normalize_(c)
        q = torch.randn((nq, d), device=dev, generator=g); normalize_(q)
        idx = FlatIndex(d, "ip", dev, scan=scan).adopt(c)
        D, I, st = idx.search_fused(q, k)
        tot.update(int(v) for v in st.cpu().tolist())
    print(spec, dict(tot), flush=True)

"""Dev helper for rocprofv3: a few launches of the fused search at one shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_
nq, n, d, k, iters = [int(a) for a in sys.argv[1:6]]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
c = torch.randn((n, d), device=dev, generator=g); normalize_(c)
q = torch.randn((nq, d), device=dev, generator=g); normalize_(q)
idx = FlatIndex(d, "ip", dev).adopt(c); idx.corpus_max_norm()
out = idx.search_fused(q, k)
for _ in range(iters):
    idx.search_fused(q, k, out)
torch.cuda.synchronize()

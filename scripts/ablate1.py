import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_
dev = torch.device("cuda", 0); nq, n = 1024, int(sys.argv[1])
g = torch.Generator(device=dev); g.manual_seed(1)
c = torch.randn((n, 128), device=dev, generator=g); normalize_(c)
q = torch.randn((nq, 128), device=dev, generator=g); normalize_(q)
idx = FlatIndex(128, "ip", dev).adopt(c); idx.corpus_max_norm()
out = idx.search_fused(q, 10)
for _ in range(2): idx.search_fused(q, 10, out)
torch.cuda.synchronize()

"""Dev helper: candidates per query a fused scan leaves for the select kernel (workspace zeroed first, non-zero keys
counted afterwards).   python scripts/dev/cand_count.py [n] [scan]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sessionsimilaritysearch_amd import _lib
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scan = sys.argv[2] if len(sys.argv) > 2 else "f16"
nq, d, k = 1024, 128, 10
g = torch.Generator(device=dev); g.manual_seed(1)
c = torch.randn((n, d), device=dev, generator=g); normalize_(c)
q = torch.randn((nq, d), device=dev, generator=g); normalize_(q)
idx = FlatIndex(d, "ip", dev, scan=scan).adopt(c)
idx.search_fused(q, k)                                  # builds images, sizes the workspace
L = _lib.lib()
nbytes = L.sss_ip_topk_f16_workspace_bytes(nq, n, d, k) if scan == "f16" else L.sss_ip_topk_workspace_bytes(nq, n, d, k, 0)
for rep in range(3):
    idx._ws = torch.zeros(idx._ws.numel(), dtype=torch.uint8, device=dev)
    D, I, st = idx.search_fused(q, k)
    torch.cuda.synchronize()
    cap = nbytes // 8 // nq
    keys = idx._ws[:nq * cap * 8].view(torch.int64).view(nq, cap)
    m = (keys != 0).sum(1).float()
    print(f"n={n} scan={scan} cap={cap}: candidates per query mean {m.mean().item():.0f} min {m.min().item():.0f} max {m.max().item():.0f}; unproven {int((st != 0).sum())}", flush=True)

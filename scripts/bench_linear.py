"""Dev helper: time sss_linear on the encoder's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sessionsimilaritysearch_amd import _lib
dev = torch.device("cuda", 0); L = _lib.lib(); st = _lib.stream_ptr(dev)
for (n, m, k) in [(5120, 672, 128), (5120, 160, 128), (5120, 384, 128), (5120, 108, 384), (8400, 128, 128), (1024, 128, 128), (160000, 672, 128)]:
    x = torch.randn((n, k), device=dev); w = torch.randn((m, k), device=dev); b = torch.randn(m, device=dev)
    y = torch.empty((n, m), device=dev)
    f = lambda: L.sss_linear(x.data_ptr(), k, w.data_ptr(), k, b.data_ptr(), y.data_ptr(), m, n, m, k, st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"n={n} m={m} k={k}: {us:.1f} us  {2.0*n*m*k/us/1e6:.1f} TFLOP/s", flush=True)

import sys, os, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sessionsimilaritysearch_amd import _lib
from sessionsimilaritysearch_amd.variants import _prob
dev = torch.device("cuda", 0); L = _lib.lib(); st = _lib.stream_ptr(dev)
for (n, m, k) in [(165000, 898, 128), (165000, 898, 384), (260000, 216, 384), (100000, 130, 128)]:
    x = torch.randn((n, k), device=dev); w = torch.randn((m, k), device=dev); y = torch.empty((n, m), device=dev)
    arr = (_lib.LinearProblem * 1)(_prob(x, w, None, y, n, m))
    f = lambda: L.sss_linear_grouped(arr, 1, k, st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"n={n} m={m} k={k}: {us:.1f} us  {2.0*n*m*k/us/1e6:.1f} TFLOP/s", flush=True)

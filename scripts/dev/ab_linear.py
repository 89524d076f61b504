"""Dev helper: times sss_linear_grouped and sss_linear (since round 3 the same kernel: one problem of the grouped GEMM) on
corpus-build shapes and on the reference model's shapes, with the vendor library's f32 GEMM (torch.mm) as a yardstick
(DESIGN.md section 5.2).  ab_linear.py [ref]"""
import sys, os, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sessionsimilaritysearch_amd import _lib
from sessionsimilaritysearch_amd.variants import _prob
dev = torch.device("cuda", 0); L = _lib.lib(); st = _lib.stream_ptr(dev)
shapes = [(165000, 898, 128), (165000, 898, 384), (260000, 216, 384), (100000, 130, 128)]
if len(sys.argv) > 1 and sys.argv[1] == "ref":      # 1024 sessions at d_in 768 / h 800 / L 3 / D 1600
    shapes = [(5152, 802, 768), (5152, 4800, 768), (5152, 800, 800), (5152, 2400, 800), (3562, 802, 800), (8714, 1500, 3168), (9287, 1600, 1600)]
def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (n, m, k) in shapes:
    x = torch.randn((n, k), device=dev); w = torch.randn((m, k), device=dev); y = torch.empty((n, m), device=dev); y2 = torch.empty((n, m), device=dev)
    arr = (_lib.LinearProblem * 1)(_prob(x, w, None, y, n, m))
    us_g = timeit(lambda: L.sss_linear_grouped(arr, 1, k, st))
    us_f = timeit(lambda: L.sss_linear(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), 0, y2.data_ptr(), y2.stride(0), n, m, k, st))
    same = bool(torch.equal(y, y2))
    us_t = timeit(lambda: torch.mm(x, w.t(), out=y2))          # yardstick only: the vendor library's f32 GEMM
    err = float((y - y2).abs().max() / y.abs().max())
    print(f"n={n} m={m} k={k}: grouped {us_g:.1f} us {2.0*n*m*k/us_g/1e6:.1f} TFLOP/s | per-op {us_f:.1f} us {2.0*n*m*k/us_f/1e6:.1f} TFLOP/s | identical {same} | "
          f"library {us_t:.1f} us {2.0*n*m*k/us_t/1e6:.1f} TFLOP/s (rel diff {err:.1e})", flush=True)

"""Dev helper: time the fused scoring kernel on random unit vectors (not the official bench)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_

# MI355X_MICROARCH.md dense peaks, TFLOP/s, of the pipe the scan that RAN issues on (a float32 index scanned through
# its f16 / split / long image runs on the 16-bit MFMA: the split scan spends three passes per pair-element)
PEAK = {"f32": 157.3, "native": 2500.0, "f16": 2500.0, "long": 2500.0, "split": 2500.0 / 3}

def run(nq, n, d, k, dtype="f32", iters=5):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    c = torch.randn((n, d), device=dev, generator=g); normalize_(c)
    q = torch.randn((nq, d), device=dev, generator=g); normalize_(q)
    if dtype == "bf16":
        from sessionsimilaritysearch_amd.index import to_bf16
        c, q = to_bf16(c), to_bf16(q)
    scan = None
    if dtype in ("f16", "split", "f32mfma"):    # float32 index, explicit candidate scan
        scan, dtype = ("f32" if dtype == "f32mfma" else dtype), "f32"
    idx = FlatIndex(d, "ip", dev, dtype=dtype, scan=scan).adopt(c)
    idx.corpus_max_norm()
    out = idx.search_fused(q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        idx.search_fused(q, k, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * nq * n * d / (ms * 1e-3) / 1e12
    bad = int(out[2].sum().item())
    print(json.dumps(dict(nq=nq, n=n, d=d, k=k, dtype=dtype, scan=idx.last_scan, ms=round(ms, 4), tflops=round(tf, 2), peak=round(PEAK[idx.last_scan], 1), frac=round(tf / PEAK[idx.last_scan], 4),
                          qps=round(nq / (ms * 1e-3)), unproven=bad)), flush=True)

if __name__ == "__main__":
    shapes = [(1024, 125_000, 128, 10), (1024, 250_000, 128, 10), (1024, 500_000, 128, 10),
              (1024, 1_000_000, 128, 10), (1024, 10_000_000, 128, 10),
              (256, 1_000_000, 128, 10), (4096, 1_000_000, 128, 10), (1024, 1_000_000, 64, 10),
              (1024, 1_000_000, 256, 10), (1024, 1_000_000, 128, 100), (1024, 125_000, 64, 10),
              (1024, 125_000, 128, 100), (1024, 4_000_000, 128, 500),
              (4096, 1_250_000, 256, 10, "bf16"), (4096, 10_000_000, 256, 10, "bf16"), (1024, 1_000_000, 256, 10, "bf16")]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) if v.isdigit() else v for v in a.split(",")) for a in sys.argv[1:]]
    for s in shapes:
        run(*s)

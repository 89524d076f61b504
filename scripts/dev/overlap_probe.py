"""Dev probe: does the query-batch encoder (6 small launches) overlap with the scoring kernel of the previous batch when the
two are enqueued on different streams?  Prints sequential vs two-stream time per step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sessionsimilaritysearch_amd import sessions as S
from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, init_weights
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_
dev = torch.device("cuda", 0)
scan = sys.argv[1] if len(sys.argv) > 1 else "f16"
cfg = EncoderConfig(d_in=128, h=128, n_layers=2, d_out=128, self_loop_rule="none")
enc = SessionEncoder(cfg, init_weights(cfg, 1236), dev)
pb = enc.prepare(S.build_batch(S.synthetic_actions(1024, 20269999, cfg.n_items, cfg.n_query)).to(dev))
g = torch.Generator(device=dev); g.manual_seed(1)
c = torch.randn((1_000_000, 128), device=dev, generator=g); normalize_(c)
idx = FlatIndex(128, "ip", dev, scan=scan).adopt(c); idx.corpus_max_norm()
q = enc(pb, l2_normalize=True).clone()
out = idx.search_fused(q, 10)
torch.cuda.synchronize()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_e = timeit(lambda: enc(pb, l2_normalize=True))
t_s = timeit(lambda: idx.search_fused(q, 10, out))
def seq():
    enc(pb, l2_normalize=True); idx.search_fused(q, 10, out)
t_seq = timeit(seq)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def two():
    with torch.cuda.stream(sa):
        enc(pb, l2_normalize=True)
    with torch.cuda.stream(sb):
        idx.search_fused(q, 10, out)
def timeit2(n=50):
    for _ in range(5): two()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(n): two()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print(f"scan={scan}: embed {t_e:.4f} ms, search {t_s:.4f} ms, sequential {t_seq:.4f} ms, two streams {timeit2():.4f} ms", flush=True)

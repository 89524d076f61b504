"""Dev helper: host enqueue time vs GPU time per step (embed nq/G sessions + search one shard), the per-rank\nfigures behind the strong-scaling projection of DESIGN.md section 7."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from sessionsimilaritysearch_amd import sessions as S
from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, init_weights
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_
from sessionsimilaritysearch_amd.distributed import HipEngine, ShardedFlatIndex
dev = torch.device("cuda", 0)
cfg = EncoderConfig(d_in=128, h=128, n_layers=2, d_out=128, self_loop_rule="none")
enc = SessionEncoder(cfg, init_weights(cfg, 1236), dev).eval()
for nsess, n in ((1024, 1000000), (512, 500000), (256, 250000), (128, 125000)):
    qb = enc.prepare_actions(S.synthetic_actions(nsess, 20269999, cfg.n_items, cfg.n_query))
    g = torch.Generator(device=dev); g.manual_seed(1)
    c = torch.randn((n, 128), device=dev, generator=g); normalize_(c)
    idx = FlatIndex(128, "ip", dev).adopt(c); idx.prepare(10)
    sh = ShardedFlatIndex(HipEngine(idx), dev)
    qall = torch.randn((1024, 128), device=dev, generator=g); normalize_(qall)      # the gathered batch: every rank searches ALL queries
    def step():
        emb = enc(qb, l2_normalize=True)             # this rank's nq / G sessions
        qall[:nsess] = emb                           # (stands in for the all-gather landing the slices in the full batch)
        return sh.search_async(qall, 10)
    for _ in range(10): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"sessions={nsess} rows={n}: host enqueue {1e6*(t1-t0)/300:.0f} us/step, total {1e6*(t2-t0)/300:.0f} us/step", flush=True)

"""Dev helper: time the fused encoder forward on a 1024-session prepared batch (kernel breakdown via rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sessionsimilaritysearch_amd import sessions as S
from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, init_weights
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = EncoderConfig(d_in=128, h=128, n_layers=2, d_out=128, self_loop_rule="none")
enc = SessionEncoder(cfg, init_weights(cfg, 1236), dev)
pb = enc.prepare(S.build_batch(S.synthetic_actions(n, 20269999, cfg.n_items, cfg.n_query)).to(dev))
for _ in range(5):
    enc(pb, l2_normalize=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    enc(pb, l2_normalize=True)
e1.record(); torch.cuda.synchronize()
print(f"sessions={n} Np={pb.Np} Nq={pb.Nq} n_exp={pb.n_clicks + pb.Nq}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per forward", flush=True)

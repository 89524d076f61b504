"""Dev helper: time the encoder forward on a prepared batch (kernel breakdown via rocprofv3).
   bench_encoder.py [sessions] [ref]   -- `ref` = the deployed model's shapes (d_in 768, h 800, 3 layers, D 1600:
   pretrain_filtered_amazon.py:267,281; config.py:15-16,21), which run on the per-op kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sessionsimilaritysearch_amd import sessions as S
from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, init_weights
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ref = len(sys.argv) > 2 and sys.argv[2] == "ref"
cfg = (EncoderConfig(d_in=768, h=800, n_layers=3, d_out=1600, n_items=100000, n_query=65) if ref else
       EncoderConfig(d_in=128, h=128, n_layers=2, d_out=128, self_loop_rule="none"))
enc = SessionEncoder(cfg, init_weights(cfg, 1236), dev)
pb = enc.prepare(S.build_batch(S.synthetic_actions(n, 20269999, cfg.n_items, cfg.n_query)).to(dev))
for _ in range(5):
    enc(pb, l2_normalize=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
iters = 10 if ref else 50
for _ in range(iters):
    enc(pb, l2_normalize=True)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
# node-linear FLOPs of one forward (the GEMM work; the aggregations are gathers)
h, L, D = cfg.h, cfg.n_layers, cfg.d_out
fl = 0.0
for l in range(L):
    dx = cfg.d_in if l == 0 else h
    fl += 2.0 * pb.Np * dx * (h + 2) + 2.0 * pb.Nq * dx * (h + 2)      # GAT source / target transforms (+ attention columns)
    fl += 2.0 * pb.Np * dx * h + 2.0 * pb.Np * h * 3 * h * 2            # GatedGraphConv weight + GRU input / hidden transforms
W = cfg.node_width
fl += 2.0 * (pb.Np + pb.Nq) * W * (D - cfg.max_seq_len) + 2.0 * 2 * (pb.n_clicks + pb.Nq) * D * D
print(f"sessions={n} Np={pb.Np} Nq={pb.Nq} n_exp={pb.n_clicks + pb.Nq} cfg=({cfg.d_in},{h},{L},{D}) fused={enc.fused_ok()}: "
      f"{us:.1f} us per forward = {n / us * 1e6:.0f} sessions/s, ~{fl / us / 1e6:.1f} TFLOP/s of node-linear work", flush=True)

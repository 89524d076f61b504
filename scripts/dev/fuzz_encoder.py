"""Dev helper: randomised parity sweep of the fused encoder (table mode) against the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import gnn_ref
from sessionsimilaritysearch_amd import sessions as S
from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, init_weights

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
bad = 0
for case in range(cases):
    d_in, h = [(64, 64), (128, 128), (32, 64), (64, 128), (128, 128)][int(rng.integers(0, 5))]
    layers = int(rng.choice([1, 2, 3]))
    d_out = int(rng.choice([64, 96, 128]))
    n = int(rng.choice([1, 7, 64, 200, 513]))
    loops = bool(rng.random() < 0.5)
    cfg = EncoderConfig(d_in=d_in, h=h, n_layers=layers, d_out=d_out, n_items=int(rng.choice([50, 3000])), n_query=int(rng.choice([9, 257])))
    cfg.self_loop_rule = "pyg_bipartite_global" if loops else "none"
    w = init_weights(cfg, int(rng.integers(0, 1 << 30)))
    b = S.build_batch(S.synthetic_actions(n, int(rng.integers(0, 1 << 30)), cfg.n_items, cfg.n_query))
    enc = SessionEncoder(cfg, w, dev)
    got = enc(b.to(dev)).cpu()
    ref = gnn_ref.encoder_forward(b.to_torch("cpu"), w, cfg.n_layers, self_loops=loops)
    err = float((got - ref).abs().max()); scale = max(1.0, float(ref.abs().max()))
    ok = err < 1e-5 * scale
    bad += 0 if ok else 1
    print(f"case {case:3d} d_in={d_in:4d} h={h:4d} L={layers} D={d_out:4d} n={n:4d} loops={int(loops)} fused={int(enc.fused_ok())} err={err:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)

"""Dev helper: algorithmic HBM bytes of one k_layer_update launch at a corpus-build batch, to set against the FETCH_SIZE /
WRITE_SIZE counters of the same run:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/lt_fetch -- python3 scripts/dev/layer_traffic.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/lt_write -- python3 scripts/dev/layer_traffic.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sessionsimilaritysearch_amd import sessions as S
from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, init_weights
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cfg = EncoderConfig(d_in=128, h=128, n_layers=2, d_out=128, self_loop_rule="none")
enc = SessionEncoder(cfg, init_weights(cfg, 1236), dev)
pb = enc.prepare_actions(S.synthetic_actions(n, 8, cfg.n_items, cfg.n_query))
for _ in range(5):
    enc(pb, l2_normalize=True)
torch.cuda.synchronize()
h = cfg.h
E, Epp = int(pb.csr_qp[1].numel()), int(pb.csr_pp[1].numel())
# layer 1 (rows of the per-batch transforms Yp [Np, 7h+32], Yq [Nq, h+32]); layer 0 in table mode reads table rows instead
per_p = (3 * h + 2) * 4 + h * 4 + h * 4            # own gh + alphas, own x, output row
per_q = 2 * 4 + h * 4
alg = pb.Np * per_p + pb.Nq * per_q + E * (h + 1) * 4 * 2 + Epp * (3 * h) * 4 + (pb.Np + pb.Nq + 2 * E + Epp) * 4 * 2
print(json.dumps(dict(sessions=n, Np=pb.Np, Nq=pb.Nq, E=E, Epp=Epp, algorithmic_bytes_layer1=alg,
                      node_rows_bytes=(pb.Np * (7 * h + 32) + pb.Nq * (h + 32)) * 4)), flush=True)

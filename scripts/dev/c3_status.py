"""Dev helper: status bits (1 full-list tail, 2 threshold, 4 near-tie window) the fused k = 500 search of config C3 leaves,
the rows kept per unproven query by the threshold rung, and the stage times of the synchronous search."""
import sys, os, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from sessionsimilaritysearch_amd import sessions as S
from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, init_weights
from sessionsimilaritysearch_amd.index import FlatIndex
n_sessions = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
scan = sys.argv[2] if len(sys.argv) > 2 else "auto"
dev = torch.device("cuda", 0)
d, nq, k = 128, 1024, 500
cfg = EncoderConfig(d_in=d, h=d, n_layers=2, d_out=d, self_loop_rule="none")
enc = SessionEncoder(cfg, init_weights(cfg, 1234 + 3), dev).eval()
xb, items = bench.build_c3_corpus(enc, cfg, n_sessions, dev)
q_acts = S.synthetic_actions(nq, 20269999, cfg.n_items, cfg.n_query).prefix(1, 2)
emb = enc(enc.prepare_actions(q_acts), l2_normalize=True)
idx = FlatIndex(d, "ip", dev, scan=scan).adopt(xb)
idx.corpus_max_norm()
D, I = idx.search_device(emb, k)                 # lets scan="auto" settle
for _ in range(2):
    D, I, st = idx.search_fused(emb, k)
torch.cuda.synchronize()
stc = st.cpu().numpy()
print("scan", idx.last_scan, "status histogram", dict(collections.Counter(int(v) for v in stc)))
bad = np.nonzero(stc)[0]
# duplicates: how many rows tie with the 500th score of each unproven query?
x = xb
for qi in bad[:12]:
    s = (x @ emb[qi]).float()
    kth = torch.topk(s, k).values[-1]
    print(f"query {qi}: status {stc[qi]} rows tied with the k-th score: {(s == kth).sum().item()}, rows >= k-th - 1e-6: {(s >= kth - 1e-6).sum().item()}")
for name, fn in (("search_fused", lambda: idx.search_fused(emb, k)), ("search_device", lambda: idx.search_device(emb, k))):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print(name, round((time.time() - t0) / 5 * 1e3, 3), "ms", "rescan", idx.last_rescan_queries, "fallback", idx.last_fallback_queries)

"""Dev helper: throughput of the path's non-dominant kernels against their own rooflines (HBM / VALU), one JSON
line each.  Algorithmic bytes are stated next to every figure."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sessionsimilaritysearch_amd import _lib, sessions as S
from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, init_weights
from sessionsimilaritysearch_amd.index import BinaryFlatIndex, FlatIndex, normalize_, to_bf16
from sessionsimilaritysearch_amd.retrieval import SessionItems, knn_item_vote

dev = torch.device("cuda", 0)
HBM = 6.3e12          # achievable HBM rate (MI355X_MICROARCH.md), B/s


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def out(**kw):
    print(json.dumps(kw), flush=True)


# ---- row kernels: normalise, f32 -> bf16
x = torch.randn((4_000_000, 128), device=dev)
t = timed(lambda: normalize_(x))
out(kernel="k_normalize_rows", rows=x.shape[0], d=128, ms=round(t * 1e3, 3), bytes=2 * x.numel() * 4,
    GBps=round(2 * x.numel() * 4 / t / 1e9, 1), frac_of_hbm=round(2 * x.numel() * 4 / t / HBM, 3))
t = timed(lambda: to_bf16(x))
out(kernel="k_f32_to_bf16", elements=x.numel(), ms=round(t * 1e3, 3), bytes=x.numel() * 6, GBps=round(x.numel() * 6 / t / 1e9, 1),
    frac_of_hbm=round(x.numel() * 6 / t / HBM, 3))
del x

# ---- graph builder + encoder at the corpus-build batch
cfg = EncoderConfig(d_in=128, h=128, n_layers=2, d_out=128, self_loop_rule="none")
enc = SessionEncoder(cfg, init_weights(cfg, 1236), dev)
acts = S.synthetic_actions(262144, 7, cfg.n_items, cfg.n_query)
T = int(acts.sess_ptr[-1])
class DevActs: pass
da = DevActs()
da.sess_ptr = torch.from_numpy(acts.sess_ptr).to(dev); da.is_search = torch.from_numpy(acts.is_search.view(np.uint8)).to(dev)
da.item_id = torch.from_numpy(acts.item_id).to(dev); da.query_tok = torch.from_numpy(acts.query_tok).to(dev)
t = timed(lambda: enc.prepare_actions(da), n=5, warm=2)
pb = enc.prepare_actions(da)
wbytes = (pb.Nq * 20 + pb.Np * 24 + (pb.Np + pb.Nq + 2) * 8 + pb.csr_qp[1].numel() * 8 + pb.csr_pp[1].numel() * 8 + pb.src_row.numel() * 8)
out(kernel="graph build (k_session_counts + scans + k_session_fill, incl. allocation + 1 read-back)", sessions=262144, actions=T,
    ms=round(t * 1e3, 3), sessions_per_s=round(262144 / t), approx_bytes=int(2 * 25 * T + wbytes),
    GBps=round((2 * 25 * T + wbytes) / t / 1e9, 1))
acts32 = S.synthetic_actions(32768, 8, cfg.n_items, cfg.n_query)
pb32 = enc.prepare_actions(acts32)
t = timed(lambda: enc(pb32, l2_normalize=True), n=10)
h, W, D = 128, 384, 128
flop = 2.0 * (pb32.Np * (7 * h + 2) * 128 * 2 + pb32.Nq * (h + 2) * 128 * 2 + (pb32.Np + pb32.Nq) * 108 * W + (pb32.Np + pb32.Nq) * 256 * 128)
out(kernel="fused encoder forward (6 launches, layer 0 in table mode)", sessions=32768, ms=round(t * 1e3, 3), sessions_per_s=round(32768 / t),
    gflop=round(flop / 1e9, 1), TFLOPs=round(flop / t / 1e12, 1), frac_of_f32_mfma=round(flop / t / 157.3e12, 3))

# ---- item vote: 1024 queries x 500 neighbours
batch = S.build_batch(S.synthetic_actions(200000, 9, cfg.n_items, cfg.n_query))
ds = SessionItems.from_batch(batch, dev)
g = torch.Generator(device=dev); g.manual_seed(3)
I = torch.randint(0, 200000, (1024, 500), device=dev, generator=g)
Dv = torch.sort(torch.rand((1024, 500), device=dev, generator=g), dim=1, descending=True).values.contiguous()
t = timed(lambda: knn_item_vote(Dv, I, ds, 10))
pairs = float((ds.ptr[1:] - ds.ptr[:-1])[I.flatten()].sum().item())
out(kernel="k_item_vote", queries=1024, sample_size=500, pairs=int(pairs), ms=round(t * 1e3, 3), queries_per_s=round(1024 / t))

# ---- Hamming scan: 10M x 256-bit codes, 1024 queries, top-100
codes = torch.randint(0, 256, (10_000_000, 32), dtype=torch.uint8, device=dev, generator=g)
qc = torch.randint(0, 256, (1024, 32), dtype=torch.uint8, device=dev, generator=g)
hidx = BinaryFlatIndex(256, dev); hidx._codes = codes
t = timed(lambda: hidx.search(qc, 100), n=3, warm=1)
ops = 1024 * 1e7 * 8 * 2          # xor + popcount-accumulate per 32-bit word
out(kernel="k_hamming_scan + select (IndexBinaryFlat.search, k=100)", n=10_000_000, nq=1024, nbits=256, ms=round(t * 1e3, 2),
    queries_per_s=round(1024 / t), valu_Tops=round(ops / t / 1e12, 2), corpus_GBps=round(10_000_000 * 32 / t / 1e9, 1),
    fallback_queries=hidx.last_fallback_queries)

#!/usr/bin/env python3
"""Headline benchmark: session queries/sec + recall@10, 1M-session corpus, d=128.

One "step" = one pass of the hot path over one batch of 1024 synthetic query sessions that are
already resident in HBM as a prepared batched graph (CSR adjacencies + pooling indices: the
reference builds its graphs on the host before model.forward too, test_amazon_filterd.py:546-551):
GNN embed (HeteroGGNN x2 -> positional-attention pooling -> L2-normalise, 6 launches) -> fused MFMA
scoring + top-10 against this rank's corpus shard (2 launches) -> (N > 1) RCCL all-gather of the
packed per-shard results -> merge.

The default run times TWO legs over the same corpus, weights and query batch, back to back, and
prints both in the one JSON line:
  * the top-level line is the REFERENCE-PRECISION leg: the candidate scan runs on the f32 MFMA over
    the float32 rows (`dtype: "f32"`, faiss IndexFlatIP scores in float32; test_amazon_filterd.py:578),
    roofline against the 157.3 TFLOP/s f32 matrix peak;
  * `fast_path` is the production default (`scan="auto"`: at k <= 16 one f16 MFMA pass over a scaled
    float16 image of the same corpus), with its own ms_per_step / value / roofline / exactness keys.
Both legs return the SAME canonical results (float64 re-score of the candidates from the float32 rows
+ per-query proof); `--scan X` / `--dtype bf16` / `--workload c3` time that single configuration only.

The timed call is the asynchronous exact search (`ShardedFlatIndex.search_async`): every query's
result carries an on-device proof of exactness and unproven queries are COUNTED on the device
(`unproven_queries` in the line, summed over all timed steps and ranks).  With that count at 0
the step did exactly the work of the synchronous `ShardedFlatIndex.search` minus one host sync;
if it is not 0 the bench re-times with the synchronous API and reports that instead.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Scaling is STRONG: the corpus (1M sessions by default) is fixed and row-sharded over the N
ranks; `value` = queries / second of the whole job (every rank ends up with the merged result).
The corpus itself is built before the timed region by embedding synthetic sessions with the
same encoder (index build; not timed, as in the reference where the index is built once).

The default run (N = 1) also appends `reference_shapes`: the deployed model's OWN shapes, which are not the metric's
configuration -- the long-row search at D = 1600, K = 100 over 1M random rows (checked against the oracle on 16 queries
inside the run) and the encoder at d_in 768 / h 800 / 3 layers / D 1600 (`--no-reference-shapes` skips it).

Extra JSON objects (see DESIGN.md "measurement"):
  roofline     -- dominant kernel k_scan<row bytes, tile rows, scan type>: algorithmic FLOPs per launch
                  / its mean duration, hipEvent-timed on its own stream inside the timed region;
                  `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes of this
                  command (profiles/r03_traffic.json; FETCH_SIZE doubled per MI355X_MICROARCH.md
                  "HBM"), printed only while the entry's `scan_hip_sha16` still equals the hash of
                  the csrc/scan.hip being run -- null otherwise (a stale figure is worse than none).
  cpu_baseline -- the oracle's restatement of the reference CPU path (torch CPU encoder +
                  faiss-shaped blocked SGEMM/top-k search) on this host's cores, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sessionsimilaritysearch_amd import _lib  # noqa: E402
from sessionsimilaritysearch_amd import sessions as S  # noqa: E402
from sessionsimilaritysearch_amd.distributed import (HipEngine, ShardedFlatIndex, gather_query_embeddings,  # noqa: E402
                                                       query_slice, shard_range)
from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, init_weights  # noqa: E402
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_, to_bf16  # noqa: E402

CONFIG_INDEX = 2                 # seeds: SURVEY.md 8(d) (20260000 + config index, 1234 + config index)
BLOCK = 32768                    # sessions generated / embedded per block (seeded per block)
FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense f32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def build_c3_corpus(enc, cfg, n_sessions, device):
    """Config C3: every session contributes 4 prefix sub-sessions (25/50/75/100 % of its actions,
    the deterministic stand-in for the reference's random cut, train_subsession_embedding.py:41).
    Returns the normalised [4 * n_sessions, d] matrix (rows ordered block / prefix / session) and
    the graph -> distinct-items CSR the neighbour vote reads (product.x of every indexed graph)."""
    from sessionsimilaritysearch_amd.retrieval import SessionItems
    out = torch.empty((4 * n_sessions, cfg.d_out), dtype=torch.float32, device=device)
    pbs, row = [], 0
    for b0 in range(0, n_sessions, BLOCK):
        nb = min(BLOCK, n_sessions - b0)
        acts = S.synthetic_actions(nb, 20260000 + 3 * 100000 + b0 // BLOCK, cfg.n_items, cfg.n_query)
        for f in (1, 2, 3, 4):
            pb = enc.prepare_actions(acts.prefix(f, 4))
            out[row:row + nb] = enc(pb, l2_normalize=True)
            keep = type("Items", (), {})()              # only what SessionItems needs (drop the big CSR buffers)
            keep.p_ptr, keep.p_ids, keep.Np = pb.p_ptr.clone(), pb.p_ids, pb.Np
            pbs.append(keep)
            row += nb
    return out, SessionItems.from_prepared(pbs)


def build_corpus_shard(enc, cfg, n_total, lo, hi, device, source):
    """Normalised session vectors of rows [lo, hi) of the corpus, on device."""
    out = torch.empty((hi - lo, cfg.d_out), dtype=torch.float32, device=device)
    if source == "random":          # config C4-style scoring corpus: N(0,1) rows, generated on device
        g = torch.Generator(device=device)
        for b0 in range(lo - lo % BLOCK, hi, BLOCK):
            g.manual_seed(20260000 + CONFIG_INDEX * 100000 + b0 // BLOCK)
            blk = torch.randn((BLOCK, cfg.d_out), device=device, generator=g)
            a, b = max(lo, b0), min(hi, b0 + BLOCK, n_total)
            out[a - lo:b - lo] = blk[a - b0:b - b0]
        normalize_(out)
        return out
    for b0 in range(lo - lo % BLOCK, hi, BLOCK):
        nb = min(BLOCK, n_total - b0)
        acts = S.synthetic_actions(nb, 20260000 + CONFIG_INDEX * 100000 + b0 // BLOCK, cfg.n_items, cfg.n_query)
        a, b = max(lo, b0), min(hi, b0 + nb)
        if a > b0 or b < b0 + nb:
            acts = acts.slice(a - b0, b - b0)
        out[a - lo:b - lo] = enc(enc.prepare_actions(acts), l2_normalize=True)     # native graph build + fused encoder
    return out


def reference_shapes_leg(device, nq, sr, say):
    """The deployed model's OWN shapes (pretrain_filtered_amazon.py:267,281; config.py:15-16,21;
    test_amazon_filterd.py:459,488,578): session vectors of D = 1600, K = 100 neighbours, and the encoder at
    d_in 768 / h 800 / 3 layers.  Not the metric's configuration (BASELINE.json quotes d = 128, k = 10) -- an extra
    object in the line so that a driver-run record carries these figures too.  Scoring: 1M random unit rows x 1600,
    the K-tiled long-row scan (csrc/scan_long.hip); a few queries are checked against the oracle's canonical search."""
    from sessionsimilaritysearch_amd.index import FlatIndex, normalize_
    n, d, k = 1_000_000, 1600, 100
    g = torch.Generator(device=device); g.manual_seed(20261600)
    c = torch.randn((n, d), device=device, generator=g); normalize_(c)
    q = torch.randn((nq, d), device=device, generator=g); normalize_(q)
    idx = FlatIndex(d, "ip", device).adopt(c)
    idx.corpus_max_norm()
    out = idx.search_fused(q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 5
    e0.record()
    for _ in range(iters):
        idx.search_fused(q, k, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    unproven = int(out[2].sum().item())
    D, I = idx.search(q[:16].cpu().numpy(), k)          # the synchronous exact API on the checked queries (numpy in, numpy out)
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    t0 = time.perf_counter()
    Dr, Ir = sr.search_exact(q[:16].cpu().numpy(), c.cpu().numpy(), k, threads=cores)
    say(f"reference shapes: scoring {ms:.3f} ms/step, oracle check of 16 queries {time.perf_counter() - t0:.1f}s")
    tf = 2.0 * nq * n * d / (ms * 1e-3) / 1e12
    search = {"corpus_rows": n, "d": d, "k": k, "query_batch": nq, "scan": idx.last_scan, "ms_per_step": round(ms, 4),
              "value": round(nq / (ms * 1e-3), 1), "unit": "queries/s", "unproven_queries": unproven,
              "ids_bit_exact": bool(np.array_equal(I, Ir)), "max_score_err": float(np.abs(D - Dr).max()), "queries_checked": 16,
              "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                           "note": "whole search (sampled threshold levels + full scan + bounds + final re-score) / algorithmic FLOPs; "
                                   "the last level alone moves 25 GB through LDS-DMA in ~3.6 ms (7 TB/s, the path that bounds it: DESIGN.md 5.4)"}}
    del idx, c, out
    torch.cuda.empty_cache()
    cfg = EncoderConfig(d_in=768, h=800, n_layers=3, d_out=1600, n_items=100000, n_query=65)
    enc = SessionEncoder(cfg, init_weights(cfg, 20260800), device).eval()
    pb = enc.prepare_actions(S.synthetic_actions(nq, 20269999, cfg.n_items, cfg.n_query))
    for _ in range(2):
        enc(pb, l2_normalize=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        enc(pb, l2_normalize=True)
    e1.record(); torch.cuda.synchronize()
    ems = e0.elapsed_time(e1) / iters
    encoder = {"d_in": 768, "h": 800, "layers": 3, "d_out": 1600, "sessions": nq, "ms_per_forward": round(ems, 4),
               "value": round(nq / (ems * 1e-3), 1), "unit": "sessions/s", "path": "fused" if enc.fused_ok() else "per-op kernels (GEMM-bound: ~86 % of the time in k_linear_grouped at ~0.73 of the f32 MFMA peak)"}
    return {"note": "the deployed model's own shapes, same run; not the metric's configuration", "search": search, "encoder": encoder,
            "ms_per_step": round(ms + ems, 4), "value": round(nq / ((ms + ems) * 1e-3), 1), "unit": "queries/s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--corpus-rows", type=int, default=1_000_000)
    ap.add_argument("--corpus-source", choices=["sessions", "random"], default="sessions")
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--recall-queries", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="index element type (bf16 + --d 256 --nq 4096 --corpus-source random = config C5)")
    ap.add_argument("--scan", choices=["both", "auto", "f16", "split", "f32"], default="both",
                    help="candidate scan of a float32 index: 'f16' = scaled float16 image, one f16 MFMA pass; "
                         "'split' = bf16 hi/lo image, three bf16 MFMA passes; 'f32' = the f32 MFMA on the float32 "
                         "rows; 'auto' = f16 for k <= 16, split up to 128, f32 beyond; 'both' (default) = the f32 leg as the "
                         "top-level line and the 'auto' leg as `fast_path`.  Results are identical.")
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--workload", choices=["search", "c3"], default="search",
                    help="c3: 4 prefix sub-sessions per session indexed, top-500 neighbours -> item vote -> top-10 items (1 GPU)")
    ap.add_argument("--sample-size", type=int, default=500)
    ap.add_argument("--no-reference-shapes", action="store_true",
                    help="skip the extra leg at the deployed model's own shapes (D = 1600, K = 100; encoder 768/800/3/1600)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    # SSS_BENCH_BACKEND=gloo is a REHEARSAL mode for a one-GPU box (ranks share the card, the
    # all-gather goes through the host); the real multi-GPU run uses RCCL ("nccl" on ROCm).
    backend = os.environ.get("SSS_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    L = _lib.lib()
    d, k, nq, n_total = args.d, args.k, args.nq, args.corpus_rows
    cfg = EncoderConfig(d_in=d, h=d, n_layers=2, d_out=d, self_loop_rule="none")
    weights = init_weights(cfg, 1234 + CONFIG_INDEX)
    enc = SessionEncoder(cfg, weights, device).eval()

    # ---- index build (not timed): this rank's rows of the corpus
    c3 = args.workload == "c3"
    if c3 and (world != 1 or args.dtype != "f32"):
        raise SystemExit("--workload c3 is the single-GPU f32 configuration")
    n_sessions = n_total
    if c3:
        n_total = 4 * n_sessions                # rows of the index
    lo, hi = shard_range(n_total, world, rank)
    t0 = time.time()
    if c3:
        xb, session_items = build_c3_corpus(enc, cfg, n_sessions, device)
    else:
        xb = build_corpus_shard(enc, cfg, n_total, lo, hi, device, args.corpus_source)
    torch.cuda.synchronize()
    log(rank, f"corpus shard rows [{lo},{hi}) built in {time.time() - t0:.1f}s")
    if args.dtype == "bf16":
        xb = to_bf16(xb)
    # ---- query batch, resident in HBM
    q_acts = S.synthetic_actions(nq, 20269999, cfg.n_items, cfg.n_query)
    if c3:
        q_acts = q_acts.prefix(1, 2)             # the query is a prefix sub-session (test_amazon_filterd.py:546)
    q_host = S.build_batch(q_acts)               # host copy: only the CPU baseline / oracle read it
    # every rank embeds nq / world of the query sessions; one all-gather hands everyone the batch
    q_lo, q_hi = query_slice(nq, world, rank)
    qbatch = enc.prepare_actions(q_acts.slice(q_lo, q_hi) if (q_lo, q_hi) != (0, nq) else q_acts)   # batched CSR session graph built on device, resident in HBM
    emb_all = torch.empty((nq, d), dtype=torch.float32, device=device)
    k_items = k
    if c3:
        from sessionsimilaritysearch_amd.retrieval import knn_item_vote
        k = args.sample_size                     # neighbours searched; k_items items voted

    def embed():
        emb = gather_query_embeddings(enc(qbatch, l2_normalize=True), nq, emb_all)
        return to_bf16(emb) if args.dtype == "bf16" else emb

    def timed(fn, n=5):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    # ---- exactness reference: the oracle's canonical top-k of the checked queries (computed once, all legs share it)
    from oracle import search_ref as sr, gnn_ref       # checker + CPU baseline only
    nrq = min(args.recall_queries, nq) if not c3 else min(8, nq)
    oracle_ref = {}

    def oracle_topk(emb):
        if "Ir" not in oracle_ref:
            q_np = emb[:nrq].float().cpu().numpy()
            Dl, Il = sr.search_exact(q_np, xb.float().cpu().numpy(), k, id_offset=lo)
            if world > 1:
                pack = torch.cat([torch.from_numpy(Il).to(device).double(), torch.from_numpy(Dl).to(device).double()], 1)
                allp = [torch.empty_like(pack) for _ in range(world)]
                dist.all_gather(allp, pack)
                Ds = [p[:, k:].float().cpu().numpy() for p in allp]
                Is = [p[:, :k].long().cpu().numpy() for p in allp]
                Dl, Il = sr.merge_topk(Ds, Is, k)
            oracle_ref["Dr"], oracle_ref["Ir"] = Dl, Il
        return oracle_ref["Dr"], oracle_ref["Ir"]

    scan_sha = hashlib.sha256(open(os.path.join(ROOT, "sessionsimilaritysearch_amd", "csrc", "scan.hip"), "rb").read()).hexdigest()[:16]

    def run_leg(scan):
        """Time one configuration of the candidate scan over the shared corpus / weights / query batch."""
        index = FlatIndex(d, "ip", device, dtype=args.dtype, scan=scan if args.dtype == "f32" else None).adopt(xb, id_offset=lo)
        index.prepare(k)                         # images + norms now, not in the first timed search
        engine = HipEngine(index)
        sharded = ShardedFlatIndex(engine, device)

        def step_async():
            emb = embed()
            res = (emb,) + tuple(sharded.search_async(emb, k))
            if c3:
                res = res + knn_item_vote(res[1], res[2], session_items, k_items)
            return res

        def step_sync():
            emb = embed()
            res = (emb,) + tuple(sharded.search(emb, k)) + (None,)
            if c3:
                res = res + knn_item_vote(res[1], res[2], session_items, k_items)
            return res

        def timed_region(step):
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            engine.unproven.zero_()
            _lib.check(L.sss_profile_enable(1), "profile_enable")
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                res = step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            el = time.perf_counter() - t0
            unp = torch.tensor([float(engine.unproven.item()), el], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(unp[:1])
                dist.all_reduce(unp[1:], op=dist.ReduceOp.MAX)
            return res, float(unp[1].item()), int(unp[0].item())

        api = "search_async + on-device unproven counter"
        res, elapsed, unproven = timed_region(step_async)
        if unproven != 0:       # some query needed the exhaustive path: report the synchronous exact API instead
            tot_ms, launches = ctypes.c_double(0), ctypes.c_int(0)
            L.sss_profile_read(ctypes.byref(tot_ms), ctypes.byref(launches))
            api = "search (synchronous exact API; %d queries were unproven in the async run)" % unproven
            res, elapsed, _ = timed_region(step_sync)
        emb, D, I = res[:3]
        tot_ms, launches = ctypes.c_double(0), ctypes.c_int(0)
        _lib.check(L.sss_profile_read(ctypes.byref(tot_ms), ctypes.byref(launches)), "profile_read")
        L.sss_profile_enable(0)
        kern_ms = tot_ms.value / max(1, launches.value)
        flop_per_launch = 2.0 * nq * (hi - lo) * d                 # 2*d FLOP per (query, corpus row) pair
        achieved = flop_per_launch / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0

        # stage breakdown (outside the timed region)
        embed_ms = timed(embed)
        search_ms = timed(lambda: sharded.search_async(emb, k))
        vote_ms = timed(lambda: knn_item_vote(D, I, session_items, k_items)) if c3 else None

        # exactness: recall@10 / id equality against the oracle (canonical scores of the stored vectors)
        Dr, Ir = oracle_topk(emb)
        I_got, D_got = I[:nrq].cpu().numpy(), D[:nrq].cpu().numpy()
        items_exact = None
        if c3:      # aggregated top-10 items of the checked queries against the oracle's get_prediction_by_knn
            ptr, its = session_items.ptr.cpu().numpy(), session_items.items.cpu().numpy()
            lists = {int(s_): its[ptr[s_]:ptr[s_ + 1]] for s_ in np.unique(Ir[Ir >= 0])}
            got_items = res[4][:nrq].cpu().numpy()
            items_exact = all([int(v) for v in got_items[r] if v >= 0] == sr.knn_item_vote(Dr[r], Ir[r], lists, k_items)
                              for r in range(nrq))

        mode = index.last_scan          # the scan the timed searches used
        split, f16 = mode == "split", mode == "f16"
        traffic = traffic_detail = None
        tpath = os.path.join(ROOT, "profiles", "r03_traffic.json")
        if os.path.exists(tpath):       # per-launch HBM bytes measured by the committed rocprofv3 --pmc passes
            with open(tpath) as f:
                tj = json.load(f)
            ent = tj.get(f"{args.dtype if mode in ('f32', 'native') else mode}:{d}:{nq}:{hi - lo}")
            if ent is not None and ent.get("scan_hip_sha16") == scan_sha:      # measured on THIS scan kernel source
                traffic_detail, traffic = ent, ent["total_bytes"]
        # Roofline of the dominant kernel.  `achieved` is algorithmic: 2*d FLOP per (query, corpus row) pair
        # (SURVEY.md section 8(d)).  The split scan spends three bf16 MFMA passes per pair-element, so the
        # ceiling of ITS algorithmic rate is the dense bf16 peak / 3; pipe_* are the executed MFMA FLOP.
        passes = 3 if split else 1
        if split:
            peak = round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1)
        elif f16:
            peak = BF16_MFMA_PEAK_TFLOPS            # the guide's dense f16 rate is the bf16 rate
        else:
            peak = FP32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
        scan_name = mode if (split or f16) else args.dtype
        image_bytes = (hi - lo) * d * (2 if f16 else 4 if split else 0)
        return {
            "value": round(nq * args.steps / elapsed, 1), "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "dtype": "bf16x3" if split else "f16" if f16 else args.dtype, "scan": mode, "timed_api": api,
            "recall_at_10": round(sr.recall_at_k(I_got, Ir, k), 6), "ids_bit_exact": bool(np.array_equal(I_got, Ir)),
            "max_score_err": float(np.abs(D_got - Dr).max()), "recall_queries_checked": nrq, "unproven_queries": unproven,
            "stage_ms": {"embed_normalize": round(embed_ms, 4), "score_topk_merge": round(search_ms, 4),
                         **({"item_vote": round(vote_ms, 4)} if c3 else {})},
            "items_bit_exact": items_exact,
            "index_bytes": {"rows": (hi - lo) * d * (4 if args.dtype == "f32" else 2), "scan_image": image_bytes},
            "arithmetic": ("candidate scan: f32 rows as bf16 hi|lo pairs, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 "
                           "(f32 accumulate); candidates re-scored in float64 from the float32 rows; per-query proof, "
                           "exhaustive exact fallback" if split else
                           "candidate scan: f32 rows and queries scaled by a power of two and rounded to float16, one pass of "
                           "v_mfma_f32_32x32x16_f16 (f32 accumulate); candidates re-scored in float64 from the float32 rows; "
                           "per-query proof from the measured rounding residuals, exhaustive exact fallback" if f16 else
                           "candidate scan on the %s MFMA over the stored %s rows; candidates re-scored in float64; per-query "
                           "proof, exhaustive exact fallback" % (args.dtype, args.dtype)),
            "roofline": {"bound": "mfma", "kernel": f"k_scan<{d * (2 if (args.dtype == 'bf16' or f16) else 4)},*,{scan_name}>",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "peak_note": ("dense bf16 MFMA peak 2500 / 3 passes" if split else
                                       "dense f16 MFMA peak (= the bf16 rate)" if f16 else
                                       "dense %s MFMA peak" % args.dtype) + " (MI355X_MICROARCH.md)",
                         "mfma_passes": passes, "pipe_achieved": round(achieved * passes, 2),
                         "pipe_peak": BF16_MFMA_PEAK_TFLOPS if (split or f16 or args.dtype == "bf16") else FP32_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_unit": "B per launch (HBM, rocprofv3 PMC)",
                         "traffic_detail": traffic_detail,
                         "kernel_ms": round(kern_ms, 4), "launches": launches.value,
                         "flop_per_launch": flop_per_launch},
        }

    # Legs: the default run times the reference-precision scan (f32 MFMA) AND the production default
    # (scan="auto") back to back; an explicit --scan / bf16 index / C3 times that one configuration.
    two_legs = args.scan == "both" and args.dtype == "f32" and not c3
    main_leg = run_leg("f32" if two_legs else ("auto" if args.scan == "both" else args.scan))
    fast_leg = run_leg("auto") if two_legs else None

    # ---- CPU baseline (rank 0, N = 1): the reference path restated on the host cores
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not c3:
        # the GPU box gives one-GPU jobs a CPU share of ~16 cores whatever os.cpu_count() says;
        # more threads than that only thrash (measured: 256 threads -> 100x slower)
        cores = max(1, min(len(os.sched_getaffinity(0)), 16))
        torch.set_num_threads(cores)
        corpus_cpu = xb.float().cpu().numpy()
        qb_cpu = q_host.to_torch("cpu")

        def clock(fn, reps):
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = fn()
            return (time.perf_counter() - t0) / reps, out
        t_embed, e = clock(lambda: gnn_ref.encoder_forward(qb_cpu, weights, cfg.n_layers, self_loops=False).numpy(), 2)
        qn = sr.normalize(e)
        sample = min(n_total, 1 << 20)                      # the whole 1M default corpus; larger corpora: a 1M-row sample, scaled linearly
        t_search, _ = clock(lambda: sr.search_fp32_blocked(qn, corpus_cpu[:sample], k, block=16384, threads=cores), 2)
        t_full = t_embed + t_search * (n_total / sample)
        log(rank, f"cpu baseline: embed {t_embed:.3f}s, search {t_search:.3f}s on {sample} rows, {cores} threads")
        cpu = {"value": round(nq / t_full, 1), "unit": "queries/s", "cores": cores, "kind": "port",
               "sample": f"{nq} query sessions embedded by the torch-CPU oracle encoder ({t_embed:.3f}s) + blocked "
                         f"float32 SGEMM/top-k (faiss-shaped) over {sample} of {n_total} corpus rows "
                         f"({t_search:.3f}s" + (", scaled linearly to the full corpus)" if sample < n_total else ")")}

    ref_shapes = None
    if rank == 0 and world == 1 and not c3 and args.scan == "both" and args.dtype == "f32" and not args.no_reference_shapes:
        xb = None                           # free the main corpus (the indexes of the two legs went with their closures)
        torch.cuda.empty_cache()
        ref_shapes = reference_shapes_leg(device, nq, sr, lambda msg: log(rank, msg))

    if rank == 0:
        m = main_leg
        line = {
            "metric": "session queries/sec", "value": m["value"], "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": m["dtype"],
            "data": "synthetic",
            "config": {"workload": (f"{n_sessions}-session corpus x 4 prefix sub-sessions = {n_total} vectors d={d}, GNN embed + "
                                    f"cosine top-{k} neighbours + neighbour item vote -> top-{k_items} items, query batch {nq}"
                                    if c3 else
                                    f"{n_total}-session corpus d={d} ({args.corpus_source}), GNN embed (2-layer "
                                    f"HeteroGGNN + positional-attention pooling + normalise) + cosine top-{k}, query batch {nq}")
                                   + "; query graphs prepared (CSR, on device) before the timed region",
                       "timed_api": m["timed_api"], "scan": m["scan"],
                       "corpus_rows": n_total, "rows_per_gpu": hi - lo, "d": d, "k": k, "query_batch": nq,
                       "index_bytes_per_gpu": m["index_bytes"],
                       "parallelism": f"corpus row-sharded x{world}; nq/{world} sessions embedded per rank; all-gather of embeddings, all-gather of results + merge" if world > 1 else "single GPU"},
            "recall_at_10": m["recall_at_10"], "ids_bit_exact": m["ids_bit_exact"], "max_score_err": m["max_score_err"],
            "recall_queries_checked": m["recall_queries_checked"], "unproven_queries": m["unproven_queries"],
            "stage_ms": m["stage_ms"],
            **({"c3": {"sessions": n_sessions, "index_rows": n_total, "sample_size": k, "items_returned": k_items,
                       "items_bit_exact": m["items_bit_exact"], "queries_checked": nrq}} if c3 else {}),
            "arithmetic": m["arithmetic"],
            "roofline": m["roofline"],
            "cpu_baseline": cpu,
        }
        if ref_shapes is not None:
            line["reference_shapes"] = ref_shapes
        if fast_leg is not None:
            f = fast_leg
            line["fast_path"] = {
                "note": "production default (scan='auto') on the same corpus, weights and query batch, same run; identical "
                        "canonical results (float64 re-score from the float32 rows + per-query proof)",
                "dtype": f["dtype"], "scan": f["scan"], "value": f["value"], "unit": "queries/s", "ms_per_step": f["ms_per_step"],
                "timed_api": f["timed_api"], "recall_at_10": f["recall_at_10"], "ids_bit_exact": f["ids_bit_exact"],
                "max_score_err": f["max_score_err"], "recall_queries_checked": f["recall_queries_checked"],
                "unproven_queries": f["unproven_queries"], "stage_ms": f["stage_ms"],
                "index_bytes_per_gpu": f["index_bytes"], "arithmetic": f["arithmetic"], "roofline": f["roofline"],
                "speedup_vs_reference_precision": round(f["value"] / m["value"], 3)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

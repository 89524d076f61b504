#!/usr/bin/env python3
"""Headline benchmark: session queries/sec + recall@10, 1M-session corpus, d=128.

One "step" = one pass of the hot path over one batch of 1024 synthetic query sessions that are
already resident in HBM as a prepared batched graph (CSR adjacencies + pooling indices: the
reference builds its graphs on the host before model.forward too, test_amazon_filterd.py:546-551):
GNN embed (HeteroGGNN x2 -> positional-attention pooling -> L2-normalise, 6 launches) -> fused MFMA
scoring + top-10 against this rank's corpus shard (2 launches) -> (N > 1) RCCL all-gather of the
packed per-shard results -> merge.  `value_incl_graph_build` / `stage_ms.graph_build` additionally
report the step that STARTS from the flat action table (native graph build, csrc/graphbuild.hip).

The default run times TWO legs over the same corpus, weights and query batch, back to back, and
prints both in the one JSON line:
  * the top-level line is the REFERENCE-PRECISION leg: the candidate scan runs on the f32 MFMA over
    the float32 rows (`dtype: "f32"`, faiss IndexFlatIP scores in float32; test_amazon_filterd.py:578),
    roofline against the 157.3 TFLOP/s f32 matrix peak;
  * `fast_path` is the production default (`scan="auto"`: one f16 MFMA pass over a scaled float16
    image of the same corpus for k <= 128, the bf16 split image up to k = 500, the f32 scan by
    escalation only), with its own ms_per_step / value / roofline / exactness keys.
Both legs return the SAME canonical results (float64 re-score of the candidates from the float32 rows
+ per-query proof); `--scan X` / `--dtype bf16` / `--workload c3` time that single configuration only.

The default run (N = 1) also appends
  * `configs`: BASELINE.json's other single-GPU-sized configurations, each timed in this same process
    with its own corpus (freed before the next), hipEvent `kernel_ms`, `roofline`, oracle-checked ids:
    `c4` = 10M random unit rows x 128, nq 1024, k 10, f32-MFMA leg (the `north_star` target: >= 0.70 of
    the fp32 MFMA roofline) + scan="auto" leg; `c5` = 10M x 256 bf16, nq 4096; `c3` = 1M sessions x 4
    prefix sub-sessions, top-500 + neighbour item vote -> top-10 items (`--no-configs` skips them);
  * `reference_shapes`: the deployed model's OWN shapes -- the long-row search at D = 1600, K = 100 over
    1M random rows and the encoder at d_in 768 / h 800 / 3 layers / D 1600 (`--no-reference-shapes`).

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
`--force-collectives` (N = 1): a one-rank RCCL group is created and the step takes the multi-rank
route -- all-gather of the query embeddings, all-gather of the packed result, k_topk_merge -- so the
exchange code executes and its cost is measured (`collectives_ms`).

Extra JSON objects (see DESIGN.md "measurement"):
  roofline     -- dominant kernel k_scan<row bytes, tile rows, scan type>: algorithmic FLOPs per launch
                  / its mean duration, hipEvent-timed on its own stream inside the timed region;
                  `traffic` = HBM bytes per launch LOOKED UP from the committed rocprofv3 PMC passes of
                  this command (profiles/rNN_traffic.json; FETCH_SIZE doubled per MI355X_MICROARCH.md
                  "HBM"), printed only while the entry's `scan_hip_sha16` still equals the hash of
                  the csrc/scan.hip being run -- null otherwise (a stale figure is worse than none).
  cpu_baseline -- the oracle's restatement of the reference CPU path (torch CPU encoder +
                  faiss-shaped blocked SGEMM/top-k search) on this host's cores, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes
import glob
import hashlib
import json
import os
import sys
import time
from dataclasses import dataclass

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

BLOCK = 32768                    # sessions generated / embedded per block (seeded per block)
FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense f32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


@dataclass
class Spec:
    """One workload: which corpus, which query batch, which index element type."""
    name: str
    config_index: int           # seeds: SURVEY.md 8(d) (20260000 + config index, 1234 + config index)
    corpus_rows: int            # sessions (c3: sessions, the index holds 4x as many prefix vectors)
    source: str                 # "sessions" (GNN-embedded synthetic sessions) | "random" (N(0,1) unit rows, on device)
    d: int = 128
    nq: int = 1024
    k: int = 10
    dtype: str = "f32"
    workload: str = "search"    # "search" | "c3"
    sample_size: int = 500      # c3: neighbours searched before the item vote
    recall_queries: int = 256


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


def build_corpus_shard(enc, cfg, n_total, lo, hi, device, source, config_index):
    """Normalised session vectors of rows [lo, hi) of the corpus, on device."""
    out = torch.empty((hi - lo, cfg.d_out), dtype=torch.float32, device=device)
    if source == "random":          # config C4-style scoring corpus: N(0,1) rows, generated on device
        g = torch.Generator(device=device)
        for b0 in range(lo - lo % BLOCK, hi, BLOCK):
            g.manual_seed(20260000 + config_index * 100000 + b0 // BLOCK)
            blk = torch.randn((BLOCK, cfg.d_out), device=device, generator=g)
            a, b = max(lo, b0), min(hi, b0 + BLOCK, n_total)
            out[a - lo:b - lo] = blk[a - b0:b - b0]
        normalize_(out)
        return out
    for b0 in range(lo - lo % BLOCK, hi, BLOCK):
        nb = min(BLOCK, n_total - b0)
        acts = S.synthetic_actions(nb, 20260000 + config_index * 100000 + b0 // BLOCK, cfg.n_items, cfg.n_query)
        a, b = max(lo, b0), min(hi, b0 + nb)
        if a > b0 or b < b0 + nb:
            acts = acts.slice(a - b0, b - b0)
        out[a - lo:b - lo] = enc(enc.prepare_actions(acts), l2_normalize=True)     # native graph build + fused encoder
    return out


def oracle_topk_chunked(sr, q_np, xb, k, lo, threads, chunk=2_000_000):
    """The oracle's canonical top-k of `q_np` over the device rows `xb` (global ids from `lo`): rows go to the
    host in chunks (bounded host memory at 10M rows), each chunk through oracle/search_exact.c, merged by the
    oracle's (score desc, id asc) merge."""
    Ds, Is = [], []
    for c0 in range(0, xb.shape[0], chunk):
        rows = xb[c0:c0 + chunk].float().cpu().numpy()
        D_, I_ = sr.search_exact(q_np, rows, min(k, rows.shape[0]), id_offset=lo + c0, threads=threads)
        if D_.shape[1] < k:
            pad = k - D_.shape[1]
            D_ = np.concatenate([D_, np.full((D_.shape[0], pad), sr.NEG_SENTINEL, np.float32)], 1)
            I_ = np.concatenate([I_, np.full((I_.shape[0], pad), -1, np.int64)], 1)
        Ds.append(D_); Is.append(I_)
    return (Ds[0], Is[0]) if len(Ds) == 1 else sr.merge_topk(Ds, Is, k)


def traffic_lookup(key, scan_sha):
    """Per-launch HBM bytes of the scan kernel from the newest committed rocprofv3 PMC summary (profiles/
    rNN_traffic.json) -- only an entry measured on THIS csrc/scan.hip (sha stamped into it) is reported."""
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        with open(tpath) as f:
            ent = json.load(f).get(key)
        if ent is not None and ent.get("scan_hip_sha16") == scan_sha:
            return ent, ent["total_bytes"], os.path.basename(tpath)
    return None, None, None


class Bench:
    def __init__(self, args, world, rank, device, dist):
        self.args, self.world, self.rank, self.device, self.dist = args, world, rank, device, dist
        self.L = _lib.lib()
        from oracle import search_ref as sr, gnn_ref       # checker + CPU baseline only
        self.sr, self.gnn_ref = sr, gnn_ref
        self.cores = max(1, min(len(os.sched_getaffinity(0)), 16))
        self.scan_sha = hashlib.sha256(open(os.path.join(ROOT, "sessionsimilaritysearch_amd", "csrc", "scan.hip"), "rb").read()).hexdigest()[:16]
        self.force = bool(args.force_collectives)

    def say(self, *a):
        log(self.rank, *a)

    # ---------------------------------------------------------------------------------------------
    def run(self, sp: Spec, scans, steps, warmup, cpu_baseline=False, incl_graph=False):
        """Build the workload's corpus shard and query batch, time one leg per entry of `scans`, and return
        {"legs": [...], "cpu": ..., "meta": ...}.  Everything the workload allocated is released on return."""
        args, world, rank, device, dist, L, sr = self.args, self.world, self.rank, self.device, self.dist, self.L, self.sr
        d, nq = sp.d, sp.nq
        c3 = sp.workload == "c3"
        cfg = EncoderConfig(d_in=d, h=d, n_layers=2, d_out=d, self_loop_rule="none")
        weights = init_weights(cfg, 1234 + sp.config_index)
        enc = SessionEncoder(cfg, weights, device).eval()
        n_sessions = sp.corpus_rows
        n_total = 4 * n_sessions if c3 else n_sessions          # rows of the index
        lo, hi = shard_range(n_total, world, rank)
        t0 = time.time()
        session_items = None
        if c3:
            xb, session_items = build_c3_corpus(enc, cfg, n_sessions, device)
        else:
            xb = build_corpus_shard(enc, cfg, n_total, lo, hi, device, sp.source, sp.config_index)
        torch.cuda.synchronize()
        self.say(f"[{sp.name}] corpus shard rows [{lo},{hi}) x {d} built in {time.time() - t0:.1f}s")
        if sp.dtype == "bf16":
            xb = to_bf16(xb)
        # ---- query batch, resident in HBM
        q_acts = S.synthetic_actions(nq, 20269999, cfg.n_items, cfg.n_query)
        if c3:
            q_acts = q_acts.prefix(1, 2)             # the query is a prefix sub-session (test_amazon_filterd.py:546)
        q_host = S.build_batch(q_acts) if cpu_baseline else None      # host copy: only the CPU baseline reads it
        # every rank embeds nq / world of the query sessions; one all-gather hands everyone the batch
        q_lo, q_hi = query_slice(nq, world, rank)
        q_mine = q_acts.slice(q_lo, q_hi) if (q_lo, q_hi) != (0, nq) else q_acts
        qbatch = enc.prepare_actions(q_mine)         # batched CSR session graph built on device, resident in HBM
        up = lambda a, t: torch.from_numpy(np.ascontiguousarray(a)).to(device, t)
        q_table = type("DeviceActions", (), {})()    # the same action table as device tensors (graph-build-inclusive step)
        q_table.sess_ptr, q_table.is_search = up(q_mine.sess_ptr, torch.int64), up(q_mine.is_search, torch.uint8)
        q_table.item_id, q_table.query_tok = up(q_mine.item_id, torch.int64), up(q_mine.query_tok, torch.int64)
        emb_all = torch.empty((nq, d), dtype=torch.float32, device=device)
        k, k_items = sp.k, sp.k
        if c3:
            from sessionsimilaritysearch_amd.retrieval import knn_item_vote
            k = sp.sample_size                       # neighbours searched; k_items items voted

        def embed(pb=None):
            emb = gather_query_embeddings(enc(qbatch if pb is None else pb, l2_normalize=True), nq, emb_all,
                                          force_collective=self.force)
            return to_bf16(emb) if sp.dtype == "bf16" else emb

        def timed(fn, n=5):
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n

        # ---- exactness reference: the oracle's canonical top-k of the checked queries (computed once, all legs share it)
        nrq = min(sp.recall_queries, nq)
        oracle_ref = {}

        def oracle_topk(emb):
            if "Ir" not in oracle_ref:
                t1 = time.time()
                q_np = emb[:nrq].float().cpu().numpy()
                Dl, Il = oracle_topk_chunked(sr, q_np, xb, k, lo, self.cores)
                if world > 1:
                    pack = torch.cat([torch.from_numpy(Il).to(device).double(), torch.from_numpy(Dl).to(device).double()], 1)
                    allp = [torch.empty_like(pack) for _ in range(world)]
                    dist.all_gather(allp, pack)
                    Ds = [p[:, k:].float().cpu().numpy() for p in allp]
                    Is = [p[:, :k].long().cpu().numpy() for p in allp]
                    Dl, Il = sr.merge_topk(Ds, Is, k)
                oracle_ref["Dr"], oracle_ref["Ir"] = Dl, Il
                self.say(f"[{sp.name}] oracle top-{k} of {nrq} queries over {hi - lo} rows in {time.time() - t1:.1f}s")
            return oracle_ref["Dr"], oracle_ref["Ir"]

        def run_leg(scan):
            """Time one configuration of the candidate scan over the shared corpus / weights / query batch."""
            index = FlatIndex(d, "ip", device, dtype=sp.dtype, scan=scan if sp.dtype == "f32" else None).adopt(xb, id_offset=lo)
            index.prepare(k)                         # images + norms now, not in the first timed search
            engine = HipEngine(index)
            sharded = ShardedFlatIndex(engine, device, force_collectives=self.force)

            def step_async(pb=None):
                emb = embed(pb)
                res = (emb,) + tuple(sharded.search_async(emb, k))
                if c3:
                    res = res + knn_item_vote(res[1], res[2], session_items, k_items)
                return res

            def step_sync(pb=None):
                emb = embed(pb)
                res = (emb,) + tuple(sharded.search(emb, k)) + (None,)
                if c3:
                    res = res + knn_item_vote(res[1], res[2], session_items, k_items)
                return res

            def timed_region(step, n_steps, n_warm, profile=True):
                for _ in range(n_warm):
                    step()
                torch.cuda.synchronize()
                engine.unproven.zero_()
                L.sss_scan_boot_expired(1)           # reset the diagnostic counter (synchronises: outside the timed region)
                if profile:
                    _lib.check(L.sss_profile_enable(1), "profile_enable")
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_steps):
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
            step = step_async
            res, elapsed, unproven = timed_region(step_async, steps, warmup)
            if unproven != 0:       # some query needed the rung / exhaustive path: report the synchronous exact API instead
                tot_ms, launches = ctypes.c_double(0), ctypes.c_int(0)
                L.sss_profile_read(ctypes.byref(tot_ms), ctypes.byref(launches))
                api = "search (synchronous exact API; %d queries were unproven in the async run)" % unproven
                step = step_sync
                res, elapsed, _ = timed_region(step_sync, steps, warmup)
            # snapshot: D / I are the sharded index's own result buffers, which the stage timings below overwrite
            emb, D, I = res[0], res[1].clone(), res[2].clone()
            # waves whose bounded wait for the shared threshold ran out in the timed searches (sss.h: sss_scan_boot_expired):
            # anything but 0 means the launch's workgroups were not co-resident and the figure below is not the kernel's
            boot_expired = int(L.sss_scan_boot_expired(1))
            if boot_expired != 0:
                self.say(f"[{sp.name}] WARNING: bootstrap wait expired in {boot_expired} waves during the timed region")
            rung_q, exhaustive_q = index.last_rescan_queries, index.last_fallback_queries     # of the last (synchronous) step
            tot_ms, launches = ctypes.c_double(0), ctypes.c_int(0)
            _lib.check(L.sss_profile_read(ctypes.byref(tot_ms), ctypes.byref(launches)), "profile_read")
            L.sss_profile_enable(0)
            kern_ms = tot_ms.value / max(1, launches.value)
            flop_per_launch = 2.0 * nq * (hi - lo) * d                 # 2*d FLOP per (query, corpus row) pair
            achieved = flop_per_launch / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0

            # the same step STARTING FROM THE ACTION TABLE: native graph build (two kernel sweeps + one 5-integer
            # read-back that sizes the outputs) -> embed -> search; (f)1 of SURVEY section 8 inside the timed step
            incl = None
            if incl_graph:
                n_g = max(5, min(steps, 50))
                _, el_g, _ = timed_region(lambda: step(enc.prepare_actions(q_table)), n_g, 2, profile=False)
                incl = {"value": round(nq * n_g / el_g, 1), "ms_per_step": round(el_g / n_g * 1e3, 4), "steps": n_g}

            # stage breakdown (outside the timed region)
            embed_ms = timed(embed)
            search_ms = timed(lambda: sharded.search_async(emb, k))
            vote_ms = timed(lambda: knn_item_vote(D, I, session_items, k_items)) if c3 else None
            graph_ms = timed(lambda: enc.prepare_actions(q_table)) if incl_graph else None
            coll = None
            if sharded.exchange:        # the two all-gathers + the merge on their own (hipEvent pair on the stream they run on)
                chunk, pack, pack_all, _, _, _, Do, Io = sharded._buffers(nq, k)
                mine = emb_all[q_lo:q_hi].clone()
                coll = {"gather_embeddings": round(timed(lambda: gather_query_embeddings(mine, nq, emb_all, force_collective=self.force), 20), 4),
                        "gather_results": round(timed(lambda: dist.all_gather_into_tensor(pack_all, pack), 20), 4),
                        "merge": round(timed(lambda: engine.merge(pack_all, chunk, sharded.world, nq, k, Do, Io), 20), 4),
                        "backend": dist.get_backend(), "ranks": sharded.world}

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
            traffic_detail, traffic, traffic_file = traffic_lookup(
                f"{sp.dtype if mode in ('f32', 'native') else mode}:{d}:{nq}:{hi - lo}", self.scan_sha)
            # Roofline of the dominant kernel.  `achieved` is algorithmic: 2*d FLOP per (query, corpus row) pair
            # (SURVEY.md section 8(d)).  The split scan spends three bf16 MFMA passes per pair-element, so the
            # ceiling of ITS algorithmic rate is the dense bf16 peak / 3; pipe_* are the executed MFMA FLOP.
            passes = 3 if split else 1
            if split:
                peak = round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1)
            elif f16:
                peak = BF16_MFMA_PEAK_TFLOPS            # the guide's dense f16 rate is the bf16 rate
            else:
                peak = FP32_MFMA_PEAK_TFLOPS if sp.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
            scan_name = mode if (split or f16) else sp.dtype
            image_bytes = (hi - lo) * d * (2 if f16 else 4 if split else 0)
            stage = {"embed_normalize": round(embed_ms, 4), "score_topk_merge": round(search_ms, 4)}
            if c3:
                stage["item_vote"] = round(vote_ms, 4)
            if graph_ms is not None:
                stage["graph_build"] = round(graph_ms, 4)
            return {
                "value": round(nq * steps / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 4),
                "dtype": "bf16x3" if split else "f16" if f16 else sp.dtype, "scan": mode, "timed_api": api,
                "recall_at_10": round(sr.recall_at_k(I_got, Ir, k), 6), "ids_bit_exact": bool(np.array_equal(I_got, Ir)),
                "max_score_err": float(np.abs(D_got - Dr).max()), "recall_queries_checked": nrq, "unproven_queries": unproven,
                "bootstrap_wait_expired": boot_expired, "stage_ms": stage, "incl_graph_build": incl, "collectives_ms": coll,
                "last_step_rung_queries": rung_q if step is step_sync else None,
                "last_step_exhaustive_queries": exhaustive_q if step is step_sync else None,
                "items_bit_exact": items_exact,
                "index_bytes": {"rows": (hi - lo) * d * (4 if sp.dtype == "f32" else 2), "scan_image": image_bytes},
                "arithmetic": ("candidate scan: f32 rows as bf16 hi|lo pairs, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 "
                               "(f32 accumulate); candidates re-scored in float64 from the float32 rows; per-query proof, "
                               "exhaustive exact fallback" if split else
                               "candidate scan: f32 rows and queries scaled by a power of two and rounded to float16, one pass of "
                               "v_mfma_f32_32x32x16_f16 (f32 accumulate); candidates re-scored in float64 from the float32 rows; "
                               "per-query proof from the measured rounding residuals, exhaustive exact fallback" if f16 else
                               "candidate scan on the %s MFMA over the stored %s rows; candidates re-scored in float64; per-query "
                               "proof, exhaustive exact fallback" % (sp.dtype, sp.dtype)),
                "roofline": {"bound": "mfma", "kernel": f"k_scan<{d * (2 if (sp.dtype == 'bf16' or f16) else 4)},*,{scan_name}>",
                             "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                             "frac": round(achieved / peak, 4),
                             "peak_note": ("dense bf16 MFMA peak 2500 / 3 passes" if split else
                                           "dense f16 MFMA peak (= the bf16 rate)" if f16 else
                                           "dense %s MFMA peak" % sp.dtype) + " (MI355X_MICROARCH.md)",
                             "mfma_passes": passes, "pipe_achieved": round(achieved * passes, 2),
                             "pipe_peak": BF16_MFMA_PEAK_TFLOPS if (split or f16 or sp.dtype == "bf16") else FP32_MFMA_PEAK_TFLOPS,
                             "traffic": traffic,
                             "traffic_unit": "B per launch (HBM; not measured in this run: looked up from the committed rocprofv3 "
                                             "PMC passes under profiles/, hash-checked against the csrc/scan.hip being run)",
                             "traffic_file": traffic_file, "traffic_detail": traffic_detail,
                             "kernel_ms": round(kern_ms, 4), "launches": launches.value,
                             "flop_per_launch": flop_per_launch},
            }

        legs = [run_leg(s) for s in scans]

        # ---- CPU baseline (rank 0, N = 1): the reference path restated on the host cores
        cpu = None
        if cpu_baseline:
            # the GPU box gives one-GPU jobs a CPU share of ~16 cores whatever os.cpu_count() says;
            # more threads than that only thrash (measured: 256 threads -> 100x slower)
            cores = self.cores
            torch.set_num_threads(cores)
            corpus_cpu = xb.float().cpu().numpy()
            qb_cpu = q_host.to_torch("cpu")

            def clock(fn, reps):
                fn()
                t0 = time.perf_counter()
                for _ in range(reps):
                    out = fn()
                return (time.perf_counter() - t0) / reps, out
            t_embed, e = clock(lambda: self.gnn_ref.encoder_forward(qb_cpu, weights, cfg.n_layers, self_loops=False).numpy(), 2)
            qn = sr.normalize(e)
            sample = min(n_total, 1 << 20)                      # the whole 1M default corpus; larger corpora: a 1M-row sample, scaled linearly
            t_search, _ = clock(lambda: sr.search_fp32_blocked(qn, corpus_cpu[:sample], k, block=16384, threads=cores), 2)
            t_full = t_embed + t_search * (n_total / sample)
            self.say(f"cpu baseline: embed {t_embed:.3f}s, search {t_search:.3f}s on {sample} rows, {cores} threads")
            cpu = {"value": round(nq / t_full, 1), "unit": "queries/s", "cores": cores, "kind": "port",
                   "sample": f"{nq} query sessions embedded by the torch-CPU oracle encoder ({t_embed:.3f}s) + blocked "
                             f"float32 SGEMM/top-k (faiss-shaped) over {sample} of {n_total} corpus rows "
                             f"({t_search:.3f}s" + (", scaled linearly to the full corpus)" if sample < n_total else ")")}
        meta = {"n_sessions": n_sessions, "n_total": n_total, "rows_per_gpu": hi - lo, "k": k, "k_items": k_items, "nrq": nrq}
        return {"legs": legs, "cpu": cpu, "meta": meta}

    # ---------------------------------------------------------------------------------------------
    def workload_text(self, sp, meta):
        if sp.workload == "c3":
            return (f"{meta['n_sessions']}-session corpus x 4 prefix sub-sessions = {meta['n_total']} vectors d={sp.d}, GNN embed + "
                    f"cosine top-{meta['k']} neighbours + neighbour item vote -> top-{meta['k_items']} items, query batch {sp.nq}"
                    "; query graphs prepared (CSR, on device) before the timed region")
        return (f"{meta['n_total']}-session corpus d={sp.d} ({sp.source}), GNN embed (2-layer "
                f"HeteroGGNN + positional-attention pooling + normalise) + cosine top-{meta['k']}, query batch {sp.nq}"
                "; query graphs prepared (CSR, on device) before the timed region")

    def leg_summary(self, leg):
        """A leg's keys for a `configs` / `fast_path` object."""
        keys = ("dtype", "scan", "value", "ms_per_step", "timed_api", "recall_at_10", "ids_bit_exact", "max_score_err",
                "recall_queries_checked", "unproven_queries", "bootstrap_wait_expired", "stage_ms", "arithmetic", "roofline")
        out = {kk: leg[kk] for kk in keys}
        out["unit"] = "queries/s"
        out["index_bytes_per_gpu"] = leg["index_bytes"]
        if leg["items_bit_exact"] is not None:
            out["items_bit_exact"] = leg["items_bit_exact"]
        if leg["collectives_ms"] is not None:
            out["collectives_ms"] = leg["collectives_ms"]
        if leg["last_step_rung_queries"] is not None:
            out["last_step_rung_queries"] = leg["last_step_rung_queries"]
            out["last_step_exhaustive_queries"] = leg["last_step_exhaustive_queries"]
        return out

    def configs_legs(self, steps, warmup):
        """BASELINE.json's other single-GPU-sized configurations, one after the other, each with its own corpus."""
        out = {"note": f"same process as the headline line, {steps} timed steps (+{warmup} warm-up) per leg, one corpus resident at a time; "
                       "ids / scores checked against the oracle's canonical top-k (items against its neighbour vote for c3)"}
        plan = [
            ("c4", Spec("c4", 4, 10_000_000, "random", d=128, nq=1024, k=10, recall_queries=16), ["f32", "auto"]),
            ("c5", Spec("c5", 5, 10_000_000, "random", d=256, nq=4096, k=10, dtype="bf16", recall_queries=16), ["native"]),
            ("c3", Spec("c3", 3, 1_000_000, "sessions", d=128, nq=1024, k=10, workload="c3", recall_queries=8), ["auto"]),
        ]
        for name, sp, scans in plan:
            t0 = time.time()
            r = self.run(sp, scans, steps, warmup)
            torch.cuda.empty_cache()
            ent = {"workload": self.workload_text(sp, r["meta"]), "corpus_rows": r["meta"]["n_total"], "d": sp.d, "k": r["meta"]["k"],
                   "query_batch": sp.nq, "steps": steps, "warmup": warmup}
            ent.update(self.leg_summary(r["legs"][0]))
            if name == "c3":
                ent["c3"] = {"sessions": r["meta"]["n_sessions"], "index_rows": r["meta"]["n_total"], "sample_size": r["meta"]["k"],
                             "items_returned": r["meta"]["k_items"], "queries_checked": r["meta"]["nrq"]}
            if len(r["legs"]) > 1:
                ent["fast_path"] = self.leg_summary(r["legs"][1])
                ent["fast_path"]["speedup_vs_reference_precision"] = round(r["legs"][1]["value"] / r["legs"][0]["value"], 3)
            ent["wall_s"] = round(time.time() - t0, 1)
            out[name] = ent
            self.say(f"[{name}] done in {ent['wall_s']}s: {ent['ms_per_step']} ms/step, roofline.frac {ent['roofline']['frac']}, "
                     f"ids_bit_exact {ent['ids_bit_exact']}")
        return out


def reference_shapes_leg(device, nq, sr, say):
    """The deployed model's OWN shapes (pretrain_filtered_amazon.py:267,281; config.py:15-16,21;
    test_amazon_filterd.py:459,488,578): session vectors of D = 1600, K = 100 neighbours, and the encoder at
    d_in 768 / h 800 / 3 layers.  Not the metric's configuration (BASELINE.json quotes d = 128, k = 10) -- an extra
    object in the line so that a driver-run record carries these figures too.  Scoring: 1M random unit rows x 1600,
    the K-tiled long-row scan (csrc/scan_long.hip); a few queries are checked against the oracle's canonical search."""
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
                           "note": "whole search (sampled threshold levels + full scan + bounds + final re-score) / algorithmic FLOPs "
                                   "(DESIGN.md 5.4)"}}
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
                         "rows; 'auto' = f16 for k <= 128, split up to k = 500, f32 by escalation only; 'both' (default) = the "
                         "f32 leg as the top-level line and the 'auto' leg as `fast_path`.  Results are identical.")
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--workload", choices=["search", "c3"], default="search",
                    help="c3: 4 prefix sub-sessions per session indexed, top-500 neighbours -> item vote -> top-10 items (1 GPU)")
    ap.add_argument("--sample-size", type=int, default=500)
    ap.add_argument("--no-reference-shapes", action="store_true",
                    help="skip the extra leg at the deployed model's own shapes (D = 1600, K = 100; encoder 768/800/3/1600)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` object (C4 10M x 128 f32 + auto legs, C5 10M x 256 bf16, C3 1M sessions)")
    ap.add_argument("--config-steps", type=int, default=5, help="timed steps per leg of the `configs` object")
    ap.add_argument("--config-index", type=int, default=2,
                    help="seed family of the weights / corpus (SURVEY.md 8(d): 1234 + index, 20260000 + index * 100000); the "
                         "`configs` object uses 4 / 5 / 3 for C4 / C5 / C3")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: create a one-rank RCCL group and run the all-gather + merge route of the multi-rank path")
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
    if world > 1 or args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        kw = {} if world > 1 else {"world_size": 1, "rank": 0}
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, **kw)
        else:
            dist.init_process_group(backend, **kw)

    c3 = args.workload == "c3"
    if c3 and (world != 1 or args.dtype != "f32"):
        raise SystemExit("--workload c3 is the single-GPU f32 configuration")
    B = Bench(args, world, rank, device, dist)
    sp = Spec("main", args.config_index, args.corpus_rows, args.corpus_source, d=args.d, nq=args.nq, k=args.k, dtype=args.dtype,
              workload=args.workload, sample_size=args.sample_size,
              recall_queries=args.recall_queries if not c3 else min(8, args.recall_queries))

    # Legs: the default run times the reference-precision scan (f32 MFMA) AND the production default
    # (scan="auto") back to back; an explicit --scan / bf16 index / C3 times that one configuration.
    two_legs = args.scan == "both" and args.dtype == "f32" and not c3
    scans = ["f32", "auto"] if two_legs else [("auto" if args.scan == "both" else args.scan) if args.dtype == "f32" else "native"]
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and not c3
    r = B.run(sp, scans, args.steps, args.warmup, cpu_baseline=want_cpu, incl_graph=not c3)
    main_leg, fast_leg = r["legs"][0], (r["legs"][1] if two_legs else None)
    meta, cpu = r["meta"], r["cpu"]
    torch.cuda.empty_cache()

    extras = rank == 0 and world == 1 and two_legs
    configs = ref_shapes = None
    if extras and not args.no_configs:
        configs = B.configs_legs(args.config_steps, 2)
    if extras and not args.no_reference_shapes:
        ref_shapes = reference_shapes_leg(device, args.nq, B.sr, lambda msg: log(rank, msg))

    if rank == 0:
        m = main_leg
        line = {
            "metric": "session queries/sec", "value": m["value"], "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": m["dtype"],
            "data": "synthetic",
            "config": {"workload": B.workload_text(sp, meta),
                       "timed_api": m["timed_api"], "scan": m["scan"],
                       "corpus_rows": meta["n_total"], "rows_per_gpu": meta["rows_per_gpu"], "d": sp.d, "k": meta["k"], "query_batch": sp.nq,
                       "index_bytes_per_gpu": m["index_bytes"],
                       "parallelism": (f"corpus row-sharded x{world}; nq/{world} sessions embedded per rank; all-gather of embeddings, "
                                       "all-gather of results + merge") if world > 1 else
                                      ("single GPU, one-rank RCCL group: all-gather of embeddings, all-gather of results + merge executed"
                                       if args.force_collectives else "single GPU")},
            "recall_at_10": m["recall_at_10"], "ids_bit_exact": m["ids_bit_exact"], "max_score_err": m["max_score_err"],
            "recall_queries_checked": m["recall_queries_checked"], "unproven_queries": m["unproven_queries"],
            "bootstrap_wait_expired": m["bootstrap_wait_expired"], "stage_ms": m["stage_ms"],
            **({"value_incl_graph_build": m["incl_graph_build"]["value"],
                "ms_per_step_incl_graph_build": m["incl_graph_build"]["ms_per_step"],
                "incl_graph_build_note": f"the same step started from the flat action table (native graph build on the device + its two small "
                                         f"host read-backs, then embed + search), {m['incl_graph_build']['steps']} timed steps"}
               if m["incl_graph_build"] else {}),
            **({"collectives_ms": m["collectives_ms"]} if m["collectives_ms"] else {}),
            **({"c3": {"sessions": meta["n_sessions"], "index_rows": meta["n_total"], "sample_size": meta["k"], "items_returned": meta["k_items"],
                       "items_bit_exact": m["items_bit_exact"], "queries_checked": meta["nrq"],
                       "last_step_rung_queries": m["last_step_rung_queries"],
                       "last_step_exhaustive_queries": m["last_step_exhaustive_queries"]}} if c3 else {}),
            "arithmetic": m["arithmetic"],
            "roofline": m["roofline"],
            "cpu_baseline": cpu,
        }
        if fast_leg is not None:
            f = B.leg_summary(fast_leg)
            f["note"] = ("production default (scan='auto') on the same corpus, weights and query batch, same run; identical "
                         "canonical results (float64 re-score from the float32 rows + per-query proof)")
            f["speedup_vs_reference_precision"] = round(fast_leg["value"] / m["value"], 3)
            if fast_leg["incl_graph_build"]:
                f["value_incl_graph_build"] = fast_leg["incl_graph_build"]["value"]
                f["ms_per_step_incl_graph_build"] = fast_leg["incl_graph_build"]["ms_per_step"]
            line["fast_path"] = f
        if configs is not None:
            line["configs"] = configs
        if ref_shapes is not None:
            line["reference_shapes"] = ref_shapes
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

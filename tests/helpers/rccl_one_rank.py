"""Child process of tests/test_distributed_gpu.py: the multi-rank exchange route of the search on a ONE-rank RCCL
group ("nccl" on ROCm).  Started as a fresh process, so the process group is created before anything else of this
process touches the GPU.  Prints one JSON line; exit code 0 = every check passed.

What runs: init_process_group("nccl", world_size=1, rank=0, device_id=cuda:0) -> SessionEncoder forward ->
gather_query_embeddings(force_collective=True) (RCCL all_gather_into_tensor) -> ShardedFlatIndex(force_collectives=True)
.search_async / .search (local fused search, RCCL all-gather of the packed int64 (ids | scores) block, k_topk_merge) ->
compared with the oracle's canonical search; the two collectives and the merge are hipEvent-timed on the stream."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
    torch.cuda.set_device(dev)
    from oracle import search_ref as sr
    from sessionsimilaritysearch_amd import sessions as S
    from sessionsimilaritysearch_amd.distributed import HipEngine, ShardedFlatIndex, gather_query_embeddings
    from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, init_weights
    from sessionsimilaritysearch_amd.index import FlatIndex, normalize_

    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    nq, d, k, n = 1024, 128, 10, 200_000
    cfg = EncoderConfig(d_in=d, h=d, n_layers=2, d_out=d, n_items=5000, n_query=129, self_loop_rule="none")
    enc = SessionEncoder(cfg, init_weights(cfg, 7), dev).eval()
    pb = enc.prepare_actions(S.synthetic_actions(nq, 11, cfg.n_items, cfg.n_query))
    emb_local = enc(pb, l2_normalize=True)
    emb_all = torch.empty_like(emb_local)
    got = gather_query_embeddings(emb_local, nq, emb_all, force_collective=True)
    out["gather_is_collective_output"] = got.data_ptr() == emb_all.data_ptr()
    out["gathered_embeddings_equal"] = bool(torch.equal(got, emb_local))

    g = torch.Generator(device=dev); g.manual_seed(5)
    c = torch.randn((n, d), device=dev, generator=g); normalize_(c)
    c[1] = c[n - 2]                                         # an exact tie far apart in the corpus
    index = FlatIndex(d, "ip", dev).adopt(c, id_offset=1000)
    eng = HipEngine(index)
    sh = ShardedFlatIndex(eng, dev, force_collectives=True)
    out["exchange_route"] = bool(sh.exchange)
    D, I, status = sh.search_async(got, k)
    D2, I2 = sh.search(got, k)
    torch.cuda.synchronize()
    Dr, Ir = sr.search_exact(got.cpu().numpy(), c.cpu().numpy(), k, id_offset=1000)
    proven = (status == 0).cpu().numpy()
    out["async_ids_equal_where_proven"] = bool(np.array_equal(I.cpu().numpy()[proven], Ir[proven]))
    out["async_scores_equal_where_proven"] = bool(np.array_equal(D.cpu().numpy()[proven], Dr[proven]))
    out["unproven"] = int((~proven).sum())
    out["sync_ids_equal"] = bool(np.array_equal(I2.cpu().numpy(), Ir))
    out["sync_scores_equal"] = bool(np.array_equal(D2.cpu().numpy(), Dr))

    def timed(fn, reps=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / reps, 4)
    chunk, pack, pack_all, _, _, _, Do, Io = sh._buffers(nq, k)
    out["ms"] = {"all_gather_embeddings_512KB": timed(lambda: gather_query_embeddings(emb_local, nq, emb_all, force_collective=True)),
                 "all_gather_results_int64_pack_120KB": timed(lambda: dist.all_gather_into_tensor(pack_all, pack)),
                 "k_topk_merge_1_shard": timed(lambda: eng.merge(pack_all, chunk, 1, nq, k, Do, Io)),
                 "search_async_with_exchange": timed(lambda: sh.search_async(got, k)),
                 "local_search_only": timed(lambda: eng.local_search(got, k, *sh._buffers(nq, k)[3:6]))}
    ok = all(out[key] for key in ("gather_is_collective_output", "gathered_embeddings_equal", "exchange_route",
                                  "async_ids_equal_where_proven", "async_scores_equal_where_proven", "sync_ids_equal",
                                  "sync_scores_equal")) and out["backend"] == "nccl"
    out["ok"] = bool(ok)
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

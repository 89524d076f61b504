"""RCCL executed for real, with one rank: the exchange route of the row-sharded search (all-gather of the query
embeddings, all-gather of the packed results, k_topk_merge) on the GPU, checked against the oracle.  The collective
needs its process group before anything else touches the GPU, so it runs in a fresh child process
(tests/helpers/rccl_one_rank.py) started -- not exec'ed -- from here."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_rccl_one_rank_exchange_route_matches_oracle(cuda, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_one_rank.py"), str(_free_port())],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and lines, f"child failed (rc {res.returncode}):\n{res.stdout[-2000:]}\n{res.stderr[-4000:]}"
    out = json.loads(lines[-1])
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):                                  # the measured floor of the two collectives + merge
        with open(os.path.join(keep, "rccl_one_rank.json"), "w") as f:
            f.write(lines[-1] + "\n")
    assert out["ok"] and out["backend"] == "nccl" and out["world"] == 1
    assert out["exchange_route"] and out["gather_is_collective_output"] and out["gathered_embeddings_equal"]
    assert out["sync_ids_equal"] and out["sync_scores_equal"]
    assert out["async_ids_equal_where_proven"] and out["async_scores_equal_where_proven"]

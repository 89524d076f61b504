"""Generates tests/golden/encoder_tiny.npz: a tiny seeded session batch, random-init weights of the
reference architecture, and the encoder outputs computed by the INDEPENDENT float64 oracle
(oracle/gnn_ref64.py: plain numpy loops written from SURVEY.md Appendix A).

    python tests/golden/make_golden_encoder.py

The reference cannot produce these numbers itself (torch_geometric is absent), so this fixture pins
the build's two restatements against each other and the HIP encoder against both; parity with the
reference stays "unpinned" (DESIGN.md).  The .npz holds inputs and outputs only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import gnn_ref64  # noqa: E402
from sessionsimilaritysearch_amd import sessions as S  # noqa: E402
from sessionsimilaritysearch_amd.encoder import EncoderConfig, init_weights  # noqa: E402

cfg = EncoderConfig(d_in=32, h=32, n_layers=2, d_out=64, n_items=40, n_query=9)
w = init_weights(cfg, 20260777)
acts = S.synthetic_actions(6, 20260777, cfg.n_items, cfg.n_query)
batch = S.build_batch(acts)
out = {}
for loops in (True, False):
    o, nodes = gnn_ref64.encoder_forward(batch, w, cfg.n_layers, self_loops=loops, get_node=True)
    tag = "loops" if loops else "noloops"
    out[f"out_{tag}"] = o
    out[f"node_q_{tag}"] = nodes["query"]
    out[f"node_p_{tag}"] = nodes["product"]
np.savez_compressed(
    os.path.join(HERE, "encoder_tiny.npz"),
    sess_ptr=acts.sess_ptr, is_search=acts.is_search, item_id=acts.item_id, query_tok=acts.query_tok,
    cfg=np.array([cfg.d_in, cfg.h, cfg.n_layers, cfg.d_out, cfg.n_items, cfg.n_query], np.int64),
    **{"w:" + k: v.numpy() for k, v in w.items()}, **out)
print("wrote encoder_tiny.npz", out["out_loops"].shape)

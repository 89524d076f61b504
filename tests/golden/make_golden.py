"""Generates tests/golden/node_asin_embedding.npz from the reference's own NodeAsinEmbedding.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
model/NodeEmbedding.py is loaded by FILE PATH (the package __init__ imports torch_geometric,
which is absent); nothing from the reference is copied -- the .npz holds inputs and outputs.
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/model/NodeEmbedding.py"
HERE = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location("ref_node_embedding", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

torch.manual_seed(20260001)
emb = mod.NodeAsinEmbedding(nproducts=97, ninp=16)
ids = torch.tensor([0, 1, 96, 5, 5, 42, 0], dtype=torch.long)
with torch.no_grad():
    out = emb(ids)
np.savez(os.path.join(HERE, "node_asin_embedding.npz"), table=emb.encoder.weight.detach().numpy(),
         ids=ids.numpy(), out=out.numpy())
print("wrote node_asin_embedding.npz", out.shape)

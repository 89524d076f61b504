"""Generates tests/golden/reference_pure.npz by RUNNING the reference's own pure numpy / Python functions of the hot
path on seeded inputs.  Build container only (needs /root/reference):  python tests/golden/make_golden_pure.py

The files these functions live in cannot be imported here (`import torch_geometric`, `faiss`, `Levenshtein` at their
tops fail), but the FUNCTION BODIES need none of that: each `def` is located in its file's syntax tree, compiled on its
own into a namespace that holds only numpy / defaultdict, and called.  Nothing of the reference's text is written
anywhere -- the .npz holds inputs and the outputs the reference's code produced for them:

  normalize              util_amazon_filtered.py:28-31   rows / sqrt(clip(sum(v^2), 1e-6)); 1-D and 2-D
  normalize (fine-tune)  fine_tune_ours.py:38-40         v / (||v|| + 1e-4)
  get_p_r                test_amazon_filterd.py:80-85    precision / recall at K
  get_prediction_by_knn  test_amazon_filterd.py:59-78    neighbour-weighted item vote -> K heaviest items

`get_prediction_by_knn` searches through `index.search` and reads `dataset[i]['product'].x`; the stubs below supply
INPUTS only (a prepared (D, I) pair, and the item list of every indexed session) -- all arithmetic, accumulation
order, sorting and tie behaviour are the reference function's own.
"""
import ast
import os
import types
from collections import defaultdict

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def extract(path, name, nth=0):
    """The nth top-level `def name` of a reference file, compiled alone."""
    src = open(os.path.join(REF, path)).read()
    defs = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name]
    mod = ast.Module(body=[defs[nth]], type_ignores=[])
    ns = {"np": np, "defaultdict": defaultdict}
    exec(compile(mod, os.path.join(REF, path), "exec"), ns)
    return ns[name], (defs[nth].lineno, defs[nth].end_lineno)


normalize, ln_norm = extract("util_amazon_filtered.py", "normalize")
normalize_ft, ln_ft = extract("fine_tune_ours.py", "normalize")
get_p_r, ln_pr = extract("test_amazon_filterd.py", "get_p_r")
get_prediction_by_knn, ln_knn = extract("test_amazon_filterd.py", "get_prediction_by_knn")
print("extracted: normalize", ln_norm, "normalize(fine-tune)", ln_ft, "get_p_r", ln_pr, "get_prediction_by_knn", ln_knn)

out = {}
rng = np.random.default_rng(20260404)

# ---------------------------------------------------------------- normalize (both rules)
x32 = rng.standard_normal((64, 128)).astype(np.float32)
x32[3] = 0.0                                             # zero row: clip keeps it finite (0 / 1e-3)
x32[4] = (1e-5 * rng.standard_normal(128)).astype(np.float32)     # sum of squares ~1e-8 < 1e-6: the clip decides
x32[5] = (3e-4 * rng.standard_normal(128) / np.sqrt(128)).astype(np.float32)   # sum of squares just below the clip
x32[6] *= 1e6
x32[7] = 0.0; x32[7, 17] = -2.5                          # one non-zero element
out["norm_x32"] = x32
out["norm_y32"] = normalize(x32.copy())
x64 = rng.standard_normal((9, 20))
x64[2] = 0.0
out["norm_x64"] = x64
out["norm_y64"] = normalize(x64.copy())
x1600 = rng.standard_normal((7, 1600)).astype(np.float32)        # the deployed session-vector width
out["norm_x1600"] = x1600
out["norm_y1600"] = normalize(x1600.copy())
v1 = rng.standard_normal(37).astype(np.float32)          # 1-D branch (whole vector)
out["norm_v1"] = v1
out["norm_w1"] = normalize(v1.copy())
out["norm_ones4"] = normalize(np.ones(4))                # the value the reference prints (test_amazon_filterd.py:866)
vz = np.zeros(8, np.float32)
out["norm_wz"] = normalize(vz.copy())
out["normft_y32"] = normalize_ft(x32.copy())
out["normft_y64"] = normalize_ft(x64.copy())
out["normft_y1600"] = normalize_ft(x1600.copy())
for key in ("norm_y32", "norm_y1600", "norm_w1", "normft_y32", "normft_y1600"):
    assert out[key].dtype == np.float32, (key, out[key].dtype)

# ---------------------------------------------------------------- get_p_r
cases = [({1, 2, 3}, [3, 9, 1, 7, 2], 3), ({1, 2, 3}, [3, 9, 1, 7, 2], 5), ({5}, [1, 2, 3], 2), ({4, 8}, [8, 4], 10),
         ({7, 7 + 1}, [7, 7, 7, 8], 3), (set(range(30)), list(range(10, 50)), 20)]
for i, (gt, pred, K) in enumerate(cases):
    p, r = get_p_r(set(gt), list(pred), K)
    out[f"pr{i}_gt"], out[f"pr{i}_pred"], out[f"pr{i}_K"] = np.array(sorted(gt), np.int64), np.array(pred, np.int64), np.int64(K)
    out[f"pr{i}_out"] = np.array([p, r], np.float64)
out["pr_cases"] = np.int64(len(cases))


# ---------------------------------------------------------------- get_prediction_by_knn
class PreparedIndex:
    """Stands in for the faiss index: returns the prepared (D, I) -- an input of the case, no arithmetic."""

    def __init__(self, D, I):
        self.D, self.I = D, I

    def search(self, emb, sample_size):
        assert emb.shape[0] == 1 and sample_size == self.D.shape[1]
        return self.D, self.I


def vote_case(tag, n_sessions, S, K, n_items, tie_scores=False, seed=0):
    r = np.random.default_rng(seed)
    lens = r.integers(1, 9, n_sessions)
    ptr = np.zeros(n_sessions + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    items = np.concatenate([r.choice(np.arange(1, n_items), size=int(l), replace=False) for l in lens]).astype(np.int64)
    # the reference's dataset entries are PyG graphs whose ['product'].x is a torch LongTensor of item ids
    dataset = [{"product": types.SimpleNamespace(x=torch.from_numpy(items[ptr[s]:ptr[s + 1]].copy()))} for s in range(n_sessions)]
    I = r.choice(n_sessions, size=S, replace=False).astype(np.int64)[None, :]
    if tie_scores:      # a few distinct similarity values, exactly representable: equal item weights (ties in the final sort)
        D = r.choice(np.array([0.5, 0.25, 0.75, 1.0], np.float32), size=S)[None, :].astype(np.float32)
    else:
        D = np.sort(r.random(S).astype(np.float32))[::-1][None, :].copy()
    pred = get_prediction_by_knn(torch.zeros((1, 4)), PreparedIndex(D, I), dataset, S, K)
    out[f"vote_{tag}_D"], out[f"vote_{tag}_I"] = D[0], I[0]
    out[f"vote_{tag}_ptr"], out[f"vote_{tag}_items"] = ptr, items
    out[f"vote_{tag}_K"] = np.int64(K)
    out[f"vote_{tag}_pred"] = np.array([int(p) for p in pred], np.int64)
    print(f"vote {tag}: {len(pred)} items from {S} neighbours")


vote_case("a", 400, 100, 10, 300, seed=1)
vote_case("b", 3000, 500, 10, 2000, seed=2)              # config C3's sample_size
vote_case("ties", 200, 60, 20, 40, tie_scores=True, seed=3)   # few items, few score values: equal weights, first-seen order decides
vote_case("short", 50, 3, 20, 500, seed=4)               # fewer distinct items than K
out["vote_tags"] = np.array(["a", "b", "ties", "short"])

np.savez(os.path.join(HERE, "reference_pure.npz"), **out)
print("wrote reference_pure.npz:", len(out), "arrays")

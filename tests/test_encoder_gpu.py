"""GPU parity: HIP session encoder (through the C ABI) vs the CPU oracle.

Floating-point path: the tolerance is the one BASELINE.json's north_star states for scores,
1e-5 (absolute, on O(1) embeddings; relative 1e-5 on the L2-normalised vectors that reach the
index).  Each kernel is also checked on its own against a float64 torch restatement.
"""
import numpy as np
import pytest
import torch

from oracle import gnn_ref
from oracle import search_ref as sr
from sessionsimilaritysearch_amd import _lib
from sessionsimilaritysearch_amd import sessions as S
from sessionsimilaritysearch_amd.encoder import EncoderConfig, SessionEncoder, build_csr, init_weights

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _st(dev):
    return _lib.stream_ptr(dev)


@pytest.mark.parametrize("n,m,k", [(1, 5, 32), (130, 108, 384), (1000, 672, 128), (257, 160, 64), (64, 128, 800)])
def test_linear_matches_float64(cuda, n, m, k):
    g = torch.Generator().manual_seed(n * 31 + m)
    x, w, b = torch.randn((n, k), generator=g), torch.randn((m, k), generator=g), torch.randn(m, generator=g)
    ref = (x.double() @ w.double().T + b.double())
    xd, wd, bd = x.to(cuda), w.to(cuda), b.to(cuda)
    y = torch.full((n, m + 4), 7.0, device=cuda)               # strided output, guard columns
    rc = _lib.lib().sss_linear(xd.data_ptr(), k, wd.data_ptr(), k, bd.data_ptr(), y.data_ptr(), m + 4, n, m, k, _st(cuda))
    _lib.check(rc, "linear")
    got = y.cpu()
    assert (got[:, m:] == 7.0).all()
    scale = float(ref.abs().max())
    assert (got[:, :m].double() - ref).abs().max() <= 2e-6 * max(scale, 1.0) * np.sqrt(k / 32)


def test_linear_is_a_k_ordered_fma_chain(cuda):
    """gfx950 f32 MFMA == sequential fmaf over k (MI355X guide): bit-exact against that chain."""
    g = torch.Generator().manual_seed(3)
    x, w = torch.randn((40, 64), generator=g), torch.randn((33, 64), generator=g)
    xd, wd = x.to(cuda), w.to(cuda)
    y = torch.empty((40, 33), device=cuda)
    _lib.check(_lib.lib().sss_linear(xd.data_ptr(), 64, wd.data_ptr(), 64, 0, y.data_ptr(), 33, 40, 33, 64, _st(cuda)), "linear")
    xn, wn = x.numpy(), w.numpy()
    acc = np.zeros((40, 33), np.float32)
    # kernel k order inside each 8-wide group: lanes<32 take k = 8u+{0..3}, lanes>=32 k = 8u+4+{0..3},
    # one MFMA step consumes (k, k+4): chain order 0,4,1,5,2,6,3,7
    for u in range(8):
        for i in range(4):
            for hh in range(2):
                kk = 8 * u + 4 * hh + i
                acc = _fma(xn[:, kk:kk + 1], wn[:, kk][None, :], acc)
    assert np.array_equal(y.cpu().numpy(), acc)


def _fma(a, b, c):
    # float32 fma via float64: a*b is exact in float64, one rounding of (a*b + c) to float32 --
    # equal to fmaf except for double-rounding cases of probability ~2^-29; the test data is fixed.
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def test_gat_aggregate_matches_oracle(cuda):
    g = torch.Generator().manual_seed(5)
    ns, nd, h = 37, 29, 64
    xs, xd = torch.randn((ns, 32), generator=g), torch.randn((nd, 32), generator=g)
    ls, ld = torch.randn((h, 32), generator=g) * 0.3, torch.randn((h, 32), generator=g) * 0.3
    a_s, a_d, b = torch.randn(h, generator=g), torch.randn(h, generator=g), torch.randn(h, generator=g)
    ei = torch.stack([torch.randint(0, ns, (90,), generator=g), torch.randint(0, nd, (90,), generator=g)])
    ei[:, :5] = torch.tensor([[0, 1, 2, 3, 4], [0, 1, 2, 3, 4]])      # src == dst edges -> dropped by the rewrite
    for loops in (True, False):
        ref = gnn_ref.gat_conv(xs.double(), xd.double(), ei, ls.double(), ld.double(), a_s.double(), a_d.double(),
                               b.double(), self_loops=loops)
        xs_l = (xs @ ls.T).to(cuda).contiguous()
        al_s = ((xs @ ls.T) * a_s).sum(-1).to(cuda).contiguous()
        al_d = ((xd @ ld.T) * a_d).sum(-1).to(cuda).contiguous()
        # the rewrite either materialised in the CSR, or applied on the fly by the kernel
        for csr_loops, n_loop in ((loops, 0), (False, min(ns, nd) if loops else 0)):
            rowptr, col, _ = build_csr(ei.to(cuda), nd, ns, csr_loops)
            out = torch.empty((nd, h), device=cuda)
            rc = _lib.lib().sss_gat_aggregate(xs_l.data_ptr(), h, al_s.data_ptr(), 1, al_d.data_ptr(), 1, rowptr.data_ptr(),
                                              col.data_ptr(), nd, h, b.to(cuda).data_ptr(), 0, n_loop, out.data_ptr(), h,
                                              _st(cuda))
            _lib.check(rc, "gat")
            assert (out.cpu().double() - ref).abs().max() < 2e-5


def test_csr_sum_and_gru_match_torch(cuda):
    g = torch.Generator().manual_seed(6)
    n, h = 50, 64
    x = torch.randn((n, h), generator=g)
    W = torch.randn((h, h), generator=g) * 0.2
    wih, whh = torch.randn((3 * h, h), generator=g) * 0.2, torch.randn((3 * h, h), generator=g) * 0.2
    bih, bhh = torch.randn(3 * h, generator=g), torch.randn(3 * h, generator=g)
    ei = torch.stack([torch.randint(0, n, (120,), generator=g), torch.randint(0, n, (120,), generator=g)])
    ew = torch.rand(120, generator=g) + 0.5
    add = torch.randn((n, h), generator=g)
    for use_w in (False, True):
        ref = gnn_ref.gated_graph_conv(x.double(), ei, W.double(), wih.double(), whh.double(), bih.double(),
                                       bhh.double(), ew.double() if use_w else None)
        ref = torch.relu(ref + add.double())
        m = (x @ W).to(cuda).contiguous()
        rowptr, col, wv = build_csr(ei.to(cuda), n, n, False, ew.to(cuda) if use_w else None)
        magg = torch.empty((n, h), device=cuda)
        rc = _lib.lib().sss_csr_weighted_sum(m.data_ptr(), h, rowptr.data_ptr(), col.data_ptr(),
                                             wv.data_ptr() if use_w else 0, n, h, magg.data_ptr(), h, _st(cuda))
        _lib.check(rc, "csr")
        gi = (magg.cpu() @ wih.T + bih).to(cuda).contiguous()
        gh = (x @ whh.T + bhh).to(cuda).contiguous()
        xd, addd = x.to(cuda), add.to(cuda)
        out = torch.empty((n, h), device=cuda)
        rc = _lib.lib().sss_gru_combine(gi.data_ptr(), 3 * h, gh.data_ptr(), 3 * h, xd.data_ptr(), h, h, addd.data_ptr(), h,
                                        n, h, out.data_ptr(), h, _st(cuda))
        _lib.check(rc, "gru")
        assert (out.cpu().double() - ref).abs().max() < 3e-5


def _run_pair(cuda, cfg, seed, n_sessions, loops=True, batch=None, fused=True):
    cfg.self_loop_rule = "pyg_bipartite_global" if loops else "none"
    w = init_weights(cfg, seed)
    b = batch if batch is not None else S.build_batch(S.synthetic_actions(n_sessions, seed, cfg.n_items, cfg.n_query))
    enc = SessionEncoder(cfg, w, cuda, fused=fused)
    assert enc.fused_ok() == fused
    got, nodes = enc(b.to(cuda), get_node=True)
    bt = b.to_torch("cpu")
    ref, rn = gnn_ref.encoder_forward(bt, w, cfg.n_layers, self_loops=loops, get_node=True)
    ref64 = gnn_ref.encoder_forward(bt, w, cfg.n_layers, self_loops=loops, dtype=torch.float64)
    return got.cpu(), nodes, ref, rn, ref64


@pytest.mark.parametrize("fused", [True, False])      # the 8-launch fused kernels and the per-op kernels
@pytest.mark.parametrize("d,layers,n,loops", [(64, 2, 100, True), (64, 2, 100, False), (128, 2, 300, True),
                                              (128, 3, 64, True), (32, 1, 5, True), (128, 2, 1024, False)])
def test_encoder_matches_oracle(cuda, d, layers, n, loops, fused):
    cfg = EncoderConfig(d_in=d, h=d, n_layers=layers, d_out=d if d > 32 else 64, n_items=5000, n_query=257)
    got, nodes, ref, rn, ref64 = _run_pair(cuda, cfg, 20260001 + d + n, n, loops, fused=fused)
    assert got.shape == ref.shape == (n, cfg.d_out)
    for t in ("query", "product"):
        assert (nodes[t].cpu() - rn[t]).abs().max() < 5e-5          # node outputs, O(1) magnitudes
    assert (got - ref).abs().max() < TOL * max(1.0, float(ref.abs().max()))
    # not further from the float64 truth than the float32 oracle itself is (x4 slack)
    e_got = (got.double() - ref64).abs().max()
    e_ref = (ref.double() - ref64).abs().max()
    assert e_got <= 4 * e_ref + 1e-6
    # what reaches the index: L2-normalised rows within 1e-5
    gn, rn_ = sr.normalize(got.numpy()), sr.normalize(ref.numpy())
    assert np.abs(gn - rn_).max() < TOL


def test_grouped_linear_is_batch_size_invariant(cuda):
    """sss_linear_grouped on a corpus-build batch (150 k rows, gather mode, two problems) equals, BIT FOR BIT, the same
    rows transformed in small batches: every output element is one k-ordered f32 fma chain whatever the grid."""
    from sessionsimilaritysearch_amd.variants import _prob
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    n_p, n_q, K = 150_001, 90_000, 128
    table = torch.randn((5000, K), generator=g).to(cuda)
    ids = torch.randint(0, 5000, (n_p,), generator=g).to(cuda)
    xq = torch.randn((n_q, K), generator=g).to(cuda)
    wp, bp = torch.randn((898, K), generator=g).to(cuda), torch.randn(898, generator=g).to(cuda)
    wq = torch.randn((130, K), generator=g).to(cuda)

    def run(lo_p, hi_p, lo_q, hi_q, act):
        yp = torch.empty((hi_p - lo_p, 898 + 2), device=cuda)
        yq = torch.empty((hi_q - lo_q, 130), device=cuda)
        xc = torch.zeros((hi_p - lo_p, K + 4), device=cuda)
        pp = _lib.LinearProblem(x=0, ldx=0, ids=ids[lo_p:hi_p].data_ptr(), table=table.data_ptr(), xcopy=xc.data_ptr(), ld_xcopy=K + 4,
                                w=wp.data_ptr(), ldw=K, bias=bp.data_ptr(), y=yp.data_ptr(), ldy=900, n=hi_p - lo_p, m=898, act=act)
        pq = _prob(xq[lo_q:hi_q], wq, None, yq, hi_q - lo_q, 130, act)
        _lib.check(L.sss_linear_grouped((_lib.LinearProblem * 2)(pp, pq), 2, K, _st(cuda)), "grouped")
        return yp[:, :898], yq, xc[:, :K]
    for act in (0, 2):
        big_p, big_q, big_x = run(0, n_p, 0, n_q, act)
        for lo in (0, 70_000, n_p - 3000):
            qlo = min(lo, n_q - 3000)
            sp, sq, sx = run(lo, lo + 3000, qlo, qlo + 3000, act)
            assert torch.equal(sp, big_p[lo:lo + 3000]) and torch.equal(sx, big_x[lo:lo + 3000])
            assert torch.equal(sq, big_q[qlo:qlo + 3000])
    ref = table[ids[:2000]].double() @ wp.double().T + bp.double()
    got, _, _ = run(0, n_p, 0, n_q, 0)
    assert (got[:2000].double() - ref).abs().max() <= 2e-6 * float(ref.abs().max()) * np.sqrt(K / 32)


def test_encoder_at_the_reference_model_shapes(cuda):
    """The deployed model's shapes: 768-wide node features (rows of random feature tables here -- the text encoder
    that produces them upstream is out of scope), h = 800, 3 layers, node width 3168, session vector D = 1600,
    batches of 200 sessions (pretrain_filtered_amazon.py:267,281; config.py:15-16,21; test_amazon_filterd.py:488)
    -- against both oracles."""
    cfg = EncoderConfig(d_in=768, h=800, n_layers=3, d_out=1600, n_items=3000, n_query=65)
    w = init_weights(cfg, 20260800)
    b = S.build_batch(S.synthetic_actions(200, 800, cfg.n_items, cfg.n_query))
    enc = SessionEncoder(cfg, w, cuda)
    got, nodes = enc(b.to(cuda), get_node=True)
    bt = b.to_torch("cpu")
    ref, rn = gnn_ref.encoder_forward(bt, w, cfg.n_layers, get_node=True)
    ref64 = gnn_ref.encoder_forward(bt, w, cfg.n_layers, dtype=torch.float64)
    assert got.shape == (200, 1600) and nodes["product"].shape[1] == 3168
    scale = max(1.0, float(ref.abs().max()))
    for t in ("query", "product"):
        assert (nodes[t].cpu() - rn[t]).abs().max() < 5e-5 * max(1.0, float(rn[t].abs().max()))
    assert (got.cpu() - ref).abs().max() < TOL * scale
    e_got, e_ref = (got.cpu().double() - ref64).abs().max(), (ref.double() - ref64).abs().max()
    assert e_got <= 4 * e_ref + 1e-6 * scale
    assert np.abs(sr.normalize(got.cpu().numpy()) - sr.normalize(ref.numpy())).max() < TOL


def test_encoder_wider_hidden_than_input_and_edge_cases(cuda):
    cfg = EncoderConfig(d_in=32, h=64, n_layers=2, d_out=96, n_items=300, n_query=33)
    # sessions with only searches (single "unknown item" node), single-click sessions, repeats
    acts = S.ActionTable(np.array([0, 2, 3, 8, 10]),
                         np.array([1, 1, 0, 0, 1, 0, 0, 0, 1, 1], bool),
                         np.array([0, 0, 7, 7, 0, 9, 7, 7, 0, 0]),
                         np.array([3, 4, 0, 0, 2, 0, 0, 0, 5, 6]))
    b = S.build_batch(acts)
    for fused in (True, False):
        got, nodes, ref, rn, _ = _run_pair(cuda, cfg, 7, 4, True, batch=b, fused=fused)
        assert (got - ref).abs().max() < TOL * max(1.0, float(ref.abs().max()))
        for t in ("query", "product"):
            assert (nodes[t].cpu() - rn[t]).abs().max() < 5e-5


def test_prepared_batch_reuse_edge_weights_and_fused_normalize(cuda):
    """The benchmark's calling pattern: one PreparedBatch, many forwards (cached workspace and
    argument blocks), outputs must not alias; edge weights (HeteroGGNN's optional edge_weight_dict,
    model/gnn.py:68-69); l2_normalize=True == normalize(forward())."""
    from sessionsimilaritysearch_amd.index import normalize
    cfg = EncoderConfig(d_in=64, h=64, n_layers=2, d_out=64, n_items=900, n_query=65, self_loop_rule="none")
    w = init_weights(cfg, 21)
    b = S.build_batch(S.synthetic_actions(50, 21, 900, 65))
    for use_w in (False, True):
        enc = SessionEncoder(cfg, w, cuda, use_edge_weight=use_w)
        pb = enc.prepare(b.to(cuda))
        o1 = enc(pb)
        o2 = enc(pb)
        assert o1.data_ptr() != o2.data_ptr() and torch.equal(o1, o2)
        on = enc(pb, l2_normalize=True)
        assert torch.equal(o1, o2)                               # still intact after another forward
        np.testing.assert_allclose(on.cpu().numpy(), sr.normalize(o1.cpu().numpy()), rtol=2e-6, atol=1e-8)
        bt = b.to_torch("cpu")
        nq_, np_ = gnn_ref.hetero_ggnn(gnn_ref.embedding_lookup(w["query_table"], bt["query"].x),
                                       gnn_ref.embedding_lookup(w["item_table"], bt["product"].x), bt.edge_index_dict, w, 2,
                                       False, bt.edge_weight_dict if use_w else None)
        ref = gnn_ref.pos_att_pool(nq_, np_, bt["query"].pos_emb_id, bt["query"].batch, bt["product"].cnt,
                                   bt["product"].pos_emb_id, bt["product"].batch, bt.num_graphs, w)
        assert (o1.cpu() - ref).abs().max() < TOL * max(1.0, float(ref.abs().max()))
        unf = SessionEncoder(cfg, w, cuda, use_edge_weight=use_w, fused=False)(b.to(cuda))
        assert (unf.cpu() - ref).abs().max() < TOL * max(1.0, float(ref.abs().max()))


def test_pooling_kernels_match_oracle_on_their_own(cuda):
    """PositionalAttentionPooling (model/gnn.py:193-217) through sss_pool_expand_mean +
    sss_linear_grouped + sss_pool_attention, fed with random node features (no GNN in front)."""
    cfg = EncoderConfig(d_in=32, h=32, n_layers=1, d_out=96, n_items=300, n_query=33)
    w = init_weights(cfg, 9)
    b = S.build_batch(S.synthetic_actions(37, 9, 300, 33))
    enc = SessionEncoder(cfg, w, cuda)
    pb = enc.prepare(b.to(cuda))
    g = torch.Generator().manual_seed(9)
    W, D, P = cfg.node_width, cfg.d_out, cfg.max_seq_len
    nq_, np_ = torch.randn((pb.Nq, W), generator=g), torch.randn((pb.Np, W), generator=g)
    bt = b.to_torch("cpu")
    ref = gnn_ref.pos_att_pool(nq_, np_, bt["query"].pos_emb_id, bt["query"].batch, bt["product"].cnt,
                               bt["product"].pos_emb_id, bt["product"].batch, bt.num_graphs, w)
    L = _lib.lib()
    NQ, NP = nq_.to(cuda), np_.to(cuda)
    Dl = D - P
    lin_q, lin_p = torch.empty((pb.Nq, Dl), device=cuda), torch.empty((pb.Np, Dl), device=cuda)
    pw = enc.pool
    P_ = _lib.LinearProblem
    mk = lambda x, wt, bias, y, n, m: P_(x=x.data_ptr(), ldx=x.stride(0), ids=0, table=0, xcopy=0, ld_xcopy=0, w=wt.data_ptr(),
                                          ldw=wt.stride(0), bias=0 if bias is None else bias.data_ptr(), y=y.data_ptr(),
                                          ldy=y.stride(0), n=n, m=m, act=0)
    arr = (P_ * 2)(mk(NP, pw["wp"], pw["bp"], lin_p, pb.Np, Dl), mk(NQ, pw["wq"], pw["bq"], lin_q, pb.Nq, Dl))
    _lib.check(L.sss_linear_grouped(arr, 2, W, _st(cuda)), "lin")
    n_exp = pb.n_clicks + pb.Nq
    node, coarse = torch.empty((n_exp, D), device=cuda), torch.empty((pb.B, D), device=cuda)
    _lib.check(L.sss_pool_expand_mean(lin_p.data_ptr(), lin_q.data_ptr(), Dl, pb.src_row.data_ptr(), pb.pos_id.data_ptr(),
                                      pb.pptr.data_ptr(), pb.qptr.data_ptr(), pb.n_clicks, pb.B, Dl, P, pw["pos"].data_ptr(),
                                      node.data_ptr(), D, coarse.data_ptr(), D, _st(cuda)), "expand")
    A, Bc = torch.empty((n_exp, D), device=cuda), torch.empty((pb.B, D), device=cuda)
    arr2 = (P_ * 2)(mk(node, pw["wn"], pw["bn"], A, n_exp, D), mk(coarse, pw["wc"], None, Bc, pb.B, D))
    _lib.check(L.sss_linear_grouped(arr2, 2, D, _st(cuda)), "lin2")
    out = torch.empty((pb.B, D), device=cuda)
    _lib.check(L.sss_pool_attention(node.data_ptr(), D, A.data_ptr(), D, Bc.data_ptr(), D, pw["watt"].data_ptr(),
                                    pb.pptr.data_ptr(), pb.qptr.data_ptr(), pb.n_clicks, pb.B, D, 0, 1e-6, 0, out.data_ptr(), D,
                                    _st(cuda)), "att")
    assert (out.cpu() - ref).abs().max() < TOL * max(1.0, float(ref.abs().max()))


def test_encoder_get_node_get_token_and_masks(cuda):
    cfg = EncoderConfig(d_in=64, h=64, n_layers=2, d_out=64, n_items=500, n_query=65)
    w = init_weights(cfg, 3)
    b = S.build_batch(S.synthetic_actions(20, 3, 500, 65))
    enc = SessionEncoder(cfg, w, cuda).eval()
    bd = b.to(cuda)
    out = enc(bd)
    o2, tok = enc(bd, get_token=True)
    o3, nodes, tok2 = enc(bd, get_node=True, get_token=True)
    assert tok == {} and tok2 == {} and torch.equal(out, o2) and torch.equal(out, o3)
    assert nodes["product"].shape[1] == cfg.node_width                  # d_in + L*h (add_input_feat=True)
    qm = torch.ones(b["query"].x.shape[0]); qm[::3] = 0
    pm = torch.ones(b["product"].x.shape[0]); pm[1::2] = 0
    got = enc(bd, query_node_mask=qm, product_node_mask=pm).cpu()
    ref = gnn_ref.encoder_forward(b.to_torch("cpu"), w, 2, query_node_mask=qm, product_node_mask=pm)
    assert (got - ref).abs().max() < TOL * max(1.0, float(ref.abs().max()))


def test_pooling_is_permutation_invariant_and_batch_independent_without_loops(cuda):
    """Size-independent properties: with self_loop_rule='none' a session's vector does not depend
    on what else is in the batch (so corpus sharding cannot change embeddings)."""
    cfg = EncoderConfig(d_in=64, h=64, n_layers=2, d_out=64, n_items=2000, n_query=129, self_loop_rule="none")
    w = init_weights(cfg, 11)
    acts = S.synthetic_actions(60, 11, 2000, 129)
    enc = SessionEncoder(cfg, w, cuda)
    full = enc(S.build_batch(acts).to(cuda)).cpu()
    part = enc(S.build_batch(acts.slice(20, 45)).to(cuda)).cpu()
    assert (full[20:45] - part).abs().max() < 2e-6


def test_end_to_end_retrieval_matches_oracle_pipeline(cuda):
    """Config C1 shape: 1000 sessions, d=64, 2 layers, all-vs-all cosine top-10."""
    from sessionsimilaritysearch_amd.index import build_index, normalize
    cfg = EncoderConfig(d_in=64, h=64, n_layers=2, d_out=64, n_items=20000, n_query=513)
    w = init_weights(cfg, 20260000)
    acts = S.synthetic_actions(1000, 20260000, 20000, 513)
    enc = SessionEncoder(cfg, w, cuda)
    embs = [enc(S.build_batch(acts.slice(lo, lo + 200)).to(cuda)) for lo in range(0, 1000, 200)]   # batches of 200 (test_amazon_filterd.py:488)
    emb = torch.cat(embs).cpu().numpy()
    refs = [gnn_ref.encoder_forward(S.build_batch(acts.slice(lo, lo + 200)).to_torch("cpu"), w, 2).numpy()
            for lo in range(0, 1000, 200)]
    ref = np.concatenate(refs)
    assert np.abs(emb - ref).max() < TOL * max(1.0, np.abs(ref).max())
    idx = build_index(emb, "cos", cuda)
    D, I = idx.search(normalize(emb), 10)
    # identical inputs -> bit-exact ids/scores: oracle search on the GPU-normalised rows
    xb = idx._xb.cpu().numpy()
    Dr, Ir = sr.search_exact(normalize(emb), xb, 10)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
    # full oracle pipeline (its own embeddings + numpy normalise): recall@10 of BASELINE.json
    Dro, Iro = sr.build_index(ref, "cos").search(sr.normalize(ref), 10)
    assert sr.recall_at_k(I, Iro, 10) >= 0.999
    assert np.abs(D - Dro).max() < 1e-5


# ---------------------------------------------------------------------------------- golden fixtures
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_gather_rows_matches_reference_node_asin_embedding(cuda):
    """tests/golden/node_asin_embedding.npz was produced by the reference's own NodeAsinEmbedding
    (model/NodeEmbedding.py:128-138, script tests/golden/make_golden.py): the HIP gather must
    reproduce it bit for bit, also into a strided node buffer."""
    z = np.load(os.path.join(GOLDEN, "node_asin_embedding.npz"), allow_pickle=False)
    table, ids = torch.from_numpy(z["table"]).to(cuda), torch.from_numpy(z["ids"]).to(cuda)
    n, d = ids.shape[0], table.shape[1]
    for ld in (d, d + 12):
        out = torch.full((n, ld), -7.0, device=cuda)
        rc = _lib.lib().sss_gather_rows(table.data_ptr(), ids.data_ptr(), n, d, out.data_ptr(), ld, _st(cuda))
        _lib.check(rc, "gather")
        got = out.cpu().numpy()
        assert np.array_equal(got[:, :d], z["out"]) and (got[:, d:] == -7.0).all()


@pytest.mark.parametrize("loops,tag", [(True, "loops"), (False, "noloops")])
def test_encoder_matches_independent_float64_fixture(cuda, loops, tag):
    """HIP encoder vs tests/golden/encoder_tiny.npz (independent numpy-float64 oracle)."""
    z = np.load(os.path.join(GOLDEN, "encoder_tiny.npz"), allow_pickle=False)
    acts = S.ActionTable(z["sess_ptr"], z["is_search"], z["item_id"], z["query_tok"])
    w = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    d_in, h, L, d_out, n_items, n_query = (int(v) for v in z["cfg"])
    cfg = EncoderConfig(d_in=d_in, h=h, n_layers=L, d_out=d_out, n_items=n_items, n_query=n_query,
                        self_loop_rule="pyg_bipartite_global" if loops else "none")
    enc = SessionEncoder(cfg, w, cuda)
    out, nodes = enc(S.build_batch(acts).to(cuda), get_node=True)
    assert np.abs(out.cpu().numpy() - z["out_" + tag]).max() < TOL
    assert np.abs(nodes["query"].cpu().numpy() - z["node_q_" + tag]).max() < TOL
    assert np.abs(nodes["product"].cpu().numpy() - z["node_p_" + tag]).max() < TOL


# ---------------------------------------------------------------------------- native graph builder
@pytest.mark.parametrize("seed,n,vocab", [(0, 1, 50), (1, 7, 50), (2, 300, 50), (3, 2500, 391572), (4, 33000, 391572)])
@pytest.mark.parametrize("loops", [False, True])
def test_native_graph_builder_is_bit_exact(cuda, seed, n, vocab, loops):
    """csrc/graphbuild.hip (action table -> CSR batch on device) against the host builder +
    torch CSR conversion it replaces -- itself checked against oracle/graph_ref.py on CPU --
    array by array, and through the encoder."""
    cfg = EncoderConfig(d_in=32, h=32, n_layers=1, d_out=64, n_items=vocab, n_query=33,
                        self_loop_rule="pyg_bipartite_global" if loops else "none")
    enc = SessionEncoder(cfg, init_weights(cfg, 5, tables=vocab < 1000), cuda, use_edge_weight=True)
    acts = S.synthetic_actions(n, seed, vocab, 33)
    ref = enc.prepare(S.build_batch(acts).to(cuda))
    got = enc.prepare_actions(acts)
    assert (got.Nq, got.Np, got.B, got.n_clicks) == (ref.Nq, ref.Np, ref.B, ref.n_clicks)
    eq = lambda a, b: torch.equal(a.to(torch.int64), b.to(torch.int64))
    assert eq(got.q_ids, ref.q_ids) and eq(got.p_ids, ref.p_ids)
    assert eq(got.q_batch, ref.q_batch) and eq(got.p_batch, ref.p_batch)
    for name in ("csr_qp", "csr_pq", "csr_pp"):
        assert eq(getattr(got, name)[0], getattr(ref, name)[0]), name + ".rowptr"
        assert eq(getattr(got, name)[1], getattr(ref, name)[1]), name + ".col"
    assert torch.equal(got.csr_pp[2], ref.csr_pp[2])
    assert eq(got.src_row, ref.src_row) and eq(got.pos_id, ref.pos_id)
    assert eq(got.pptr, ref.pptr) and eq(got.qptr, ref.qptr) and got.n_self_loop == ref.n_self_loop
    if vocab < 1000:
        assert torch.equal(enc(got), enc(ref))
    if n <= 2500:
        # ... and directly against the ORACLE (oracle/graph_ref.py: the per-session restatement of
        # sequence_to_graph + Batch.from_data_list, util_amazon_filtered.py:128-142,180-218), converted
        # to CSR-by-target here with a plain numpy stable sort -- no product code in between.
        from oracle import graph_ref
        o = graph_ref.collate([graph_ref.session_to_graph(s_) for s_ in graph_ref.actions_to_sessions(acts)])
        Nq, Np = len(o["q_x"]), len(o["p_x"])
        assert (got.Nq, got.Np, got.B) == (Nq, Np, n)

        def csr(src, dst, n_dst, w=None):
            order = np.argsort(dst, kind="stable")
            rowptr = np.zeros(n_dst + 1, np.int64)
            np.cumsum(np.bincount(dst, minlength=n_dst), out=rowptr[1:])
            return rowptr, src[order], None if w is None else w[order]
        npy = lambda t: t.cpu().numpy().astype(np.int64)
        for name, (rp, col, w) in (("csr_qp", csr(o["qp0"], o["qp1"], Np)), ("csr_pq", csr(o["qp1"], o["qp0"], Nq)),
                                   ("csr_pp", csr(o["pp0"], o["pp1"], Np, o["pp_w"]))):
            assert np.array_equal(npy(getattr(got, name)[0]), rp), name + ".rowptr vs oracle"
            assert np.array_equal(npy(getattr(got, name)[1]), col), name + ".col vs oracle"
            if w is not None:
                assert np.array_equal(got.csr_pp[2].cpu().numpy(), w), name + ".weight vs oracle"
        assert np.array_equal(npy(got.q_ids), o["q_x"]) and np.array_equal(npy(got.p_ids), o["p_x"])
        assert np.array_equal(npy(got.q_batch), o["q_batch"]) and np.array_equal(npy(got.p_batch), o["p_batch"])
        src_row = np.r_[np.repeat(np.arange(Np), o["p_cnt"]), np.arange(Nq)]
        assert np.array_equal(npy(got.src_row), src_row)
        assert np.array_equal(npy(got.pos_id), np.r_[o["p_pos"], o["q_pos"]])
        clicks_per_graph = np.bincount(o["p_batch"], weights=o["p_cnt"], minlength=n).astype(np.int64)
        assert np.array_equal(npy(got.pptr), np.r_[0, np.cumsum(clicks_per_graph)])
        assert np.array_equal(npy(got.qptr), np.r_[0, np.cumsum(np.bincount(o["q_batch"], minlength=n))])


def test_native_graph_builder_edge_cases(cuda):
    """Search-only sessions (the "unknown item" node), repeated items, self transitions, a session
    of 64 actions, and the > 64 actions error."""
    cfg = EncoderConfig(d_in=32, h=32, n_layers=1, d_out=96, n_items=300, n_query=33, max_seq_len=65)
    enc = SessionEncoder(cfg, init_weights(cfg, 7), cuda)
    rng = np.random.default_rng(0)
    long_srch = rng.random(64) < 0.3
    ptr = np.array([0, 2, 3, 8, 10, 74])
    srch = np.r_[np.array([1, 1, 0, 0, 1, 0, 0, 0, 1, 1], bool), long_srch]
    item = np.r_[np.array([0, 0, 7, 7, 0, 9, 7, 7, 0, 0]), np.where(long_srch, 0, rng.integers(1, 6, 64))]
    tok = np.r_[np.array([3, 4, 0, 0, 2, 0, 0, 0, 5, 6]), np.where(long_srch, rng.integers(1, 33, 64), 0)]
    acts = S.ActionTable(ptr, srch, item, tok)
    ref = enc.prepare(S.build_batch(acts).to(cuda))
    got = enc.prepare_actions(acts)
    eq = lambda a, b: torch.equal(a.to(torch.int64), b.to(torch.int64))
    for name in ("csr_qp", "csr_pq", "csr_pp"):
        assert eq(getattr(got, name)[0], getattr(ref, name)[0]) and eq(getattr(got, name)[1], getattr(ref, name)[1]), name
    assert eq(got.src_row, ref.src_row) and eq(got.pos_id, ref.pos_id) and eq(got.p_ids, ref.p_ids) and eq(got.q_ids, ref.q_ids)
    assert torch.equal(enc(got), enc(ref))
    too_long = S.ActionTable(np.array([0, 65]), np.zeros(65, bool), np.arange(1, 66), np.zeros(65, np.int64))
    with pytest.raises(_lib.SssError):
        enc.prepare_actions(too_long)


# ---------------------------------------------------------------------------- cache ownership, id range checks
def test_two_encoders_share_one_prepared_batch(cuda):
    """The fused forward caches its workspace / argument blocks with the PreparedBatch; the cache is keyed
    by encoder, so a second encoder (other weights, other widths) over the SAME prepared batch uses its own."""
    acts = S.synthetic_actions(300, 11, 500, 33)
    cfg_a = EncoderConfig(d_in=32, h=32, n_layers=2, d_out=64, n_items=500, n_query=33)
    cfg_b = EncoderConfig(d_in=32, h=64, n_layers=1, d_out=96, n_items=500, n_query=33)
    wa, wb = init_weights(cfg_a, 1), init_weights(cfg_b, 2)
    wb["item_table"], wb["query_table"] = wa["item_table"], wa["query_table"]
    enc_a, enc_b = SessionEncoder(cfg_a, wa, cuda), SessionEncoder(cfg_b, wb, cuda)
    pb = enc_a.prepare_actions(acts)
    out_a1 = enc_a(pb).clone()
    out_b = enc_b(pb).clone()           # same pb, different encoder
    out_a2 = enc_a(pb)
    batch = S.build_batch(acts).to_torch("cpu")
    ref_a = gnn_ref.encoder_forward(batch, wa, cfg_a.n_layers).numpy()
    ref_b = gnn_ref.encoder_forward(batch, wb, cfg_b.n_layers).numpy()
    assert torch.equal(out_a1, out_a2)
    assert np.abs(out_a1.cpu().numpy() - ref_a).max() < TOL * max(1.0, np.abs(ref_a).max())
    assert np.abs(out_b.cpu().numpy() - ref_b).max() < TOL * max(1.0, np.abs(ref_b).max())
    del enc_a                           # freeing the first encoder must not matter to the second
    assert torch.equal(enc_b(pb), out_b)


def test_feature_ids_are_range_checked(cuda):
    """An item / query id outside its table raises what nn.Embedding raises upstream (IndexError), before any launch."""
    cfg = EncoderConfig(d_in=32, h=32, n_layers=1, d_out=64, n_items=50, n_query=9)
    enc = SessionEncoder(cfg, init_weights(cfg, 3), cuda)
    ok = S.ActionTable(np.array([0, 3]), np.array([True, False, False]), np.array([0, 49, 7]), np.array([8, 0, 0]))
    enc(enc.prepare_actions(ok))
    for item, tok in ((50, 8), (49, 9), (-1, 8)):
        bad = S.ActionTable(np.array([0, 3]), np.array([True, False, False]), np.array([0, item, 7]), np.array([tok, 0, 0]))
        with pytest.raises(IndexError):
            enc.prepare_actions(bad)
        with pytest.raises(IndexError):
            enc.prepare(S.build_batch(bad).to(cuda))


def test_cooperative_query_embedding_slices_equal_the_full_batch(cuda):
    """Multi-GPU query embedding (distributed.query_slice / gather_query_embeddings): every rank embeds nq / world of
    the query sessions and an all-gather concatenates them.  Sessions are independent graphs (self_loop_rule
    "none"), so the concatenation must equal -- bit for bit -- the embedding of the whole batch on one rank.  The
    ranks are played one after the other on this GPU; the collective itself is covered by tests/test_distributed_cpu.py."""
    from sessionsimilaritysearch_amd.distributed import gather_query_embeddings, query_slice
    cfg = EncoderConfig(d_in=128, h=128, n_layers=2, d_out=128, n_items=5000, n_query=257, self_loop_rule="none")
    enc = SessionEncoder(cfg, init_weights(cfg, 31), cuda)
    nq = 1024
    acts = S.synthetic_actions(nq, 20269999, cfg.n_items, cfg.n_query)
    full = enc(enc.prepare_actions(acts), l2_normalize=True)
    for world in (2, 4, 8):
        parts = []
        for rank in range(world):
            lo, hi = query_slice(nq, world, rank)
            assert hi - lo == nq // world
            parts.append(enc(enc.prepare_actions(acts.slice(lo, hi)), l2_normalize=True))
        assert torch.equal(torch.cat(parts), full)
    assert query_slice(1000, 3, 1) == (0, 1000)            # world does not divide nq: every rank embeds everything
    assert gather_query_embeddings(full, nq) is full       # single process: nothing to gather


@pytest.mark.parametrize("shape", ["fused", "wide"])
def test_use_id_embedding_concatenates_the_id_row_in_front_of_the_text_features(cuda, shape):
    """use_id_embedding=True (model/model.py:264,288-289): embedding['product'] = concat(id_embedding(x), text features)
    -- product rows d_id + d_in wide, query rows d_in wide, every lazy (-1, -1) weight shaped accordingly.  Against the
    oracle run on the concatenated inputs: table mode (item_text_table stand-in), `.feat` mode (text features brought
    by the batch -> sss_gather_concat_rows writes [id row | feat] in one pass), node outputs at the reference's widths,
    and the input masks."""
    if shape == "fused":
        cfg = EncoderConfig(d_in=64, d_id=32, h=128, n_layers=2, d_out=128, n_items=700, n_query=33)
    else:       # the deployed widths: 768-wide text features + a 32-wide id embedding, h = 800, D = 1600 (per-op kernels)
        cfg = EncoderConfig(d_in=768, d_id=32, h=800, n_layers=3, d_out=1600, n_items=300, n_query=17)
    w = init_weights(cfg, 61)
    assert w["item_table"].shape[1] == cfg.d_id and w["gat_pq.0.lin_src"].shape[1] == cfg.d_p
    assert w["gat_qp.0.lin_src"].shape[1] == cfg.d_in and w["pool.query_lin.w"].shape[1] == cfg.node_width_q
    enc = SessionEncoder(cfg, w, cuda)
    assert enc.fused_ok() == (shape == "fused")
    acts = S.synthetic_actions(50 if shape == "fused" else 24, 61, cfg.n_items, cfg.n_query)
    b = S.build_batch(acts)
    ref, rn = gnn_ref.encoder_forward(b.to_torch("cpu"), w, cfg.n_layers, get_node=True, use_id_embedding=True)
    scale = max(1.0, float(ref.abs().max()))
    got, gn = enc(b.to(cuda), get_node=True)
    assert (got.cpu() - ref).abs().max() < TOL * scale
    assert gn["query"].shape[1] == cfg.node_width_q and gn["product"].shape[1] == cfg.node_width
    for t in ("query", "product"):
        assert (gn[t].cpu() - rn[t]).abs().max() < TOL * max(1.0, float(rn[t].abs().max())), t
    got2 = enc(enc.prepare_actions(acts))                                # native graph build, table mode
    assert (got2.cpu() - ref).abs().max() < TOL * scale
    # the batch brings its own text features (what the out-of-scope text encoder would emit)
    g = torch.Generator().manual_seed(62)
    bt = b.to_torch("cpu")
    bt["query"].feat = torch.randn((bt["query"].x.shape[0], cfg.d_in), generator=g)
    bt["product"].feat = torch.randn((bt["product"].x.shape[0], cfg.d_in), generator=g)
    qm = (torch.rand(bt["query"].x.shape[0], generator=g) > 0.3).float()
    pm = (torch.rand(bt["product"].x.shape[0], generator=g) > 0.3).float()
    ref_f = gnn_ref.encoder_forward(bt, w, cfg.n_layers, use_id_embedding=True, query_node_mask=qm, product_node_mask=pm)
    w_nf = {k: v for k, v in w.items() if k != "item_text_table"}        # no stand-in table needed in this mode
    enc_f = SessionEncoder(cfg, w_nf, cuda)
    bd = b.to(cuda)
    bd["query"].feat, bd["product"].feat = bt["query"].feat.to(cuda), bt["product"].feat.to(cuda)
    got_f = enc_f(bd, query_node_mask=qm.to(cuda), product_node_mask=pm.to(cuda))
    assert (got_f.cpu() - ref_f).abs().max() < TOL * max(1.0, float(ref_f.abs().max()))


def test_gather_concat_rows_is_bit_exact(cuda):
    """sss_gather_concat_rows: out[i] = [table[ids[i]] | feat[i] | zeros] -- a copy kernel, compared bit for bit."""
    g = torch.Generator().manual_seed(63)
    table, feat = torch.randn((40, 32), generator=g), torch.randn((77, 72), generator=g)     # feat rows strided (72 > 64)
    ids = torch.randint(0, 40, (77,), generator=g)
    out = torch.full((77, 140), 9.0, device=cuda)
    td, fd, idd = table.to(cuda), feat.to(cuda), ids.to(cuda)
    rc = _lib.lib().sss_gather_concat_rows(td.data_ptr(), idd.data_ptr(), 32, fd.data_ptr(), 72, 64, 8, 77, out.data_ptr(), 140, _st(cuda))
    _lib.check(rc, "sss_gather_concat_rows")
    ref = torch.cat([table[ids], feat[:, :64], torch.zeros(77, 8), torch.full((77, 36), 9.0)], dim=1)
    assert torch.equal(out.cpu(), ref)
    rc = _lib.lib().sss_gather_concat_rows(0, 0, 0, fd.data_ptr(), 72, 64, 32, 77, out.data_ptr(), 140, _st(cuda))       # query rows: [feat | zeros]
    _lib.check(rc, "sss_gather_concat_rows")
    assert torch.equal(out.cpu()[:, :96], torch.cat([feat[:, :64], torch.zeros(77, 32)], dim=1))

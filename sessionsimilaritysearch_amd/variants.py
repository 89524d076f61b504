"""The reference's other conv / pool variants behind the same ``forward()`` shapes, on the same
CSR-by-target + segment kernels:

* ``HeteroSAGE``        <- ``GNN`` = 3 x ``SAGEConv((-1,-1), h)`` made heterogeneous with ``to_hetero``
                           (``model/gnn.py:83-121``; aggr = 'sum' over edge types, ``config.py:18``)
* ``GraphPooling``      <- ``model/gnn.py:123-143`` (mean / add / max + Linear; dropout is identity in eval)
* ``AttentionPooling``  <- ``model/gnn.py:145-161`` (row-wise dot instead of the dense [n_nodes, B] matrix)
* ``SRGNNPooling``      <- ``model/gnn.py:164-181``
* ``MLPHead``           <- ``model/model.py:40-73`` in eval mode (BatchNorm's eval-mode affine + the relu the
                           reference applies to it run as a second epilogue stage of the Linear's GEMM; tanh at
                           the end when ``last_act``; ``F.dropout`` without ``training=`` is a no-op only for
                           p = 0, which is what the scripts use)
* ``BinarizeHead``      <- ``model/model.py:105-138`` in eval mode: ``sign(lin1(tanh(mlp(x))))`` (with ``jump``:
                           ``lin1([tanh(mlp(x)) ; x])``) -- the +-1 codes ``pack_sign_bits`` / ``BinaryFlatIndex``
                           take (fine_tune_ours.py:839-843)

Every arithmetic step runs in ``libsss.so`` (``sss_csr_mean``, ``sss_segment_reduce``,
``sss_attention_dot_pool``, ``sss_pool_attention``, ``sss_linear_grouped``); torch owns memory only.
Weights are flat ``{name: tensor}`` dicts; conv / pool widths are multiples of 32 up to 2048 (the reference runs them at
``gnn_nout = 800``, config.py:15-16: rows wider than 256 floats are walked in column chunks), the MLP / binarize heads
take any width (the reference's 1600 -> 3000 -> 2000 -> 250).
"""
from __future__ import annotations

import torch

from . import _lib
from .sessions import EDGE_PP, EDGE_PQ, EDGE_QP


def _st(dev):
    return _lib.stream_ptr(dev)


def _prob(x, w, bias, y, n, m, act=0, post=None):
    return _lib.LinearProblem(x=x.data_ptr(), ldx=x.stride(0), ids=0, table=0, xcopy=0, ld_xcopy=0, w=w.data_ptr(),
                              ldw=w.stride(0), bias=0 if bias is None else bias.data_ptr(), y=y.data_ptr(), ldy=y.stride(0),
                              n=n, m=m, act=act, post_scale=0 if post is None else post[0].data_ptr(),
                              post_shift=0 if post is None else post[1].data_ptr())


def linear(x, w, bias=None, act=0, out=None, post=None):
    """y = act(x w^T + bias) through ``sss_linear_grouped`` (act: 0 none, 1 relu, 2 tanh, 3 sign,
    4 tanh(tanh(.))); ``post = (scale, shift)``: then ``relu(y * scale + shift)`` per column.  ``x`` may be
    wider than ``w`` has columns only by zero padding on both sides (K = w.shape[1], a multiple of 32)."""
    n, k = x.shape[0], w.shape[1]
    m = w.shape[0]
    if out is None:
        out = torch.empty((n, m), dtype=torch.float32, device=x.device)
    arr = (_lib.LinearProblem * 1)(_prob(x, w, bias, out, n, m, act, post))
    _lib.check(_lib.lib().sss_linear_grouped(arr, 1, k, _st(x.device)), "sss_linear_grouped")
    return out


def _pad32(n):
    return (n + 31) // 32 * 32


def _pad_cols(w, dev):
    """Weight [m, k] -> device [m, pad32(k)] with zero columns (the GEMM's K is a multiple of 32)."""
    w = w.detach().to(torch.float32)
    out = torch.zeros((w.shape[0], _pad32(w.shape[1])), dtype=torch.float32, device=dev)
    out[:, :w.shape[1]] = w
    return out


def _d(t, dev):
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


class HeteroSAGE:
    """Three hetero SAGEConv layers + relu; returns the LAST layer's node features
    ``{'query': [Nq, h], 'product': [Np, h]}`` (``GNN.forward``, model/gnn.py:114-121).
    weights: ``sage.{l}.{qp|pq|pp}.lin_l.w [h, d]`` / ``.lin_l.b [h]`` / ``.lin_r.w [h, d]``."""

    def __init__(self, weights, n_layers=3, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.layers = []
        for l in range(n_layers):
            g = lambda e, n: weights[f"sage.{l}.{e}.{n}"]
            # sum over edge types is linear: one GEMM per destination type over [agg_1 | agg_2 | x_dst]
            wp = torch.cat([g("qp", "lin_l.w"), g("pp", "lin_l.w"), g("qp", "lin_r.w") + g("pp", "lin_r.w")], dim=1)
            bp = g("qp", "lin_l.b") + g("pp", "lin_l.b")
            wq = torch.cat([g("pq", "lin_l.w"), g("pq", "lin_r.w")], dim=1)
            bq = g("pq", "lin_l.b")
            self.layers.append(tuple(_d(t, self.device) for t in (wp, bp, wq, bq)))

    @torch.no_grad()
    def forward(self, x_q, x_p, csr_qp, csr_pq, csr_pp):
        """``csr_*`` = (rowptr, col) int32 by target: qp -> products, pq -> queries, pp -> products."""
        L, dev = _lib.lib(), self.device
        Nq, Np = x_q.shape[0], x_p.shape[0]
        for wp, bp, wq, bq in self.layers:
            d = x_q.shape[1]
            zp = torch.empty((Np, 3 * d), dtype=torch.float32, device=dev)
            zq = torch.empty((Nq, 2 * d), dtype=torch.float32, device=dev)
            zp[:, 2 * d:] = x_p
            zq[:, d:] = x_q
            for (rowptr, col), src, dst, off, n in ((csr_qp, x_q, zp, 0, Np), (csr_pp, x_p, zp, d, Np), (csr_pq, x_p, zq, 0, Nq)):
                view = dst[:, off:off + d]
                _lib.check(L.sss_csr_mean(src.data_ptr(), src.stride(0), rowptr.data_ptr(), col.data_ptr(), n, d,
                                          view.data_ptr(), dst.stride(0), _st(dev)), "sss_csr_mean")
            h = wp.shape[0]
            out_p = torch.empty((Np, h), dtype=torch.float32, device=dev)
            out_q = torch.empty((Nq, h), dtype=torch.float32, device=dev)
            linear(zp, wp, bp, act=1, out=out_p)                    # K differs per destination type: two launches
            linear(zq, wq, bq, act=1, out=out_q)
            x_q, x_p = out_q, out_p
        return {"query": x_q, "product": x_p}


def segment_reduce(x, ptr, mode, weight=None):
    """Per-graph mean / add / max (mode 0 / 1 / 2) of node rows sorted by graph; ``ptr`` int32 [B+1]."""
    B = ptr.shape[0] - 1
    out = torch.empty((B, x.shape[1]), dtype=torch.float32, device=x.device)
    rc = _lib.lib().sss_segment_reduce(x.data_ptr(), x.stride(0), 0 if weight is None else weight.data_ptr(), ptr.data_ptr(),
                                       B, x.shape[1], mode, out.data_ptr(), out.stride(0), _st(x.device))
    _lib.check(rc, "sss_segment_reduce")
    return out


class GraphPooling:
    MODES = {"mean": 0, "add": 1, "max": 2}

    def __init__(self, pooling_key, weights, device):
        self.pooling_key = pooling_key           # like the reference, the key is only looked at in forward()
        self.w, self.b = _d(weights["lin.w"], device), _d(weights["lin.b"], device)

    @torch.no_grad()
    def forward(self, x, ptr):
        if self.pooling_key == "sort":
            # model/gnn.py:135-136 calls `global_sort_pool(x, batch)` WITHOUT its required `k` (PyG 2.0.4:
            # global_sort_pool(x, batch, k) sorts each graph's nodes by their last channel, keeps / zero-pads to k nodes
            # and returns [B, k * d] -- which `self.lin` (num_in wide) could not take either): the reference raises this
            # TypeError the first time the branch runs, so does the drop-in.
            raise TypeError("global_sort_pool() missing 1 required positional argument: 'k'")
        if self.pooling_key not in self.MODES:
            raise Exception("Unrecognized pooling key: " + self.pooling_key)
        return linear(segment_reduce(x, ptr, self.MODES[self.pooling_key]), self.w, self.b)


class AttentionPooling:
    def __init__(self, weights, device):
        self.w, self.b = _d(weights["lin.w"], device), _d(weights["lin.b"], device)

    @torch.no_grad()
    def forward(self, x, ptr):
        B = ptr.shape[0] - 1
        pooled = torch.empty((B, x.shape[1]), dtype=torch.float32, device=x.device)
        rc = _lib.lib().sss_attention_dot_pool(x.data_ptr(), x.stride(0), ptr.data_ptr(), B, x.shape[1], pooled.data_ptr(),
                                               pooled.stride(0), _st(x.device))
        _lib.check(rc, "sss_attention_dot_pool")
        return linear(pooled, self.w, self.b)


class SRGNNPooling:
    """local = sum(x * last_click_mask); att = lin3(sigmoid(lin1(local)[batch] + lin2(x)));
    out = lin4([local ; sum(x * att)])."""

    def __init__(self, weights, device):
        g = lambda n: _d(weights[n], device)
        self.w1, self.b1, self.w2, self.b2 = g("lin1.w"), g("lin1.b"), g("lin2.w"), g("lin2.b")
        self.w3 = g("lin3.w").view(-1).contiguous()
        self.w4, self.b4 = g("lin4.w"), g("lin4.b")

    @torch.no_grad()
    def forward(self, x, ptr, last_click_mask):
        L, dev = _lib.lib(), x.device
        B, d = ptr.shape[0] - 1, x.shape[1]
        rep = torch.empty((B, 2 * d), dtype=torch.float32, device=dev)
        mask = last_click_mask.to(dev, torch.float32).contiguous()
        rc = L.sss_segment_reduce(x.data_ptr(), x.stride(0), mask.data_ptr(), ptr.data_ptr(), B, d, 1, rep.data_ptr(),
                                  rep.stride(0), _st(dev))
        _lib.check(rc, "sss_segment_reduce")
        local = rep[:, :d]
        a = torch.empty((x.shape[0], d), dtype=torch.float32, device=dev)
        b = torch.empty((B, d), dtype=torch.float32, device=dev)
        arr = (_lib.LinearProblem * 2)(_prob(x, self.w2, self.b2, a, x.shape[0], d), _prob(local, self.w1, self.b1, b, B, d))
        _lib.check(L.sss_linear_grouped(arr, 2, d, _st(dev)), "sss_linear_grouped")
        zero = torch.zeros(B + 1, dtype=torch.int32, device=dev)       # no second row range
        glob = rep[:, d:]
        rc = L.sss_pool_attention(x.data_ptr(), x.stride(0), a.data_ptr(), d, b.data_ptr(), d, self.w3.data_ptr(), ptr.data_ptr(),
                                  zero.data_ptr(), 0, B, d, 0, 1e-6, 1, glob.data_ptr(), rep.stride(0), _st(dev))
        _lib.check(rc, "sss_pool_attention")
        return linear(rep, self.w4, self.b4)


class MLPHead:
    """``MLP`` in eval mode.  weights: ``layers.{i}.w/.b`` for the Linear layers (in order) and
    ``bn.{i}.mean/.var/.gamma/.beta`` for the BatchNorm1d after each but the last (eps 1e-5).  Any widths:
    activations are kept in buffers padded to a multiple of 32 columns (zero pad), weights likewise."""

    def __init__(self, weights, n_hidden_layers, device, last_act=True, jump=False):
        self.jump, self.last_act = jump, last_act
        self.hidden = []
        f64 = lambda t: t.detach().to(torch.float64)
        for i in range(n_hidden_layers + 1):
            s = f64(weights[f"bn.{i}.gamma"]) / torch.sqrt(f64(weights[f"bn.{i}.var"]) + 1e-5)
            # the reference applies relu to EVERY module of layers[:-1], Linear and BatchNorm alike
            # (model/model.py:63-65): relu(bn(relu(lin(x)))).  Both stages are epilogue steps of the Linear's GEMM:
            # act = relu, then the BatchNorm eval affine + relu per output column.
            self.hidden.append((_pad_cols(weights[f"layers.{i}.w"], device), _d(weights[f"layers.{i}.b"], device),
                                _d(s.float(), device),
                                _d((f64(weights[f"bn.{i}.beta"]) - f64(weights[f"bn.{i}.mean"]) * s).float(), device)))
        last = n_hidden_layers + 1
        self.n_in = int(weights["layers.0.w"].shape[1])
        self.n_hidden = int(weights["layers.0.w"].shape[0])
        self.wl_raw = weights[f"layers.{last}.w"]
        if jump:            # last Linear reads [inp ; hidden]: pad each part of K separately so both stay 32-aligned
            wl = torch.zeros((self.wl_raw.shape[0], _pad32(self.n_in) + _pad32(self.n_hidden)))
            wl[:, :self.n_in] = self.wl_raw[:, :self.n_in]
            wl[:, _pad32(self.n_in):_pad32(self.n_in) + self.n_hidden] = self.wl_raw[:, self.n_in:]
            self.wl = _d(wl, device)
        else:
            self.wl = _pad_cols(self.wl_raw, device)
        self.bl = _d(weights[f"layers.{last}.b"], device)
        self.n_out = int(self.wl_raw.shape[0])

    def _in_buffer(self, x):
        """x [n, n_in] -> a buffer whose row is [x | 0-pad | hidden | 0-pad] (jump) or [x | 0-pad]."""
        n, dev = x.shape[0], x.device
        kin, kh = _pad32(self.n_in), _pad32(self.n_hidden)
        if not self.jump and kin == self.n_in and x.is_contiguous():
            return x, None
        buf = torch.zeros((n, kin + (kh if self.jump else 0)), dtype=torch.float32, device=dev)
        buf[:, :self.n_in] = x
        return buf[:, :kin], buf

    @torch.no_grad()
    def forward(self, x, act_last=None, out=None):
        n, dev = x.shape[0], x.device
        kin, kh = _pad32(self.n_in), _pad32(self.n_hidden)
        xin, whole = self._in_buffer(x)
        cur = xin
        for li, (w, b, s, t) in enumerate(self.hidden):
            last_hidden = li == len(self.hidden) - 1
            if self.jump and last_hidden:
                dst = whole[:, kin:kin + self.n_hidden]              # lands next to the input: the concat is free
            else:
                full = torch.zeros((n, kh), dtype=torch.float32, device=dev)
                dst = full[:, :self.n_hidden]
            linear(cur, w, b, act=1, out=dst, post=(s, t))           # relu(bn(relu(lin(x))))
            cur = whole[:, kin:kin + kh] if (self.jump and last_hidden) else full
        src = whole if self.jump else cur
        if act_last is None:
            act_last = 2 if self.last_act else 0
        return linear(src, self.wl, self.bl, act=act_last, out=out)

    __call__ = forward


class BinarizeHead:
    """``BinarizeHead`` in eval mode (model/model.py:105-138): ``sign(lin1(h))`` with ``h = x`` when there is
    no mlp, else ``h = tanh(mlp(x))`` (``jump``: ``h = [tanh(mlp(x)) ; x]``).  The reference returns
    ``(sign(out) - tanh(out)).detach() + tanh(out)``, which is exactly ``sign(out)`` in float32 (for |t| < 1 the
    rounding of ``1 - t`` is undone by adding ``t`` back: the sum lies within half an ulp of 1).  weights:
    ``lin1.w [n_output, n_input]``, ``lin1.b``; ``mlp`` is an ``MLPHead`` (its ``last_act`` tanh is applied, then
    the head's own tanh: ``tanh(tanh(.))`` in one epilogue).  ``forward(x, pre_sign=True)`` returns ``lin1(h)``
    before the sign (what the parity test compares)."""

    def __init__(self, weights, mlp, device, jump=False):
        self.mlp, self.jump = mlp, jump
        self.n_in = int(weights["lin1.w"].shape[1])
        w = weights["lin1.w"]
        if mlp is not None and jump:
            m_out = mlp.n_out
            wl = torch.zeros((w.shape[0], _pad32(m_out) + _pad32(self.n_in - m_out)))
            wl[:, :m_out] = w[:, :m_out]
            wl[:, _pad32(m_out):_pad32(m_out) + self.n_in - m_out] = w[:, m_out:]
            self.w1 = _d(wl, device)
        else:
            self.w1 = _pad_cols(w, device)
        self.b1 = _d(weights["lin1.b"], device)

    @torch.no_grad()
    def forward(self, x, pre_sign=False):
        n, dev = x.shape[0], x.device
        act = 0 if pre_sign else 3
        if self.mlp is None:
            k = self.w1.shape[1]
            if x.shape[1] != k or not x.is_contiguous():
                buf = torch.zeros((n, k), dtype=torch.float32, device=dev)
                buf[:, :x.shape[1]] = x
                x = buf
            return linear(x, self.w1, self.b1, act=act)
        m_out = self.mlp.n_out
        km = _pad32(m_out)
        if self.jump:
            buf = torch.zeros((n, self.w1.shape[1]), dtype=torch.float32, device=dev)
            buf[:, km:km + x.shape[1]] = x
            h = buf
        else:
            buf = torch.zeros((n, km), dtype=torch.float32, device=dev)
            h = buf
        self.mlp.forward(x, act_last=4 if self.mlp.last_act else 2, out=buf[:, :m_out])    # tanh(mlp(x))
        return linear(h, self.w1, self.b1, act=act)

    __call__ = forward

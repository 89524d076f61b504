"""ctypes binding of libsss.so (the C ABI declared in include/sss.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, this
module raises.  ``build()`` compiles the library in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsss.so")
CSRC = os.path.join(_HERE, "csrc")

_lib = None

# name -> (restype, argtypes); mirrors include/sss.h one to one
_SIGNATURES = {
    "sss_version": (c_int, []),
    "sss_last_error": (ctypes.c_char_p, []),
    "sss_normalize_rows": (c_int, [c_void_p, c_int64, c_int, c_int64, c_float, c_int, c_void_p]),
    "sss_row_norm_max": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "sss_f32_to_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "sss_ip_topk_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int, c_int]),
    "sss_ip_topk_state_bytes": (c_size_t, [c_int64]),
    "sss_ip_topk": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_float,
                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "sss_split_bf16": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "sss_ip_topk_split": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_float,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t,
                                  c_void_p]),
    "sss_abs_max": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "sss_f16_shift": (c_int, [c_float]),
    "sss_scale_f16": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "sss_ip_topk_f16_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int]),
    "sss_f16_resid_max": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "sss_ip_topk_f16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_float, c_int64, c_int, c_int, c_int64, c_float,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t,
                                c_void_p]),
    "sss_ip_topk_long_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int]),
    "sss_ip_topk_long": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_float, c_int64, c_int, c_int, c_int64,
                                 c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sss_ip_topk_threshold_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int]),
    "sss_ip_topk_threshold": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_int64,
                                      c_int, c_int, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                      c_void_p]),
    "sss_ip_topk_exhaustive_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sss_ip_topk_exhaustive": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                       c_int64, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                       c_void_p]),
    "sss_ip_topk_exhaustive_lb": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                          c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sss_topk_merge": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int, c_void_p, c_void_p,
                               c_void_p]),
    "sss_scan_boot_expired": (c_int, [c_int]),
    "sss_profile_enable": (c_int, [c_int]),
    "sss_profile_read": (c_int, [c_void_p, c_void_p]),
    "sss_gather_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "sss_gather_concat_rows": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p, c_int64,
                                       c_void_p]),
    "sss_linear": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                           c_int, c_int, c_void_p]),
    "sss_gat_aggregate": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                  c_int64, c_int, c_void_p, c_int, c_int64, c_void_p, c_int64, c_void_p]),
    "sss_csr_weighted_sum": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                     c_void_p, c_int64, c_void_p]),
    "sss_gru_combine": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p,
                                c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "sss_pool_expand": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "sss_segment_pool": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p,
                                 c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "sss_segment_ptr": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "sss_pack_sign_bits": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int, c_void_p]),
    "sss_hamming_topk_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sss_hamming_topk_capacity": (c_int, [c_int64, c_int64]),
    "sss_hamming_topk": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_void_p]),
    "sss_hamming_topk_exhaustive_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sss_hamming_topk_exhaustive": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int64,
                                            c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sss_graph_scratch_ints": (c_size_t, [c_int64]),
    "sss_graph_counts": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sss_graph_fill": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "sss_knn_item_vote": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "sss_linear_grouped": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "sss_hetero_layer_update": (c_int, [c_void_p, c_void_p]),
    "sss_pool_expand_mean": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_int64, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "sss_pool_attention": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                   c_int64, c_int64, c_int, c_int, c_float, c_int, c_void_p, c_int64, c_void_p]),
    "sss_pool_attention_tab": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_float,
                                       c_void_p, c_int64, c_void_p]),
    "sss_csr_mean": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "sss_segment_reduce": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "sss_attention_dot_pool": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
}


class LinearProblem(ctypes.Structure):
    """``sss_linear_problem`` of include/sss.h."""
    _fields_ = [("x", c_void_p), ("ldx", c_int64), ("ids", c_void_p), ("table", c_void_p), ("xcopy", c_void_p),
                ("ld_xcopy", c_int64), ("w", c_void_p), ("ldw", c_int64), ("bias", c_void_p), ("y", c_void_p),
                ("ldy", c_int64), ("n", c_int64), ("m", c_int32), ("act", c_int32), ("post_scale", c_void_p), ("post_shift", c_void_p)]


class GraphOut(ctypes.Structure):
    """``sss_graph_out`` of include/sss.h."""
    _fields_ = [(n, c_void_p) for n in ("q_x", "q_batch", "q_pos", "p_x", "p_batch", "p_cnt", "rowptr_qp", "col_qp",
                                        "rowptr_pq", "col_pq", "rowptr_pp", "col_pp", "w_pp", "src_row", "pos_id")]


class LayerArgs(ctypes.Structure):
    """``sss_layer_args`` of include/sss.h."""
    _fields_ = [("yp", c_void_p), ("ld_yp", c_int64), ("yq", c_void_p), ("ld_yq", c_int64), ("h", c_int32), ("d_x", c_int32),
                ("rowptr_qp", c_void_p), ("col_qp", c_void_p), ("rowptr_pp", c_void_p), ("col_pp", c_void_p),
                ("w_pp", c_void_p), ("bias_qp", c_void_p), ("b_ih", c_void_p),
                ("xin_p", c_void_p), ("ld_xin", c_int64), ("out_p", c_void_p), ("ld_out_p", c_int64), ("np", c_int64),
                ("rowptr_pq", c_void_p), ("col_pq", c_void_p), ("bias_pq", c_void_p),
                ("out_q", c_void_p), ("ld_out_q", c_int64), ("nq", c_int64), ("n_self_loop", c_int64),
                ("row_p", c_void_p), ("row_q", c_void_p), ("x0_p", c_void_p), ("ld_x0_p", c_int64),
                ("xq_table", c_void_p), ("ld_xq", c_int64), ("x0_q", c_void_p), ("ld_x0_q", c_int64)]


def exported_symbols():
    return sorted(_SIGNATURES)


def build(verbose: bool = False) -> str:
    """Compile libsss.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libsss.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout)
    return LIB_PATH


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "sessionsimilaritysearch_amd/csrc`). There is no CPU fallback.")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


class SssError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().sss_last_error()
        raise SssError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def stream_ptr(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(t, name, dtype=None):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise SssError(f"{name} must be a CUDA (HIP) tensor")
    if dtype is not None and t.dtype != dtype:
        raise SssError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise SssError(f"{name} must be contiguous")
    return t

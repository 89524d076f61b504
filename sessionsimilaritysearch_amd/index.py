"""faiss-shaped flat index on one MI355X + the reference's ``build_index`` / ``normalize``.

Drop-in surface (same names, argument meaning and error behaviour as the reference):

* ``normalize(vec)``                  <- ``util_amazon_filtered.py:28-31``
* ``build_index(emb, metric)``        <- ``test_amazon_filterd.py:207-223`` ('cos' | 'l2' | 'ip',
                                         anything else raises ``RuntimeError("Unregnozed metric", metric)``)
* ``FlatIndex(d).add(x)`` / ``.search(x, k) -> (D, I)`` / ``.ntotal`` / ``.d``
                                      <- ``faiss.IndexFlatIP`` / ``IndexFlatL2`` as used at
                                         ``test_amazon_filterd.py:212-220,578``

All arithmetic runs in the HIP kernels of ``libsss.so``; torch only owns device memory and the
stream.  There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

FUSED_DIMS = {"f32": (64, 128, 256), "bf16": (128, 256, 512)}   # row bytes 256 / 512 / 1024
F16_SCAN_DIMS = (128, 256, 512)                                  # scaled-f16 image: 2 bytes per element
# scan="auto": the fastest scan whose error bound is still small against the spacing of the scores
# around rank k (the spacing shrinks as k grows): one-pass f16 up to k = 128, bf16 split up to k = 500
# (every k the fused path serves); the f32 MFMA scan is reached by escalation only.  Measured on random unit
# rows, d = 128, 1024 queries (round 3): f16 leaves 0 of 6144 queries unproven at k <= 64 on 10M rows, 8 at k = 128,
# 60 at k = 200 (the k classes are cut there); split leaves 0-1 up to k = 500, at 0.4x the f32 scan's time.  A
# search whose fallback share exceeds AUTO_ESCALATE moves that k class one scan up for the following searches
# (near-duplicate-heavy or unusually dense corpora); an unproven query costs a share of one more scan of the
# corpus for the unproven ones only (the threshold rung), so a few per batch are cheaper than the slower scan.
AUTO_F16_MAX_K = 128
AUTO_SPLIT_MAX_K = 500
AUTO_ESCALATE = 0.005
AUTO_DECAY_SEARCHES = 64        # clean searches at an escalated level before the class steps back down one scan
_LADDER = ("f16", "split", "f32")
FUSED_MAX_K = 500
LONG_MAX_K = 1024                                                # sss_ip_topk_long: what its exhaustive fallback resolves
LONG_MAX_ROW_BYTES = 16384
DTYPE_CODE = {"f32": 0, "bf16": 1}                               # include/sss.h: dtype
_EXHAUSTIVE_WS_BYTES = 1 << 30
SEARCH_CHUNK = 65536             # queries per fused call of search_device (workspace 16 KB per query)
SEARCH_CHUNK_LONG = 16384        # ... on the long-row path (64 KB per query)


def _dev(device=None):
    if not torch.cuda.is_available():
        raise _lib.SssError("no HIP device available: the session-similarity path runs on MI355X only")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
    if device.type != "cuda":
        raise _lib.SssError(f"device must be a HIP device, got {device}")
    return torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())


def _as_device_f32(x, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    if not isinstance(x, torch.Tensor):
        raise TypeError("expected a numpy array or torch tensor")
    return x.to(device=device, dtype=torch.float32).contiguous()


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    """float32 CUDA tensor -> bfloat16 (round to nearest even) through ``sss_f32_to_bf16``."""
    _lib.require_cuda(x, "x", torch.float32)
    if x.numel() % 8:
        raise _lib.SssError("to_bf16: element count must be a multiple of 8")
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    rc = _lib.lib().sss_f32_to_bf16(x.data_ptr(), x.numel(), y.data_ptr(), _lib.stream_ptr(x.device))
    _lib.check(rc, "sss_f32_to_bf16")
    return y


def normalize_(x: torch.Tensor, eps: float = 1e-6, rule: int = 0) -> torch.Tensor:
    """In-place row normalisation of a CUDA float32 [n, d] tensor (rows may be strided)."""
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype != torch.float32 or not x.is_cuda:
        raise _lib.SssError("normalize_: need a CUDA float32 [n, d] tensor with unit inner stride")
    rc = _lib.lib().sss_normalize_rows(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), eps, rule,
                                       _lib.stream_ptr(x.device))
    _lib.check(rc, "sss_normalize_rows")
    return x


def normalize(vec, eps: float = 1e-6, rule: int = 0):
    """Reference ``normalize``: ``v / sqrt(clip(sum(v**2), 1e-6))`` row-wise (1-D input: the
    whole vector).  numpy in -> numpy out; CUDA tensor in -> new CUDA tensor out.
    ``rule=1, eps=1e-4`` gives the fine-tune scripts' ``v / (||v|| + 1e-4)``."""
    is_np = isinstance(vec, np.ndarray)
    dev = _dev() if is_np or not vec.is_cuda else vec.device
    one_d = vec.ndim == 1
    x = _as_device_f32(vec, dev)
    if x.data_ptr() == (vec.data_ptr() if isinstance(vec, torch.Tensor) else 0):
        x = x.clone()
    if one_d:
        x = x.view(1, -1)
    d = x.shape[1]
    if d % 4:                       # kernel moves 16 bytes per lane: pad the row, cut it back
        pad = torch.zeros(x.shape[0], (d + 3) // 4 * 4, device=dev, dtype=torch.float32)
        pad[:, :d] = x
        x = normalize_(pad, eps, rule)[:, :d].contiguous()
    else:
        normalize_(x, eps, rule)
    if one_d:
        x = x.view(-1)
    return x.cpu().numpy() if is_np else x


class FlatIndex:
    """Exact flat index (faiss ``IndexFlatIP`` / ``IndexFlatL2`` semantics) with
    the canonical result contract of DESIGN.md: scores are float64-accumulated dot products
    rounded to float32, ordered by (score desc, id asc); missing results are (-FLT_MAX, -1).

    ``dtype="bf16"`` (BASELINE config C5) stores the corpus -- and rounds every query -- to
    bfloat16 and scores on the bf16 MFMA; the contract is then defined on the ROUNDED vectors
    (float64 dot of the stored bf16 values).

    ``scan`` picks how a float32 index finds its candidates (the results are the same, they are
    re-scored from the float32 rows and proven per query either way; what differs is speed and how
    many near-tied queries are left to the exhaustive fallback):
    ``"f16"`` keeps a float16 image of the corpus scaled by one power of two (half the bytes) and
    scans it with ONE f16 MFMA pass -- score error ~4e-4 |q||c| at d = 128 (d in 128/256/512);
    ``"split"`` keeps each element as a bfloat16 hi/lo pair (same bytes as the f32 row) and scans
    with three bf16 MFMA passes -- error <= ~2^-14 |q||c|;
    ``"f32"`` scans the float32 rows on the f32 MFMA (error ~ d 2^-24) and needs no second image;
    ``"auto"`` (default) takes "f16" for k <= 128 where the shape allows and "split" up to k = 500 (every k
    the fused path serves; the scores around rank k lie closer together as k grows); "f32" is reached by
    escalation only: a k class moves one scan up when a search left more than 0.5 % of its queries (at least
    4) unproven, the step is taken back (and the class pinned) when the slower scan proves no more of them,
    and a class steps back down after 64 CONSECUTIVE clean searches.  Images are built on first use
    (``prepare(k)`` does it ahead of time) and extended as rows are added.

    Queries a scan leaves unproven are resolved in two further stages, both exact: the THRESHOLD RUNG
    (``search_threshold``: one more matrix-core scan for just those queries that keeps every row able to
    reach the k-th score already known, near ties and duplicate rows alike) and, for what exceeds its
    capacity, the exhaustive kernels (``search_exhaustive``)."""

    def __init__(self, d: int, metric: str = "ip", device=None, dtype: str = "f32", scan: str | None = None):
        if metric not in ("ip", "l2"):
            raise ValueError("metric must be 'ip' or 'l2'")
        if dtype not in DTYPE_CODE:
            raise ValueError("dtype must be 'f32' or 'bf16'")
        if dtype == "bf16" and d % 8:
            raise ValueError("bf16 index needs d % 8 == 0")
        if scan is None:
            scan = "auto" if dtype == "f32" else "native"
        if scan not in (("auto", "f16", "split", "f32") if dtype == "f32" else ("native",)):
            raise ValueError("scan must be 'auto', 'f16', 'split' or 'f32' for a float32 index")
        self.scan = scan
        self.last_scan = None           # the scan the last fused search used
        self._auto_level = {}           # scan="auto": k class -> lowest ladder level still allowed
        self._auto_clean = {}           # scan="auto": k class -> consecutive clean searches at the escalated level
        self._auto_rows = 0             # scan="auto": corpus size when a class last escalated
        self.d = int(d)
        self.metric = metric
        self.dtype = dtype
        self._tdtype = torch.float32 if dtype == "f32" else torch.bfloat16
        self.device = _dev(device)
        # derived corpus images, built on first use and extended as rows are added
        self._split = None              # [cap, 2d] bf16 hi|lo image of the rows        ("split" scan)
        self._split_done = 0            # rows of it that are valid
        self._f16 = None                # [cap, d] float16 image of rows * 2^_c_shift  ("f16" scan)
        self._f16_done = 0
        self._c_shift = 0
        self._amax_t = torch.zeros(1, dtype=torch.float32, device=self.device)    # largest |element| in the f16 image
        self._resid_t = torch.zeros(1, dtype=torch.float32, device=self.device)   # largest row residual norm of it
        self._resid = None
        self._xb = torch.empty((0, self.d), dtype=self._tdtype, device=self.device)
        self._store = self._xb          # backing storage of _xb (grown geometrically by add())
        self._cmax_t = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._cmax = None
        self._ws = None
        self._state = None              # per-query state words of sss_ip_topk: zeroed once, kept zero by the kernels
        self.id_offset = 0              # global id of row 0 (row-sharded corpora)
        self.last_fallback_queries = 0  # queries of the last search() re-run exhaustively
        self.last_rescan_queries = 0    # queries of the last search() the fused scan left unproven (threshold rung first)

    @property
    def ntotal(self) -> int:
        return int(self._xb.shape[0])

    def add(self, x):
        """Append rows (copied, ids = insertion order) -- ``IndexFlatIP.add``."""
        x = self._rows(x, "add")
        n_old = self.ntotal
        if n_old + x.shape[0] > self._store.shape[0]:          # amortised growth: no re-copy per add()
            cap = max(n_old + x.shape[0], 2 * self._store.shape[0])
            store = torch.empty((cap, self.d), dtype=self._tdtype, device=self.device)
            store[:n_old] = self._xb
            self._store = store
        self._store[n_old:n_old + x.shape[0]] = x
        self._xb = self._store[:n_old + x.shape[0]]
        self._norm_max(x)
        # streaming adds keep what the searches have learned about this corpus; only once it has doubled since
        # an escalation was earned is that treated as a different corpus (adopt() always resets)
        if self._auto_level and self.ntotal > 2 * max(1, self._auto_rows):
            self._auto_level.clear()
            self._auto_clean.clear()

    def scan_for(self, k: int) -> str:
        """Which candidate scan a fused search for k results uses ("" = none: exhaustive path)."""
        if self.metric != "ip" or k <= 0 or self.ntotal == 0:
            return ""
        fused_shape = self.d in FUSED_DIMS[self.dtype] or (self.dtype == "f32" and self.d in F16_SCAN_DIMS)
        if not fused_shape:
            return self._long_or_none(k)         # rows longer than the register-resident scans take (D = 1600 ...)
        if k > FUSED_MAX_K:
            return ""
        if self.dtype != "f32":
            return "native"
        want = self.scan
        if want == "auto":
            level = max(self._k_class(k), self._auto_level.get(self._k_class(k), 0))
            served = [s for s in _LADDER[level:] if self._scan_served(s)]
            # nothing at or above the wanted level fits this d (e.g. d = 512: only the f16 image does):
            # stay on the fastest scan that does rather than fall off the ladder
            served = served or [s for s in _LADDER if self._scan_served(s)]
            return served[0] if served else ""
        if want == "f16" and not self._scan_served("f16"):
            want = "split"
        return want if self._scan_served(want) else ""

    def _long_or_none(self, k: int) -> str:
        """"long": the K-tiled scan for rows beyond the register-resident kernels (``sss_ip_topk_long``)."""
        row_bytes = self.d * (4 if self.dtype == "f32" else 2)
        return "long" if (self.d % 64 == 0 and row_bytes <= LONG_MAX_ROW_BYTES and k <= LONG_MAX_K) else ""

    def _scan_served(self, scan: str) -> bool:
        """Does a fused kernel exist for this scan at this d?"""
        return self.d in (F16_SCAN_DIMS if scan == "f16" else FUSED_DIMS["f32"])

    def next_scan(self, scan: str) -> str:
        """The next scan up the precision ladder that this d can run ("" = none)."""
        if self.dtype != "f32" or scan not in _LADDER:
            return ""
        return next((s for s in _LADDER[_LADDER.index(scan) + 1:] if self._scan_served(s)), "")

    @staticmethod
    def _k_class(k: int) -> int:
        return 0 if k <= AUTO_F16_MAX_K else 1 if k <= AUTO_SPLIT_MAX_K else 2

    def _note_fallbacks(self, k: int, nq: int, bad: int):
        """scan="auto": move this k class one scan up when too many queries of a search were left
        unproven by it -- only to a scan this d can run -- and back down one scan after
        AUTO_DECAY_SEARCHES consecutive clean searches (one near-duplicate-heavy batch does not demote
        the index for good); an escalation that proves no more queries than the faster scan did is undone."""
        if self.scan != "auto" or self.last_scan not in _LADDER:
            return
        kc = self._k_class(k)
        share = bad / max(nq, 1)
        # The first search after an escalation tells whether it helped: exact ties (duplicate rows at the k-th place)
        # stay unproven under EVERY scan -- the threshold rung resolves them, at a cost that hardly depends on how
        # many there are -- so a slower scan that still leaves more than AUTO_ESCALATE of the batch unproven only
        # costs time.  Such a step is taken back and the class pinned for AUTO_DECAY_SEARCHES searches.
        probe = self._auto_clean.pop(("probe", kc), None)
        if probe is not None and nq >= 32 and bad >= 4 and share > AUTO_ESCALATE:
            self._auto_level[kc] = probe[0]
            self._auto_clean[("pin", kc)] = AUTO_DECAY_SEARCHES
            self._auto_clean[kc] = 0
            return
        pin = self._auto_clean.get(("pin", kc), 0)
        if pin > 0:
            self._auto_clean[("pin", kc)] = pin - 1
            return
        if nq >= 32 and bad >= 4 and bad > AUTO_ESCALATE * nq:
            up = self.next_scan(self.last_scan)
            if up:
                self._auto_clean[("probe", kc)] = (_LADDER.index(self.last_scan), share)
                self._auto_level[kc] = _LADDER.index(up)
                self._auto_rows = self.ntotal
            self._auto_clean[kc] = 0
        elif self._auto_level.get(kc, 0) > kc and nq >= 32:
            self._auto_clean[kc] = self._auto_clean.get(kc, 0) + 1 if bad == 0 else 0      # consecutive: any unproven query restarts the count
            if self._auto_clean[kc] >= AUTO_DECAY_SEARCHES:
                self._auto_level[kc] -= 1
                self._auto_clean[kc] = 0

    def _grow_image(self, img, done, width, tdtype):
        """The image tensor with room for every row of the store, its first `done` rows kept."""
        if img is None or img.shape[0] < self._store.shape[0] or img.shape[0] < self.ntotal:
            new = torch.empty((max(self._store.shape[0], self.ntotal), width), dtype=tdtype, device=self.device)
            if img is not None and done:
                new[:done] = img[:done]
            img = new
        return img

    def _ensure_f16(self):
        """Bring the scaled float16 image up to date.  Its shift is fixed by the largest |element|
        present when it was last rebuilt (which then lies in [2^12, 2^13)); rows that would push an
        element past 2^15 trigger a rebuild of the whole image with a new shift."""
        n, lo = self.ntotal, self._f16_done
        if lo == n and self._f16 is not None:
            return
        L, st = _lib.lib(), _lib.stream_ptr(self.device)
        _lib.check(L.sss_abs_max(self._xb[lo:].data_ptr(), (n - lo) * self.d, self._amax_t.data_ptr(), st), "sss_abs_max")
        amax = float(self._amax_t.item())
        self._f16 = self._grow_image(self._f16, lo, self.d, torch.float16)
        if lo == 0 or not (amax * 2.0 ** self._c_shift < 32768.0):
            self._c_shift = int(L.sss_f16_shift(amax))
            self._resid_t.zero_()
            lo = 0
        _lib.check(L.sss_scale_f16(self._xb[lo:].data_ptr(), (n - lo) * self.d, self._c_shift,
                                   self._f16[lo:].data_ptr(), st), "sss_scale_f16")
        _lib.check(L.sss_f16_resid_max(self._xb[lo:].data_ptr(), self._f16[lo:].data_ptr(), n - lo, self.d,
                                       self._c_shift, self._resid_t.data_ptr(), st), "sss_f16_resid_max")
        self._resid = None
        self._f16_done = n

    def corpus_resid_norm(self) -> float:
        if self._resid is None:
            self._resid = float(self._resid_t.item())
        return self._resid

    def _ensure_split(self):
        """Bring the bf16 hi|lo image up to date."""
        n, lo = self.ntotal, self._split_done
        if lo == n and self._split is not None:
            return
        self._split = self._grow_image(self._split, lo, 2 * self.d, torch.bfloat16)
        rc = _lib.lib().sss_split_bf16(self._xb[lo:].data_ptr(), n - lo, self.d, self._split[lo:].data_ptr(),
                                       _lib.stream_ptr(self.device))
        _lib.check(rc, "sss_split_bf16")
        self._split_done = n

    def prepare(self, k: int = 10):
        """Build whatever a fused search for k results needs (images, norms) now rather than on
        the first search; returns the scan that will be used."""
        mode = self.scan_for(k)
        if mode == "f16" or (mode == "long" and self.dtype == "f32"):
            self._ensure_f16()
            self.corpus_resid_norm()
        elif mode == "split":
            self._ensure_split()
        self.corpus_max_norm()
        return mode

    def _rows(self, x, what):
        """Input rows as a contiguous device tensor of the index's element type."""
        if isinstance(x, torch.Tensor) and x.dtype == torch.bfloat16 and self.dtype == "bf16":
            x = x.to(self.device).contiguous()
        else:
            x = _as_device_f32(x, self.device)
            if self.dtype == "bf16":
                x = to_bf16(x)
        if x.dim() != 2 or x.shape[1] != self.d:
            raise ValueError(f"{what}: expected [n, {self.d}], got {tuple(x.shape)}")
        return x

    def _norm_max(self, x):
        if x.shape[0] and self.d % (4 if self.dtype == "f32" else 8) == 0:
            rc = _lib.lib().sss_row_norm_max(x.data_ptr(), x.shape[0], self.d, DTYPE_CODE[self.dtype],
                                             self._cmax_t.data_ptr(), _lib.stream_ptr(self.device))
            _lib.check(rc, "sss_row_norm_max")
        self._cmax = None

    def adopt(self, xb: torch.Tensor, id_offset: int = 0):
        """Use an existing CUDA [n, d] tensor of the index's element type as the corpus without
        copying it."""
        _lib.require_cuda(xb, "xb", self._tdtype)
        if xb.dim() != 2 or xb.shape[1] != self.d:
            raise ValueError("adopt: wrong shape")
        self._xb = self._store = xb
        self.id_offset = int(id_offset)
        self._cmax_t.zero_()
        self._norm_max(xb)
        self._split, self._split_done = None, 0
        self._f16, self._f16_done = None, 0
        self._amax_t.zero_()
        self._resid_t.zero_()
        self._auto_level.clear()
        self._auto_clean.clear()
        return self

    def corpus_max_norm(self) -> float:
        if self._cmax is None:
            self._cmax = float(self._cmax_t.item())
        return self._cmax

    # ------------------------------------------------------------------ device-level search
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def fused_ok(self, k: int) -> bool:
        return self.scan_for(k) != ""

    def search_fused(self, q: torch.Tensor, k: int, out=None, unproven_count=None):
        """Enqueue the fused MFMA scoring + top-k on the current stream; no host sync.
        Returns (D [nq,k] f32, I [nq,k] i64, status [nq] i32) CUDA tensors; rows with
        status != 0 must be re-run with ``search_exhaustive`` (``search`` does that).
        ``unproven_count``: optional CUDA int32 [1] tensor, incremented once per unproven query."""
        L = _lib.lib()
        _lib.require_cuda(q, "q", self._tdtype)
        nq, n = q.shape[0], self.ntotal
        if out is None:
            D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
            status = torch.empty((nq,), dtype=torch.int32, device=self.device)
        else:
            D, I, status = out
        mode = self.last_scan = self.prepare(k)
        if mode == "":
            raise _lib.SssError("search_fused: this index / k has no fused path (use search)")
        if mode == "long":
            ws = self._workspace(L.sss_ip_topk_long_workspace_bytes(nq, n, self.d, DTYPE_CODE[self.dtype]))
            image = self._f16 if self.dtype == "f32" else self._xb
            rc = L.sss_ip_topk_long(q.data_ptr(), nq, self._xb.data_ptr(), DTYPE_CODE[self.dtype], image.data_ptr(), self._c_shift,
                                    self.corpus_resid_norm() if self.dtype == "f32" else 0.0, n, self.d, k, self.id_offset,
                                    self.corpus_max_norm(), D.data_ptr(), I.data_ptr(), status.data_ptr(), ws.data_ptr(),
                                    ws.numel(), _lib.stream_ptr(self.device))
            _lib.check(rc, "sss_ip_topk_long")
            if unproven_count is not None:
                unproven_count += (status != 0).sum().to(torch.int32)
            return D, I, status
        if mode == "f16":
            nbytes = L.sss_ip_topk_f16_workspace_bytes(nq, n, self.d, k)
        else:
            nbytes = L.sss_ip_topk_workspace_bytes(nq, n, self.d, k, DTYPE_CODE[self.dtype])
        ws = self._workspace(nbytes)
        sbytes = L.sss_ip_topk_state_bytes(nq)
        if self._state is None or self._state.numel() < sbytes:
            self._state = torch.zeros(sbytes, dtype=torch.uint8, device=self.device)
        tail = (self.corpus_max_norm(), D.data_ptr(), I.data_ptr(), status.data_ptr(),
                0 if unproven_count is None else unproven_count.data_ptr(),
                self._state.data_ptr(), self._state.numel(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(self.device))
        if mode == "f16":
            rc = L.sss_ip_topk_f16(q.data_ptr(), nq, self._xb.data_ptr(), self._f16.data_ptr(), self._c_shift,
                                   self.corpus_resid_norm(), n, self.d, k, self.id_offset, *tail)
        elif mode == "split":
            rc = L.sss_ip_topk_split(q.data_ptr(), nq, self._xb.data_ptr(), self._split.data_ptr(), n, self.d, k,
                                     self.id_offset, *tail)
        else:
            rc = L.sss_ip_topk(q.data_ptr(), nq, self._xb.data_ptr(), n, self.d, k, DTYPE_CODE[self.dtype],
                               self.id_offset, *tail)
        if rc != 0:
            self._state = None          # re-made (zeroed) on the next call
        _lib.check(rc, "sss_ip_topk")
        return D, I, status

    def rung_scan(self) -> str:
        """The scan the threshold rung uses: the one-pass f16 image where the shape has one (cheapest pass
        over the corpus; its wider error window only means a few more rows to re-score), else the index's
        own rows."""
        if self.metric != "ip" or self.ntotal == 0:
            return ""
        if self.dtype != "f32":
            return "native" if self.d in FUSED_DIMS[self.dtype] else ""
        if not any(self._scan_served(s) for s in _LADDER):
            return ""                            # long rows: their scan IS a threshold scan; what it leaves is mass ties
        if self.scan == "auto":
            # an image that is already complete beats building another one (n * d * 2 bytes) for a handful of queries
            f16_ready = self._f16 is not None and self._f16_done == self.ntotal
            split_ready = self._split is not None and self._split_done == self.ntotal
            if self._scan_served("f16") and (f16_ready or not split_ready):
                return "f16"
            if split_ready and self._scan_served("split"):
                return "split"
        if self.scan == "f16" and self._scan_served("f16"):
            return "f16"
        if self.scan == "split" and self._scan_served("split"):
            return "split"
        return "f32" if self._scan_served("f32") else ("f16" if self._scan_served("f16") else "")

    def search_threshold(self, q: torch.Tensor, k: int, D: torch.Tensor, I: torch.Tensor, status: torch.Tensor, rows):
        """Threshold rung (``sss_ip_topk_threshold``) for the query rows ``rows`` a fused search left
        unproven: one more scan for just those queries keeps every corpus row that could still reach the
        k-th score already known (column k-1 of their rows of D) and re-scores them all.  Resolved rows of
        D / I are rewritten and their status set to 0; returns the rows still unproven."""
        mode = self.rung_scan()
        if mode == "" or rows.numel() == 0 or k > 8192:
            return rows
        L = _lib.lib()
        if mode == "f16":
            self._ensure_f16()
            image, code, shift, resid = self._f16, 3, self._c_shift, self.corpus_resid_norm()
        elif mode == "split":
            self._ensure_split()
            image, code, shift, resid = self._split, 2, 0, 0.0
        else:
            image, code, shift, resid = self._xb, DTYPE_CODE[self.dtype], 0, 0.0
        sel = rows.to(device=self.device, dtype=torch.int32).contiguous()
        n = self.ntotal
        ws = self._workspace(L.sss_ip_topk_threshold_workspace_bytes(sel.numel(), n, self.d, code))
        rc = L.sss_ip_topk_threshold(q.data_ptr(), sel.data_ptr(), sel.numel(), self._xb.data_ptr(), DTYPE_CODE[self.dtype],
                                     image.data_ptr(), code, shift, resid, n, self.d, k, self.id_offset,
                                     self.corpus_max_norm(), D.data_ptr(), I.data_ptr(), status.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _lib.stream_ptr(self.device))
        _lib.check(rc, "sss_ip_topk_threshold")
        return sel[status[sel.long()] != 0]

    def fix_unproven(self, q: torch.Tensor, k: int, D: torch.Tensor, I: torch.Tensor, status: torch.Tensor) -> int:
        """Make the result of a fused search exact for every query: unproven ones (status != 0) go through
        the threshold rung, what that leaves (more tied rows than its capacity) through the exhaustive
        kernels.  One host sync per stage that has work.  Returns the number of queries the fused scan
        had left unproven."""
        bad = torch.nonzero(status).flatten()
        nbad = int(bad.numel())
        self.last_rescan_queries, self.last_fallback_queries = nbad, 0
        if nbad:
            left = self.search_threshold(q, k, D, I, status, bad)
            if left.numel():
                self.last_fallback_queries = int(left.numel())
                self.search_exhaustive(q, k, D, I, left, bounded=True)
        self._note_fallbacks(k, q.shape[0], nbad)
        return nbad

    def search_exhaustive(self, q: torch.Tensor, k: int, D: torch.Tensor, I: torch.Tensor, rows=None, bounded=False):
        """Exhaustive exact path for query rows ``rows`` (all when None); writes into D / I.
        ``bounded``: D[rows, k-1] holds a valid lower bound of each query's k-th best score (what a
        fused search leaves behind for its unproven queries) -- lets the kernel skip the float64
        chain for rows that cannot matter."""
        L = _lib.lib()
        n = self.ntotal
        if rows is None:
            rows = torch.arange(q.shape[0], dtype=torch.int32, device=self.device)
        rows = rows.to(device=self.device, dtype=torch.int32).contiguous()
        per = max(1, min(65535, _EXHAUSTIVE_WS_BYTES // max(1, 4 * n)))
        metric = 0 if self.metric == "ip" else 1
        for lo in range(0, rows.numel(), per):
            sel = rows[lo:lo + per].contiguous()
            nbytes = L.sss_ip_topk_exhaustive_workspace_bytes(sel.numel(), n)
            ws = self._workspace(nbytes)
            if bounded and metric == 0:
                lb = D[sel.long(), k - 1].contiguous()
                rc = L.sss_ip_topk_exhaustive_lb(q.data_ptr(), sel.data_ptr(), sel.numel(), self._xb.data_ptr(), n,
                                                 self.d, k, DTYPE_CODE[self.dtype], self.id_offset, lb.data_ptr(),
                                                 D.data_ptr(), I.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 _lib.stream_ptr(self.device))
            else:
                rc = L.sss_ip_topk_exhaustive(q.data_ptr(), sel.data_ptr(), sel.numel(), self._xb.data_ptr(), n,
                                              self.d, k, DTYPE_CODE[self.dtype], self.id_offset, metric, D.data_ptr(),
                                              I.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(self.device))
            _lib.check(rc, "sss_ip_topk_exhaustive")

    def search_device(self, q: torch.Tensor, k: int):
        """Exact search, CUDA tensors in and out (syncs once to read the status vector)."""
        if q.dtype != self._tdtype:
            q = self._rows(q, "search")
        nq = q.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        self.last_fallback_queries = self.last_rescan_queries = 0
        if nq == 0:
            return D, I
        if self.ntotal == 0:
            D.fill_(-3.4028234663852886e38 if self.metric == "ip" else 3.4028234663852886e38)
            I.fill_(-1)
            return D, I
        if self.d % (4 if self.dtype == "f32" else 8):
            raise _lib.SssError("d must be a multiple of 4 (f32) / 8 (bf16)")
        if self.fused_ok(k):
            # the per-query workspace is 16 KB (fused scans) to 64 KB (long rows, threshold rung): the reference hands
            # `index.search` its whole test set at once (test_amazon_filterd.py:578), so large batches go in chunks
            step = SEARCH_CHUNK_LONG if self.scan_for(k) == "long" else SEARCH_CHUNK
            status = torch.empty((nq,), dtype=torch.int32, device=self.device)
            rescans = fallbacks = 0
            for lo in range(0, nq, step):
                hi = min(nq, lo + step)
                part = (D[lo:hi], I[lo:hi], status[lo:hi])
                self.search_fused(q[lo:hi], k, part)
                self.fix_unproven(q[lo:hi], k, *part)
                rescans += self.last_rescan_queries
                fallbacks += self.last_fallback_queries
            self.last_rescan_queries, self.last_fallback_queries = rescans, fallbacks
        else:
            self.last_fallback_queries = nq
            self.search_exhaustive(q, k, D, I)
        return D, I

    def search(self, x, k: int):
        """``index.search(x, k) -> (D, I)``: numpy in -> numpy out (faiss), tensor in -> tensors."""
        is_np = isinstance(x, np.ndarray)
        q = self._rows(x, "search")
        D, I = self.search_device(q, int(k))
        if is_np:
            return D.cpu().numpy(), I.cpu().numpy()
        return D, I


def build_index(emb, metric: str, device=None) -> FlatIndex:
    """Reference ``build_index(emb, metric)`` (test_amazon_filterd.py:207-223)."""
    if metric == "cos":
        index = FlatIndex(emb.shape[1], "ip", device)
        index.add(normalize(emb))
    elif metric == "l2":
        index = FlatIndex(emb.shape[1], "l2", device)
        index.add(emb)
    elif metric == "ip":
        index = FlatIndex(emb.shape[1], "ip", device)
        index.add(emb)
    else:
        raise RuntimeError("Unregnozed metric", metric)
    return index


# ------------------------------------------------------------------------------- binary codes
def pack_sign_bits(emb, code_bytes: int | None = None) -> torch.Tensor:
    """``np.packbits(((emb + 1) / 2).astype(int), axis=1)`` of the reference (fine_tune_ours.py:839-840,
    871-872) on the device: ``emb`` is the (+-1 valued) BinarizeHead output [n, c]; returns uint8
    [n, code_bytes] (default ceil(c / 8), zero padded like packbits)."""
    dev = _dev() if isinstance(emb, np.ndarray) or not emb.is_cuda else emb.device
    x = _as_device_f32(emb, dev)
    n, c = x.shape
    nbytes = (c + 7) // 8 if code_bytes is None else int(code_bytes)
    out = torch.empty((n, nbytes), dtype=torch.uint8, device=dev)
    rc = _lib.lib().sss_pack_sign_bits(x.data_ptr(), n, c, x.stride(0), out.data_ptr(), nbytes, _lib.stream_ptr(dev))
    _lib.check(rc, "sss_pack_sign_bits")
    return out


class BinaryFlatIndex:
    """``faiss.IndexBinaryFlat(nbits)`` as the reference uses it (fine_tune_ours.py:841-843,876):
    ``add(codes)`` with uint8 [n, nbits / 8] rows, ``search(codes, k) -> (D int32, I int64)`` by
    Hamming distance ascending, ties by ascending id.  Codes of 128 / 256 / 512 bits run on the
    fused scan; other widths (and unproven queries) go through the exhaustive kernels after the
    codes are zero padded to the next supported width (padding adds no distance)."""

    WIDTHS = (16, 32, 64)

    def __init__(self, nbits: int, device=None):
        if nbits % 8:
            raise ValueError("nbits must be a multiple of 8")
        self.d = int(nbits)
        self.code_bytes = nbits // 8
        if self.code_bytes > 64:
            raise ValueError("codes longer than 512 bits are not supported")
        self._w = next(w for w in self.WIDTHS if w >= self.code_bytes)     # stored (padded) row bytes
        self.device = _dev(device)
        self._codes = torch.empty((0, self._w), dtype=torch.uint8, device=self.device)
        self._store = self._codes       # backing storage of _codes (grown geometrically by add())
        self._ws = None
        self.id_offset = 0
        self.last_fallback_queries = 0

    @property
    def ntotal(self) -> int:
        return int(self._codes.shape[0])

    def _rows(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint8))
        x = x.to(self.device, torch.uint8)
        if x.dim() != 2 or x.shape[1] != self.code_bytes:
            raise ValueError(f"expected uint8 [n, {self.code_bytes}], got {tuple(x.shape)}")
        if self._w != self.code_bytes:
            pad = torch.zeros((x.shape[0], self._w), dtype=torch.uint8, device=self.device)
            pad[:, :self.code_bytes] = x
            x = pad
        return x.contiguous()

    def add(self, codes):
        x = self._rows(codes)
        n_old = self.ntotal
        if n_old + x.shape[0] > self._store.shape[0]:          # amortised growth, as FlatIndex.add
            cap = max(n_old + x.shape[0], 2 * self._store.shape[0])
            store = torch.empty((cap, self._w), dtype=torch.uint8, device=self.device)
            store[:n_old] = self._codes
            self._store = store
        self._store[n_old:n_old + x.shape[0]] = x
        self._codes = self._store[:n_old + x.shape[0]]

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def search(self, codes, k: int):
        is_np = isinstance(codes, np.ndarray)
        L = _lib.lib()
        q = self._rows(codes)
        nq, n, k = q.shape[0], self.ntotal, int(k)
        D = torch.full((nq, k), 0x7fffffff, dtype=torch.int32, device=self.device)
        I = torch.full((nq, k), -1, dtype=torch.int64, device=self.device)
        self.last_fallback_queries = 0
        if nq and n:
            st = _lib.stream_ptr(self.device)
            if k <= L.sss_hamming_topk_capacity(nq, n):
                ws = self._workspace(L.sss_hamming_topk_workspace_bytes(nq, n))
                status = torch.empty((nq,), dtype=torch.int32, device=self.device)
                rc = L.sss_hamming_topk(q.data_ptr(), nq, self._codes.data_ptr(), n, self._w, k, self.id_offset, D.data_ptr(),
                                        I.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(), st)
                _lib.check(rc, "sss_hamming_topk")
                bad = torch.nonzero(status).flatten().to(torch.int32)
            else:                               # k beyond the fused capacity: everything through the exhaustive path
                bad = torch.arange(nq, dtype=torch.int32, device=self.device)
            self.last_fallback_queries = int(bad.numel())
            per = max(1, min(65535, _EXHAUSTIVE_WS_BYTES // max(1, 2 * n)))
            for lo in range(0, bad.numel(), per):
                sel = bad[lo:lo + per].contiguous()
                ws = self._workspace(L.sss_hamming_topk_exhaustive_workspace_bytes(sel.numel(), n))
                rc = L.sss_hamming_topk_exhaustive(q.data_ptr(), sel.data_ptr(), sel.numel(), self._codes.data_ptr(), n,
                                                   self._w, k, self.id_offset, D.data_ptr(), I.data_ptr(), ws.data_ptr(),
                                                   ws.numel(), st)
                _lib.check(rc, "sss_hamming_topk_exhaustive")
        if is_np:
            return D.cpu().numpy(), I.cpu().numpy()
        return D, I

"""Dev helper (not product): per-workgroup timeline of k_scan from the stamped build of libsss
(scripts/dev/libsss_tl.so: see make_timeline_src.py).  Usage: timeline.py nq n d k"""
import sys, os, ctypes, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sessionsimilaritysearch_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("SSS_TL_LIB", "libsss_tl.so"))
import torch
from sessionsimilaritysearch_amd.index import FlatIndex, normalize_

nq, n, d, k = (int(v) for v in sys.argv[1:5])
scan = sys.argv[5] if len(sys.argv) > 5 else "split"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
c = torch.randn((n, d), device=dev, generator=g); normalize_(c)
q = torch.randn((nq, d), device=dev, generator=g); normalize_(q)
idx = FlatIndex(d, "ip", dev, scan=scan).adopt(c)
idx.corpus_max_norm()
out = idx.search_fused(q, k)
for _ in range(20):
    idx.search_fused(q, k, out)
torch.cuda.synchronize()
L = _lib.lib()
nwg = 512
W = 16
buf = (ctypes.c_ulonglong * (nwg * W))()
fn = L.sss_debug_timeline
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = fn(buf, nwg * W)
a = np.array(buf, dtype=np.uint64).reshape(nwg, W).astype(np.int64)
rt0, rt1 = a[:, 0], a[:, 7]
t = a[:, 1:6]
start_skew_us = (rt0 - rt0.min()) / 100.0
end_us = (rt1 - rt0.min()) / 100.0
dur_us = (rt1 - rt0) / 100.0
cyc = t[:, 4] - t[:, 0]
ghz = cyc / (dur_us * 1e3)
print(json.dumps(dict(rc=rc, nq=nq, n=n, d=d, k=k, scan=scan,
    kernel_us=float(end_us.max()), start_skew_us_max=float(start_skew_us.max()),
    wg_dur_us=dict(min=float(dur_us.min()), med=float(np.median(dur_us)), max=float(dur_us.max())),
    end_us=dict(min=float(end_us.min()), med=float(np.median(end_us)), max=float(end_us.max())),
    clock_ghz_med=float(np.median(ghz)),
    prologue_cyc_med=float(np.median(t[:, 2] - t[:, 0])), loop_cyc_med=float(np.median(t[:, 3] - t[:, 2])),
    loop_cyc_min=float((t[:, 3] - t[:, 2]).min()), loop_cyc_max=float((t[:, 3] - t[:, 2]).max()),
    tail_cyc_med=float(np.median(t[:, 4] - t[:, 3])),
    boot_done_cyc_med=float(np.median(a[:, 8] - t[:, 2])), loop_start_us_med=float(np.median((a[:, 9] - rt0.min()) / 100.0)), rare_cyc_med=float(np.median(a[:, 10])), rare_med=float(np.median(a[:, 6])), rare_max=int(a[:, 6].max()))))

live = dur_us > 1.0
G_ = (nq + 255) // 256
bid = np.arange(nwg)
print("WG durations (us) of the live workgroups: pct 5/25/50/75/95/100 =", [round(float(np.percentile(dur_us[live], p)), 1) for p in (5, 25, 50, 75, 95, 100)])
print("by XCD (bid & 7): mean / max", [(round(float(dur_us[live & ((bid & 7) == x)].mean()), 1), round(float(dur_us[live & ((bid & 7) == x)].max()), 1)) for x in range(8)])
print("by query group:   mean / max", [(round(float(dur_us[live & (((bid >> 3) % G_) == g)].mean()), 1), round(float(dur_us[live & (((bid >> 3) % G_) == g)].max()), 1)) for g in range(G_)])
order = np.argsort(-dur_us)[:12]
print("slowest workgroups (bid, us, rare entries):", [(int(b), round(float(dur_us[b]), 1), int(a[b, 6])) for b in order])
tb = (ctypes.c_ulonglong * 192)()
L.sss_debug_tiles.argtypes = [ctypes.c_void_p]
L.sss_debug_tiles(tb)
tt = np.array(tb, dtype=np.uint64).astype(np.int64)
for o in (0, 96):
    v = tt[o:o + 96]; v = v[v > 0]
    print("tile-end deltas (cycles) wg", 0 if o == 0 else 101, ":", (v[0] - a[0 if o == 0 else 101, 3]), list(np.diff(v))[:70])

wb = (ctypes.c_ulonglong * 672)()
L.sss_debug_w.argtypes = [ctypes.c_void_p]
L.sss_debug_w(wb)
w = np.array(wb, dtype=np.uint64).astype(np.int64)
nt = int((tt[:96] > 0).sum())
print("wg 0 wave 0 per tile: vmcnt wait", list(w[:nt])[:40])
print("wg 0 wave 0 per tile: barrier wait", list(w[96:96 + nt])[:40])
print("wg 0 wave 0 per tile: rare cycles", list(np.diff(np.concatenate([[0], w[192:192 + nt]])))[:40])
print("wg 0 wave 0 per tile: tile_top cycles", [int(v) for v in w[288:288 + nt]][:40])
print("wg 0 wave 0 per tile: publish cycles", [int(v) for v in w[480:480 + nt]][:40])
print("wg 0 wave 0 per tile: tau_fetch cycles", [int(v) for v in w[576:576 + nt]][:40])
print("wg 0 wave 0 per tile: mfma_sub cycles", [int(v) for v in w[384:384 + nt]][:40])

"""Dev helper: copy the judged summaries of a scripts/collect_profiles.sh run from gpurun_out/prof_<tag>/ into
profiles/ (kernel stats, bench lines, per-kernel PMC means) and write profiles/<tag>_traffic.json, which bench.py
reads for `roofline.traffic` -- every entry stamped with the sha256 of the csrc/scan.hip it was measured on
(bench.py prints null for a stale entry).      python scripts/summarize_profiles.py r03"""
import csv, hashlib, json, os, re, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof_" + tag), os.path.join(root, "profiles")
sha = hashlib.sha256(open(os.path.join(root, "sessionsimilaritysearch_amd", "csrc", "scan.hip"), "rb").read()).hexdigest()[:16]
for name in ("bench_default", "bench_split", "bench_force_collectives", "bench_c4_10m", "bench_c5_bf16", "bench_c3", "encoder", "search_shapes"):
    if os.path.exists(f"{src}/{name}_kernel_stats.csv"):
        shutil.copy(f"{src}/{name}_kernel_stats.csv", f"{dst}/{tag}_{name}_kernel_stats.csv")
    lj = f"{src}/{name}_line.json"
    if os.path.exists(lj) and os.path.getsize(lj) > 0:
        shutil.copy(lj, f"{dst}/{tag}_{name}_line.json")
if os.path.exists(f"{src}/components.txt"):
    shutil.copy(f"{src}/components.txt", f"{dst}/{tag}_components.txt")
for name in ("search_shapes", "encoder"):
    if os.path.exists(f"{src}/{name}.log"):
        lines = [l for l in open(f"{src}/{name}.log") if l.startswith("{") or l.startswith("sessions=")]
        open(f"{dst}/{tag}_{name}_lines.txt", "w").writelines(lines)

DT = {"0": "f32", "1": "bf16", "2": "split", "3": "f16"}


def scan_type(kernel):
    """k_scan<row bytes, tile rows, element type, waves, threshold form> -> 'f32' | 'bf16' | 'split' | 'f16'"""
    m = re.search(r"k_scan<\s*\d+,\s*\d+,\s*(\d)", kernel)
    return DT.get(m.group(1)) if m else None


def summarize(name):
    path = f"{src}/{name}_counters.csv"
    if not os.path.exists(path):
        return {}
    agg = defaultdict(list)
    rows = list(csv.DictReader(open(path)))
    # k_scan_long runs once per sample level of a search (three at 1M x 1600, K = 100), always in the same order: the
    # dispatches are labelled by their position in that cycle, so that the LAST level -- the full scan -- has a line of its own
    long_ids = sorted({int(r["Dispatch_Id"]) for r in rows if "k_scan_long" in r["Kernel_Name"]})
    level_of = {d: i % 3 + 1 for i, d in enumerate(long_ids)}
    for r in rows:
        kn = r["Kernel_Name"]
        if "k_scan" in kn or "k_select" in kn or "k_bound" in kn or "k_long_setup" in kn:
            name_ = kn.split("(")[0]
            if "k_scan_long" in kn:
                name_ += f" level {level_of[int(r['Dispatch_Id'])]} of 3"
            agg[(name_, r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(f"{dst}/{tag}_{name}.csv", "w") as f:
        f.write("kernel,counter,launches,mean_per_launch\n")
        for (k, c), v in sorted(agg.items()):
            f.write(f'"{k}",{c},{len(v)},{sum(v) / len(v):.3f}\n')
    return {(scan_type(k), c): sum(v) / len(v) for (k, c), v in agg.items() if scan_type(k)}


fetch, write = summarize("pmc_fetch"), summarize("pmc_write")
for name in ("pmc_mfma", "pmc_issue", "pmc_mfma_c5", "pmc_long_tcc", "pmc_long_sq", "pmc_long_sq2", "pmc_long_fetch"):
    summarize(name)
fetch4 = summarize("pmc_fetch_c4")
fetch_split, fetch5 = summarize("pmc_fetch_split"), summarize("pmc_fetch_c5")
n1, nq1, d1 = 1000000, 1024, 128
doc = ("HBM bytes per launch of the scan kernel from rocprofv3 --pmc passes (scripts/collect_profiles.sh). FETCH_SIZE / "
       "WRITE_SIZE are reported in KB; FETCH_SIZE counts 16-B/lane streaming reads at half (MI355X_MICROARCH.md 'HBM'): "
       "bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024.  Key = scan:d:nq:rows_per_gpu; algorithmic_bytes = the scanned corpus "
       "image once + the query batch.  scan_hip_sha16 = sha256 of the csrc/scan.hip the counters were collected on.")
traffic = {"_doc": doc}


def entry(key, fetch_kb, write_kb, alg, source):
    if fetch_kb is None:
        return
    traffic[key] = {"fetch_bytes": round(fetch_kb * 2048), "write_bytes": None if write_kb is None else round(write_kb * 1024),
                    "total_bytes": round(fetch_kb * 2048 + (write_kb or 0) * 1024), "algorithmic_bytes": alg,
                    "scan_hip_sha16": sha, "source": source}


entry(f"f16:{d1}:{nq1}:{n1}", fetch.get(("f16", "FETCH_SIZE")), write.get(("f16", "WRITE_SIZE")), n1 * d1 * 2 + nq1 * d1 * 4,
      f"profiles/{tag}_pmc_fetch.csv + {tag}_pmc_write.csv")
entry(f"f32:{d1}:{nq1}:{n1}", fetch.get(("f32", "FETCH_SIZE")), write.get(("f32", "WRITE_SIZE")), n1 * d1 * 4 + nq1 * d1 * 4,
      f"profiles/{tag}_pmc_fetch.csv + {tag}_pmc_write.csv")
entry(f"split:{d1}:{nq1}:{n1}", fetch_split.get(("split", "FETCH_SIZE")), None, n1 * d1 * 4 + nq1 * d1 * 4, f"profiles/{tag}_pmc_fetch_split.csv")
entry("bf16:256:4096:10000000", fetch5.get(("bf16", "FETCH_SIZE")), None, 10000000 * 256 * 2 + 4096 * 256 * 2, f"profiles/{tag}_pmc_fetch_c5.csv")
n4 = 10000000
entry(f"f32:{d1}:{nq1}:{n4}", fetch4.get(("f32", "FETCH_SIZE")), None, n4 * d1 * 4 + nq1 * d1 * 4, f"profiles/{tag}_pmc_fetch_c4.csv")
entry(f"f16:{d1}:{nq1}:{n4}", fetch4.get(("f16", "FETCH_SIZE")), None, n4 * d1 * 2 + nq1 * d1 * 4, f"profiles/{tag}_pmc_fetch_c4.csv")
json.dump(traffic, open(f"{dst}/{tag}_traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))

"""Dev helper: copy the judged summaries of a scripts/collect_profiles.sh run from gpurun_out/prof_<tag>/ into
profiles/ (kernel stats, bench lines, per-kernel PMC means) and write profiles/<tag>_traffic.json, which bench.py
reads for `roofline.traffic`.      python scripts/summarize_profiles.py r02"""
import csv, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof_" + tag), os.path.join(root, "profiles")
for name in ("bench_default", "bench_split", "bench_f32", "bench_c4_10m", "bench_c4_10m_f32", "bench_c5_bf16", "bench_c3",
             "encoder", "search_shapes"):
    shutil.copy(f"{src}/{name}_kernel_stats.csv", f"{dst}/{tag}_{name}_kernel_stats.csv")
    lj = f"{src}/{name}_line.json"
    if os.path.exists(lj) and os.path.getsize(lj) > 0:
        shutil.copy(lj, f"{dst}/{tag}_{name}_line.json")
if os.path.exists(f"{src}/components.txt"):
    shutil.copy(f"{src}/components.txt", f"{dst}/{tag}_components.txt")
for name in ("search_shapes", "encoder"):
    lines = [l for l in open(f"{src}/{name}.log") if l.startswith("{") or l.startswith("sessions=")]
    open(f"{dst}/{tag}_{name}_lines.txt", "w").writelines(lines)


def summarize(name):
    rows = list(csv.DictReader(open(f"{src}/{name}_counters.csv")))
    agg = defaultdict(list)
    for r in rows:
        kn = r["Kernel_Name"]
        if "k_scan<" in kn or "k_select" in kn:
            agg[(kn.split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(f"{dst}/{tag}_{name}.csv", "w") as f:
        f.write("kernel,counter,launches,mean_per_launch\n")
        for (k, c), v in sorted(agg.items()):
            f.write(f'"{k}",{c},{len(v)},{sum(v) / len(v):.3f}\n')
    return {(k, c): sum(v) / len(v) for (k, c), v in agg.items()}


scan = lambda d: [v for (k, c), v in d.items() if "k_scan<" in k][0]
fetch, write = scan(summarize("pmc_fetch")), scan(summarize("pmc_write"))
for name in ("pmc_mfma", "pmc_issue", "pmc_lds", "pmc_mfma_split", "pmc_mfma_f32", "pmc_mfma_c5"):
    summarize(name)
fetch_split, fetch_f32, fetch5 = scan(summarize("pmc_fetch_split")), scan(summarize("pmc_fetch_f32")), scan(summarize("pmc_fetch_c5"))
n1, nq1, d1 = 1000000, 1024, 128
traffic = {
    "_doc": "HBM bytes per launch of the scan kernel from rocprofv3 --pmc passes (scripts/collect_profiles.sh). FETCH_SIZE / "
            "WRITE_SIZE are reported in KB; FETCH_SIZE counts 16-B/lane streaming reads at half (MI355X_MICROARCH.md 'HBM'): "
            "bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024.  Key = scan:d:nq:rows_per_gpu; algorithmic_bytes = the scanned corpus "
            "image once + the query batch.",
    f"f16:{d1}:{nq1}:{n1}": {"fetch_bytes": round(fetch * 2048), "write_bytes": round(write * 1024),
                             "total_bytes": round(fetch * 2048 + write * 1024),
                             "algorithmic_bytes": n1 * d1 * 2 + nq1 * d1 * 4,
                             "source": f"profiles/{tag}_pmc_fetch.csv + {tag}_pmc_write.csv"},
    f"split:{d1}:{nq1}:{n1}": {"fetch_bytes": round(fetch_split * 2048), "write_bytes": None, "total_bytes": round(fetch_split * 2048),
                               "algorithmic_bytes": n1 * d1 * 4 + nq1 * d1 * 4, "source": f"profiles/{tag}_pmc_fetch_split.csv"},
    f"f32:{d1}:{nq1}:{n1}": {"fetch_bytes": round(fetch_f32 * 2048), "write_bytes": None, "total_bytes": round(fetch_f32 * 2048),
                             "algorithmic_bytes": n1 * d1 * 4 + nq1 * d1 * 4, "source": f"profiles/{tag}_pmc_fetch_f32.csv"},
    "bf16:256:4096:10000000": {"fetch_bytes": round(fetch5 * 2048), "write_bytes": None, "total_bytes": round(fetch5 * 2048),
                               "algorithmic_bytes": 10000000 * 256 * 2 + 4096 * 256 * 2,
                               "source": f"profiles/{tag}_pmc_fetch_c5.csv"},
}
json.dump(traffic, open(f"{dst}/{tag}_traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))

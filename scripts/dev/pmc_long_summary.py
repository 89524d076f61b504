"""Summarises gpurun_out/pmc_long/*.csv: per counter, the value of the LARGEST k_scan_long dispatch (the last level)."""
import csv, glob, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_long"
for f in sorted(glob.glob(os.path.join(root, "*.csv"))):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(dict)
    for r in rows:
        if "k_scan_long" not in r.get("Kernel_Name", ""): continue
        by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        by[r["Dispatch_Id"]]["_grid"] = float(r.get("Grid_Size", 0))
    if not by: print(os.path.basename(f), "no k_scan_long rows"); continue
    gmax = max(v["_grid"] for v in by.values())
    last = [v for v in by.values() if v["_grid"] == gmax]
    keys = sorted(k for k in last[0] if k != "_grid")
    print(os.path.basename(f), f"(mean over {len(last)} last-level dispatches, grid {int(gmax)})")
    for k in keys:
        print(f"   {k:40s} {sum(v.get(k, 0.0) for v in last) / len(last):.4g}")

"""Dev helper: mean counter value per kernel from a rocprofv3 --pmc counter_collection csv (FETCH_SIZE / WRITE_SIZE in KB;
FETCH_SIZE x 2 on gfx950 for 16-B/lane streaming reads -- MI355X_MICROARCH.md 'HBM')."""
import csv, sys, collections
acc = collections.defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "k_scan" not in k and "k_select" not in k:
        continue
    mean = sum(v) / len(v)
    extra = f"  = {mean * 2048 / 1e6:.1f} MB (x2)" if c == "FETCH_SIZE" else f"  = {mean * 1024 / 1e6:.1f} MB" if c == "WRITE_SIZE" else ""
    print(f"{k:70s} {c:12s} n={len(v):4d} mean={mean:.1f}{extra}")

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/long; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_search_gpu.py -x -q -k "long or c3 or thr or rung or k500 or large_k" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash scripts/dev/abn.sh 3 "libsss_base.so tree" 1024,1000000,1600,100 200,1000000,1600,100 1024,1000000,1600,10 200,100000,1600,100 > $O/ab.txt 2>&1
cat $O/ab.txt | cut -c1-130
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 scripts/quick_search_bench.py 1024,1000000,1600,100 > $O/st.log 2>&1
cat $O/st/*/*kernel_stats.csv | grep -E "k_select_all|k_scan_long|k_bound|k_thr_prepare" | cut -c1-150 | tee $O/stats.txt
rm -rf $O/st
timeout -k 10 300 python3 scripts/dev/fuzz_search.py 41 24 long > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
python3 - <<'PY'
import re, collections
acc=collections.defaultdict(list); cur=None
for l in open('gpurun_out/long/ab.txt'):
    if l.startswith('=='): cur=l.split()[1]; continue
    m=re.search(r'"nq": (\d+), "n": (\d+), "d": (\d+), "k": (\d+).*"ms": ([\d.]+)', l)
    if m: acc[(m.group(1),m.group(2),m.group(4),cur)].append(float(m.group(5)))
for k in sorted(acc): print(k, acc[k], round(sum(acc[k])/len(acc[k]),4))
PY

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sel; mkdir -p $O
timeout -k 10 400 python3 scripts/dev/fuzz_search.py 31 40 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
timeout -k 10 600 python -m pytest tests/test_search_gpu.py tests/test_search_fuzz_gpu.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash scripts/dev/abn.sh 4 "libsss_base.so tree" 1024,125000,128,10,f16 1024,1000000,128,10,f16 1024,125000,128,10,f32mfma 1024,1000000,128,10,split > $O/ab.txt 2>&1
python3 - <<'PY'
import re, collections
acc=collections.defaultdict(list); cur=None
for l in open('gpurun_out/sel/ab.txt'):
    if l.startswith('=='): cur=l.split()[1]; continue
    m=re.search(r'"n": (\d+).*"scan": "(\w+)", "ms": ([\d.]+)', l)
    if m: acc[(m.group(1),m.group(2),cur)].append(float(m.group(3)))
for k in sorted(acc): print(k, acc[k], round(sum(acc[k])/len(acc[k]),4))
PY

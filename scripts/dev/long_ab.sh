#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/long; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_search_gpu.py -x -q -k "long or c3 or thr or rung or k500 or large_k" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash scripts/dev/abn.sh 2 "libsss_dj.so tree" 1024,1000000,1600,100 200,1000000,1600,100 1024,1000000,1600,10 1024,100000,1600,100 1024,4000000,128,500,split > $O/ab.txt 2>&1
cat $O/ab.txt | cut -c1-130
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 scripts/quick_search_bench.py 1024,1000000,1600,100 > $O/st.log 2>&1
cat $O/st/*/*kernel_stats.csv | grep -E "k_select_all|k_scan_long|k_bound|k_thr_prepare" | cut -c1-150 | tee $O/stats.txt
rm -rf $O/st

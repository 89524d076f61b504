#!/bin/bash
# correctness subset, A/B timing and a FETCH_SIZE pass for the leader/follower refresh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lead; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_search_gpu.py -x -q > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash scripts/dev/abn.sh 4 "libsss_base.so tree" 1024,125000,128,10,f16 1024,1000000,128,10,f16 1024,10000000,128,10,f16 1024,1000000,128,10,split 1024,1000000,128,10,f32mfma > $O/ab.txt 2>&1
cat $O/ab.txt | cut -c1-120
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 scripts/quick_search_bench.py 1024,1000000,128,10,f16 1024,125000,128,10,f16 > $O/pf.log 2>&1
cat $O/pf/*/*counter_collection.csv > $O/pf_counters.csv; rm -rf $O/pf
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 scripts/quick_search_bench.py 1024,1000000,128,10,f16 > $O/pw.log 2>&1
cat $O/pw/*/*counter_collection.csv > $O/pw_counters.csv; rm -rf $O/pw
python3 scripts/dev/pmc_quick.py $O/pf_counters.csv $O/pw_counters.csv | tee $O/pmc.txt

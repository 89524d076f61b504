#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun, from the
# repo root):   bash scripts/collect_profiles.sh r03
# Kernel-trace/stats passes and PMC passes are separate rocprofv3 runs (gpurun refuses a combination;
# FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# The program sits directly after `--` (no env/bash hop).  The default bench runs BOTH legs (f32-MFMA scan = the
# top-level line, scan='auto' = fast_path), so one PMC pass sees both scan kernels; they are told apart by the
# element-type template argument of k_scan<row bytes, tile rows, type, waves, threshold form>.
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="bench.py --steps 20 --warmup 3 --no-cpu-baseline"
stats() {   # name, args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.log" 2>&1
    cat "$OUT/$name"/*/*kernel_stats.csv > "$OUT/${name}_kernel_stats.csv" 2>/dev/null
    grep '^{"metric"' "$OUT/$name.log" > "$OUT/${name}_line.json" 2>/dev/null
    echo "[$name] done"
}
pmc() {     # name, counters, args...
    local name=$1; local ctr=$2; shift 2
    rocprofv3 --pmc $ctr --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.log" 2>&1
    cat "$OUT/$name"/*/*counter_collection.csv > "$OUT/${name}_counters.csv" 2>/dev/null
    echo "[$name] done"
}
stats bench_default $BENCH
stats bench_split $BENCH --scan split
stats bench_c4_10m bench.py --steps 5 --warmup 2 --no-cpu-baseline --corpus-rows 10000000 --corpus-source random
stats bench_c5_bf16 bench.py --steps 10 --warmup 2 --no-cpu-baseline --corpus-rows 10000000 --corpus-source random --dtype bf16 --d 256 --nq 4096
stats bench_c3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload c3 --corpus-rows 1000000
stats encoder scripts/bench_encoder.py 1024
stats search_shapes scripts/quick_search_bench.py 1024,125000,128,10,f16 1024,1000000,128,10,f16 1024,10000000,128,10,f16 1024,1000000,256,10,f16 1024,125000,128,10,split 1024,1000000,128,10,split 1024,1000000,64,10,split 1024,1000000,128,100,split 1024,125000,128,10,f32mfma 1024,1000000,128,10,f32mfma 1024,125000,64,10,f32mfma 1024,125000,128,100,f32mfma 1024,1000000,1600,100 200,1000000,1600,100 1024,1000000,1600,10 1024,100000,1600,100
PM="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-shapes"
pmc pmc_fetch FETCH_SIZE $PM
pmc pmc_write WRITE_SIZE $PM
pmc pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" $PM
pmc pmc_issue "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $PM
pmc pmc_fetch_split FETCH_SIZE $PM --scan split
PM5="bench.py --steps 4 --warmup 1 --no-cpu-baseline --corpus-rows 10000000 --corpus-source random --dtype bf16 --d 256 --nq 4096"
pmc pmc_fetch_c5 FETCH_SIZE $PM5
pmc pmc_mfma_c5 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" $PM5
python3 scripts/bench_components.py 2>/dev/null | grep '^{' > "$OUT/components.txt"
echo "[components] done"
ls -la "$OUT"

#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun, from the repo root), in two
# calls (a gpurun call is capped at 20 minutes):
#     bash scripts/collect_profiles.sh r04 stats      kernel-trace / --stats passes
#     bash scripts/collect_profiles.sh r04 pmc        counter passes
# Kernel-trace/stats passes and PMC passes are separate rocprofv3 runs (gpurun refuses a combination; FETCH_SIZE and
# WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").  The program sits directly after `--`
# (no env/bash hop).  The default bench runs BOTH legs (f32-MFMA scan = the top-level line, scan='auto' = fast_path) plus
# the `configs` (C4 / C5 / C3) and `reference_shapes` objects in ONE process, so its kernel-stats table covers every
# scan kernel of the round; they are told apart by the template arguments of
# k_scan<row bytes, tile rows, type (0 f32 / 1 bf16 / 2 split / 3 f16), waves, threshold form, append form>.
set -u
TAG=${1:-r04}
WHAT=${2:-stats}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
stats() {   # name, args...
    local name=$1; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.log" 2>&1
    cat "$OUT/$name"/*/*kernel_stats.csv > "$OUT/${name}_kernel_stats.csv" 2>/dev/null
    grep '^{"metric"' "$OUT/$name.log" > "$OUT/${name}_line.json" 2>/dev/null
    rm -rf "$OUT/$name"                         # (the raw traces are large; only the summaries travel back)
    echo "[$name] done"
}
pmc() {     # name, counters, args...
    local name=$1; local ctr=$2; shift 2
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.log" 2>&1
    cat "$OUT/$name"/*/*counter_collection.csv > "$OUT/${name}_counters.csv" 2>/dev/null
    rm -rf "$OUT/$name"
    echo "[$name] done"
}
if [ "$WHAT" = stats ]; then
    stats bench_default bench.py --steps 20 --warmup 3 --no-cpu-baseline
    stats bench_split bench.py --steps 20 --warmup 3 --no-cpu-baseline --scan split --no-configs --no-reference-shapes
    stats bench_force_collectives bench.py --steps 20 --warmup 3 --no-cpu-baseline --force-collectives --no-configs --no-reference-shapes
    stats encoder scripts/bench_encoder.py 1024
    stats search_shapes scripts/quick_search_bench.py 1024,125000,128,10,f16 1024,1000000,128,10,f16 1024,10000000,128,10,f16 1024,1000000,256,10,f16 1024,125000,128,10,split 1024,1000000,128,10,split 1024,1000000,64,10,split 1024,1000000,128,100,split 1024,125000,128,10,f32mfma 1024,1000000,128,10,f32mfma 1024,125000,64,10,f32mfma 1024,125000,128,100,f32mfma 1024,1000000,1600,100 200,1000000,1600,100 1024,1000000,1600,10 1024,100000,1600,100 1024,1000000,1024,100
    python3 scripts/bench_components.py 2>/dev/null | grep '^{' > "$OUT/components.txt"
    echo "[components] done"
else
    PM="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-shapes --no-configs"
    pmc pmc_fetch FETCH_SIZE $PM
    pmc pmc_write WRITE_SIZE $PM
    pmc pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" $PM
    pmc pmc_issue "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $PM
    pmc pmc_fetch_split FETCH_SIZE $PM --scan split
    PM4="bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-reference-shapes --no-configs --corpus-rows 10000000 --corpus-source random --config-index 4"
    pmc pmc_fetch_c4 FETCH_SIZE $PM4
    PM5="bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-reference-shapes --no-configs --corpus-rows 10000000 --corpus-source random --dtype bf16 --d 256 --nq 4096 --config-index 5"
    pmc pmc_fetch_c5 FETCH_SIZE $PM5
    pmc pmc_mfma_c5 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" $PM5
    # the long-row scan (1M x 1600, K = 100): L2 hit rate, wave states, LDS, matrix pipe, fetch
    LS="scripts/quick_search_bench.py 1024,1000000,1600,100"
    pmc pmc_long_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" $LS
    pmc pmc_long_sq "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" $LS
    pmc pmc_long_sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" $LS
    pmc pmc_long_fetch FETCH_SIZE $LS
fi
ls -la "$OUT" | head -60

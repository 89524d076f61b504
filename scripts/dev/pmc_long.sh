#!/bin/bash
# PMC passes on the long-row scan (1M x 1600, K = 100, 1024 queries): what bounds k_scan_long's last level?
# Separate rocprofv3 runs per counter group (slot limits), the program directly after `--`.
set -u
OUT=gpurun_out/pmc_long
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SHAPE=${1:-1024,1000000,1600,100}
pass() { local name=$1; local ctr=$2
  timeout -k 5 150 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/$name" -- python3 scripts/quick_search_bench.py $SHAPE > "$OUT/$name.log" 2>&1
  cat "$OUT/$name"/*/*counter_collection.csv > "$OUT/${name}.csv" 2>/dev/null; echo "[$name] $(wc -l < $OUT/${name}.csv 2>/dev/null) rows"; }
[ "${SKIP_TCC:-0}" = 1 ] || pass tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
# (a TA_* pass -- TA_TA_BUSY_sum, TA_ADDR_STALLED_BY_TC_CYCLES_sum, ... -- hung the profiler on this pool for 7 minutes: not collected)
pass sq "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
pass sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
pass fetch "FETCH_SIZE"

#!/bin/bash
# Collect PMC counters for a python command in separate passes (gfx950: 8 SQ slots, 4 TCC slots per pass).
# usage: tools/pmc_passes.sh <outdir> <python script + args...>
set -u
OUT=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp
i=0
while read -r SET; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$R/$OUT/pass$i" -- python3 "$@" > "$R/$OUT/pass$i.log" 2>&1
done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE
SETS

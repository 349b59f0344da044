#!/bin/bash
# memory-pipeline PMC passes (TA / TD / TCP busy + stalls).  usage: tools/pmc_mem.sh <outdir> <python script + args...>
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
TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
SETS

#!/bin/bash
# VALU / LDS / wait PMC passes for the pooling kernels.  usage: tools/pmc_gather2.sh <outdir> <python script + args...>
set -u
OUT=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp
i=0
while read -r SET; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $SET --output-format csv -d "$R/$OUT/pass$i" -- python3 "$@" > "$R/$OUT/pass$i.log" 2>&1
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM
SETS

#!/bin/bash
# rocprofv3 --pmc passes (SQ counters) of a python command, one pass per counter set; per-kernel sums go to gpurun_out/<name>_pmc.txt
# usage: tools/prof_pmc.sh <name> <kernel substring> <script.py> [args...]
set -u
NAME=$1; KERN=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "import torch" >/dev/null 2>&1
cd /tmp
: > "$OUT/${NAME}_pmc.txt"
i=0
while read -r SET; do
  i=$((i+1))
  rm -rf /tmp/pmc_${NAME}_$i
  timeout -s KILL ${PROF_TIMEOUT:-200} rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_${NAME}_$i -- python3 "$R/$1" "${@:2}" > "$OUT/${NAME}_pmc$i.log" 2>&1
  f=$(find /tmp/pmc_${NAME}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$KERN" >> "$OUT/${NAME}_pmc.txt" <<'PY'
import csv, sys, collections
f, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for row in csv.DictReader(open(f)):
    if kern in row["Kernel_Name"]:
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in acc:
    print(f"{k}: {acc[k] / max(n[k], 1):.1f} per dispatch ({n[k]} dispatches)")
PY
done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
SETS
cat "$OUT/${NAME}_pmc.txt"

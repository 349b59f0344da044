#!/bin/bash
# Regenerate the judged artefacts under gpurun_out/prof (copy into profiles/ afterwards):
#   bench JSON line, rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE PMC passes, SQ passes of the
#   persistent kernels; the same for the shipped 5-layer MultiviewC config (the pipelined kernel).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python3 -c "import torch" >/dev/null 2>&1
timeout -s KILL 700 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout -s KILL 300 python3 bench.py --workload multiviewc_156x156x5 --cpu-seconds 0 > "$OUT/bench_mc5.json" 2>> "$OUT/bench.err"
cd /tmp
SHORT="--cpu-seconds 0 --steps 5 --warmup 2 --fp32-steps 0 --c5-steps 0 --rotate 0 --proxy-steps 0 --loop-steps 0 --train-steps 0"
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --cpu-seconds 0 --proxy-steps 0 --loop-steps 0 > "$OUT/trace.log" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace5" -- python3 "$R/bench.py" --workload multiviewc_156x156x5 --cpu-seconds 0 --fp32-steps 0 --rotate 0 --proxy-steps 0 --loop-steps 0 > "$OUT/trace5.log" 2>&1
for W in multiviewc_200x200x1 multiviewc_156x156x5; do
  T=$([ $W = multiviewc_200x200x1 ] && echo "" || echo "5")
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch$T" -- python3 "$R/bench.py" --workload $W $SHORT > "$OUT/fetch$T.log" 2>&1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write$T" -- python3 "$R/bench.py" --workload $W $SHORT > "$OUT/write$T.log" 2>&1
  (cd "$R" && python3 tools/pmc_to_traffic.py "$OUT/fetch$T" "$OUT/write$T" "$OUT/pmc_traffic$T.json")
done
i=0
while read -r SET; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $SET --output-format csv -d "$OUT/sq$i" -- python3 "$R/bench.py" --fp32-steps 3 --train-steps 0 --cpu-seconds 0 --steps 5 --warmup 2 --c5-steps 0 --rotate 0 --proxy-steps 0 --loop-steps 0 > "$OUT/sq$i.log" 2>&1
done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
GRBM_GUI_ACTIVE
SETS
# the same four passes on the shipped five-layer MultiviewC frame: the pipelined kernel where it ships (round-5 verdict: SQ_WAIT_ANY there)
mkdir -p "$OUT/mc5"
i=0
while read -r SET; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $SET --output-format csv -d "$OUT/mc5/sq$i" -- python3 "$R/bench.py" --workload multiviewc_156x156x5 $SHORT > "$OUT/mc5/sq$i.log" 2>&1
done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
GRBM_GUI_ACTIVE
SETS
cd "$R"
python3 tools/pmc_sq_summary.py "$OUT" "$OUT/fused_sq.json"
python3 tools/pmc_sq_summary.py "$OUT/mc5" "$OUT/fused_sq_mc5.json"
rm -rf "$OUT/mc5"/sq[0-9] "$OUT/mc5"/*.log; rmdir "$OUT/mc5" 2>/dev/null
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/trace5" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_mc5.csv" \;
rm -rf "$OUT/trace" "$OUT/trace5" "$OUT"/fetch* "$OUT"/write* "$OUT"/sq[0-9] "$OUT"/*.log
head -c 600 "$OUT/bench.json"; echo; head -8 "$OUT/kernel_stats.csv"; head -c 1500 "$OUT/fused_sq.json"

#!/bin/bash
# Regenerate the judged artefacts under gpurun_out/prof (copy into profiles/ afterwards):
#   bench JSON line, rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE PMC passes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python3 -c "import torch" >/dev/null 2>&1
timeout -s KILL 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"     # also leaves the TunableOp file (unused now)
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --cpu-seconds 0 > "$OUT/trace.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/bench.py" --cpu-seconds 0 --steps 5 --warmup 2 > "$OUT/fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/bench.py" --cpu-seconds 0 --steps 5 --warmup 2 > "$OUT/write.log" 2>&1
cd "$R"
python3 tools/pmc_to_traffic.py "$OUT/fetch" "$OUT/write" "$OUT/pmc_traffic.json"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/trace" "$OUT/fetch" "$OUT/write"
head -c 1500 "$OUT/bench.json"; echo; head -8 "$OUT/kernel_stats.csv"

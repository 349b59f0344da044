#!/bin/bash
# rocprofv3 --kernel-trace --stats of a python command; the kernel_stats.csv goes to gpurun_out/<name>_kernel_stats.csv.
# usage: tools/prof_stats.sh <name> <script.py> [args...]      (every step under its own timeout)
set -u
NAME=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "import torch" >/dev/null 2>&1
cd /tmp
rm -rf /tmp/prof_$NAME
timeout -s KILL ${PROF_TIMEOUT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -- python3 "$R/$1" "${@:2}" > "$OUT/${NAME}_prof.log" 2>&1
find /tmp/prof_$NAME -name "*kernel_stats.csv" -exec cp {} "$OUT/${NAME}_kernel_stats.csv" \;
head -${PROF_LINES:-8} "$OUT/${NAME}_kernel_stats.csv" | cut -c1-200

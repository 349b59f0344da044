#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pipe_frame.py -x -q -m gpu > gpurun_out/ab_test.log 2>&1; echo "test rc=$?" >> gpurun_out/ab_test.log
for cfg in "multiviewc_200x200x1" "multiviewc_156x156x5" "wildtrack_120x360x8" "multiviewx_160x250x8"; do
  for rep in 1 2; do
    echo "== new $cfg"; timeout 600 python tools/bench_pipe.py $cfg 2>&1 | grep -iE "pipe_collapse  |Error" | head -2
    echo "== base $cfg"; VFA_AMD_LIB=/root/repo/tools/ab/libvfa_base.so timeout 600 python tools/bench_pipe.py $cfg 2>&1 | grep -iE "pipe_collapse  |Error" | head -2
  done
done > gpurun_out/ab.log 2>&1
tail -3 gpurun_out/ab_test.log; cat gpurun_out/ab.log

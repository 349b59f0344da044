#!/bin/bash
# Builds one library per compile-time ablation mask of the PRODUCTION pipelined kernel (VFA_PIPE_ABLATE, vfa_pipe.hip) into
# tools/scratch/ablate/libvfa_hip_ab<mask>.so (git-ignored; they travel to the GPU box).  Time them with
#     python tools/time_pipe_libs.py <workload> [--cams=N]
# Masks: 1 no window fills, 2 no pooling, 4 no MFMAs, 64 loop + tables + barriers only.  Results of an ablated build are meaningless.
set -e
cd "$(dirname "$0")/../vfa_amd/csrc"
OUT=../../tools/scratch/ablate
mkdir -p $OUT
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -std=c++17 -Wall -I../../include"
OBJS="vfa_kernels.o vfa_collapse.o vfa_collapse_gemm.o vfa_fused.o vfa_eval.o vfa_integral.o vfa_lateral.o vfa_grad.o"
make -j8 >/dev/null
for m in ${MASKS:-1 2 4 6 7 64}; do
  ( /opt/rocm/bin/hipcc $FLAGS -DVFA_PIPE_ABLATE=$m ${EXTRA} -c -o $OUT/vfa_pipe_ab$m.o vfa_pipe.hip &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT/libvfa_hip_ab$m.so $OBJS $OUT/vfa_pipe_ab$m.o && rm $OUT/vfa_pipe_ab$m.o ) &
done
wait
ls -la $OUT

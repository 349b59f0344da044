#!/usr/bin/env python3
"""Forward + backward of the aggregate on one workload, both training paths: fused forward + recomputing backward
(`vfa_op.FUSED_TRAIN`, the default) and the unfused kernels with vox / lin saved by autograd."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vfa_amd  # noqa: E402
from vfa_amd import ops  # noqa: E402
from vfa_amd.synthetic import make_workload  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--workload", default="multiviewc_200x200x1")
p.add_argument("--steps", type=int, default=5)
a = p.parse_args()
dev = torch.device("cuda:0")
wl = make_workload(a.workload, channels=256, seed=0)
n = wl["n_cam"]
torch.manual_seed(0)
mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
lats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev).requires_grad_(True) for s in range(3)]
calibs, grid = wl["calibs"].to(dev), wl["grid"].to(dev)


def step():
    out = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
    out.sum().backward()


from vfa_amd import vfa_op  # noqa: E402

for fused in (True, False):
    vfa_op.FUSED_TRAIN = fused
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    with ops.KernelTimer() as kt:
        step()
        torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 1e9
    torch.cuda.reset_peak_memory_stats()
    print(f"{a.workload} {'fused forward + recomputing backward' if fused else 'unfused (vox, lin saved)'}: forward+backward {dt * 1e3:.2f} ms/step, peak memory {peak:.2f} GB")
    for k, v in kt.summary().items():
        print(f"  {k}: {v['launches']} launches/step, {v['ms']:.3f} ms/step")

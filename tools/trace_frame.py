#!/usr/bin/env python3
"""A few frames of the shipped per-frame path, for `rocprofv3 --kernel-trace` (timeline of one frame: tools/trace_frame.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfa_amd
from vfa_amd import vfa_op
from vfa_amd.synthetic import make_workload

dev = torch.device("cuda:0")
wl = make_workload("multiviewc_200x200x1", channels=256, seed=0)
n = wl["n_cam"]
mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
feats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev) for s in range(3)]
calibs, grid = wl["calibs"].to(dev), wl["grid"].to(dev)
out = torch.empty(grid.shape[1] * grid.shape[2], 256, device=dev)
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
        vfa_op.fused_frame(mods, feats, calibs, grid, out=out)
torch.cuda.synchronize()

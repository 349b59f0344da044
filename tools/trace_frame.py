#!/usr/bin/env python3
"""A few frames of the shipped per-frame path, for `rocprofv3 --kernel-trace` (timeline of one frame: tools/trace_summary.py).

    python tools/trace_frame.py [frames] [workload] [--cams=N]      (the first N cameras of the rig: a rank's share)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfa_amd
from vfa_amd.synthetic import make_workload

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
frames = int(pos[0]) if pos else 60
name = pos[1] if len(pos) > 1 else "multiviewc_200x200x1"
cams = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--cams=")), "0"))
dev = torch.device("cuda:0")
wl = make_workload(name, channels=256, seed=0, **({"n_cam": cams} if cams else {}))
n = wl["n_cam"]
mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
feats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev) for s in range(3)]
calibs, grid = wl["calibs"].to(dev), wl["grid"].to(dev)
with torch.no_grad():
    for _ in range(frames):
        out = vfa_amd.aggregate_views(*mods, *feats, calibs, grid)
torch.cuda.synchronize()

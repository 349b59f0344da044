#!/usr/bin/env python3
"""Time vfa_project_gather_backward_f32 per scale (run-combined atomics vs. LDS-privatised scatter)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vfa_amd  # noqa: E402
from vfa_amd import _lib, ops  # noqa: E402
from vfa_amd.synthetic import make_workload  # noqa: E402

dev = torch.device("cuda:0")
wl = make_workload(sys.argv[1] if len(sys.argv) > 1 else "multiviewc_200x200x1", channels=256, seed=0)
n = wl["n_cam"]
mod = vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
zl, co = mod._kernel_geometry(dev)
calibs = wl["calibs"].reshape(n, 12).to(dev)
grid = wl["grid"].reshape(-1, 3).to(dev)
kind = _lib.CONV_KIND[wl["args"].data]
size = wl["args"].image_size[::-1]
modes = [("atomics", 0, 0), ("lds, 32 in a line", 1, 0), ("lds, 4 x 8 patches", 1, 1)]
gw = wl["grid"].shape[-2]
for s in range(3):
    feat = torch.cat([wl["features"][c][s] for c in range(n)]).to(dev)
    Hf, Wf = feat.shape[-2:]
    gvox = torch.randn(n, grid.shape[0], zl.numel() * 256, device=dev)
    line = [f"stride {8 << s} ({Hf}x{Wf})"]
    for name, cache, dbg in modes:
        for _ in range(2):
            ops.project_gather_backward(gvox, (n, Hf + 2, Wf + 2, 256), calibs, grid, zl, co, kind, size, kernel=None if cache else 'direct', grid_w=gw if dbg else 0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.project_gather_backward(gvox, (n, Hf + 2, Wf + 2, 256), calibs, grid, zl, co, kind, size, kernel=None if cache else 'direct', grid_w=gw if dbg else 0)
        e1.record()
        torch.cuda.synchronize()
        line.append(f"{name} {e0.elapsed_time(e1) / 5 * 1e3:.0f} us")
    print("  ".join(line))

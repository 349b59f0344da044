#!/usr/bin/env python3
"""Time the two pooling kernels of `vfa_project_gather_f32` per scale of one workload (back-to-back launches)."""
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
kind, size = _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1]


def timeit(f, reps=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for s in range(3):
    feat = torch.cat([wl["features"][c][s] for c in range(n)]).to(dev)
    integral = ops.integral_image(feat)
    vox = torch.empty((n, grid.shape[0], zl.numel() * 256), device=dev)
    row = [f"{k} {timeit(lambda: ops.project_gather(integral, calibs, grid, zl, co, kind, size, out=vox, kernel=k)):.0f} us"
           for k in ("tap_cache", "direct")]
    ref = ops.project_gather(integral, calibs, grid, zl, co, kind, size, kernel="direct")
    same = torch.equal(ref.view(torch.int32), ops.project_gather(integral, calibs, grid, zl, co, kind, size, kernel="tap_cache").view(torch.int32))
    print(f"stride {8 << s}: " + " | ".join(row) + f" | bitwise equal: {same}")

#!/usr/bin/env python3
"""Launch only the fused projection+pooling kernel a few times on one scale of a workload (target for rocprofv3)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vfa_amd  # noqa: E402
from vfa_amd import _lib, ops  # noqa: E402
from vfa_amd.synthetic import make_workload  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--workload", default="multiviewc_200x200x1")
p.add_argument("--scale", type=int, default=2)
p.add_argument("--launches", type=int, default=5)
p.add_argument("--ws", action="store_true")
a = p.parse_args()
dev = torch.device("cuda:0")
wl = make_workload(a.workload, channels=256, seed=0)
n = wl["n_cam"]
mod = vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
zl, co = mod._kernel_geometry(dev)
grid_flat = wl["grid"].reshape(-1, 3).to(dev).contiguous()
calibs = wl["calibs"].reshape(n, 12).to(dev).contiguous()
lat = torch.cat([wl["features"][c][a.scale] for c in range(n)]).to(dev)
integral = ops.integral_image(lat)
vox = torch.empty((n, grid_flat.shape[0], zl.numel() * 256), device=dev)
ws = torch.empty(_lib.lib().vfa_gather_workspace_bytes(n, zl.numel(), grid_flat.shape[0]), dtype=torch.uint8, device=dev)
for _ in range(a.launches):
    if a.ws:
        ops.project_gather_ws(integral, calibs, grid_flat, zl, co, _lib.CONV_KIND[wl["args"].data],
                              wl["args"].image_size[::-1], out=vox, workspace=ws)
        continue
    ops.project_gather(integral, calibs, grid_flat, zl, co, _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1],
                       out=vox)
torch.cuda.synchronize()
print("done")

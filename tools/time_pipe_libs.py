#!/usr/bin/env python3
"""Launch time of `vfa_pipe_collapse_relu_sum_f32` under several builds of the library (one child process per build: the library is
loaded once per process): the shipped one and the compile-time ablations tools/ablate_pipe.sh left in tools/scratch/ablate/.

    python tools/time_pipe_libs.py [workload] [--cams=N] [--libs=a.so,b.so]
"""
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys
sys.path.insert(0, %r)
import torch, vfa_amd
from vfa_amd import _lib, ops
from vfa_amd.synthetic import make_workload
name, cams = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
wl = make_workload(name, channels=256, seed=0, **({"n_cam": cams} if cams else {}))
n = wl["n_cam"]
torch.manual_seed(0)
mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
nl = mods[0].num_grid_layer
lats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev) for s in range(3)]
calibs, grid = wl["calibs"].to(dev), wl["grid"].to(dev)
L, W = grid.shape[1:3]
rows = L
while rows > 4 and ops.pipe_workspace_bytes(n, rows, W, nl, 3) > (3 << 30):
    rows = max(4, ((rows + 1) // 2 + 3) // 4 * 4)
if rows < L:
    L = rows
    grid = grid[:, :L].contiguous()
zl, co = mods[0]._kernel_geometry(dev)
sizes = [tuple(l.shape[-2:]) for l in lats]
with torch.no_grad():
    integrals = ops.integral_images(lats)
    ws = ops.pipe_records(calibs, grid, zl, co, _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1], sizes,
                          weights=[m.collapse.weight for m in mods])
    ops.pipe_balance(ws, n, (L, W), nl, 3)
    biases = [m.collapse.bias for m in mods]
    out = torch.empty(L * W, 256, device=dev)
    reps = int(sys.argv[3])
    for _ in range(max(3, reps // 2)):
        ops.pipe_collapse(integrals, biases, ws, (L, W), nl, out=out)
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.pipe_collapse(integrals, biases, ws, (L, W), nl, out=out)
        e1.record(); e1.synchronize()
        best.append(e0.elapsed_time(e1) / reps * 1e3)
import numpy as np
lay = ops.pipe_workspace_layout(n, L, W, nl, 3)
cyc = ws[lay["balance"] + 4096:lay["balance"] + 4096 + 8 * 512].cpu().numpy().view(np.uint64).astype(np.float64)
cyc = cyc[cyc > 0]
print(json.dumps({"us": round(sorted(best)[1], 1), "rows": L, "nl": nl, "n": n, "wgs": int(cyc.size), "wg_cycles_max_over_mean": round(float(cyc.max() / cyc.mean()), 3),
                  "wg_cycles_min_over_mean": round(float(cyc.min() / cyc.mean()), 3), "wg_kcycles_mean": round(float(cyc.mean()) / 1e3, 1),
                  "wg_kcycles_max": round(float(cyc.max()) / 1e3, 1)}))
""" % REPO

name = next((a for a in sys.argv[1:] if not a.startswith("--")), "multiviewc_156x156x5")
cams = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--cams=")), "0")
reps = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--reps=")), "20")
libs = next((a.split("=")[1].split(",") for a in sys.argv[1:] if a.startswith("--libs=")), None)
rts = next((a.split("=")[1].split(",") for a in sys.argv[1:] if a.startswith("--rt=")), [""])  # VFA_AMD_PIPE_RT settings to loop over ("" = the library's choice)
if libs is None:
    libs = [os.path.join(REPO, "vfa_amd", "csrc", "libvfa_hip.so")] + sorted(glob.glob(os.path.join(REPO, "tools", "scratch", "ablate", "*.so")))
for rnd in range(2):  # (twice, alternating: devices ramp their clocks)
    for lib in libs:
        for rt in rts:
            env = dict(os.environ, VFA_AMD_LIB=lib)
            if rt:
                env["VFA_AMD_PIPE_RT"] = rt
            r = subprocess.run([sys.executable, "-c", CHILD, name, cams, reps], env=env, capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
            print(f"{name} cams={cams} {os.path.basename(lib):28s} rt={rt or '-'} {line}", flush=True)

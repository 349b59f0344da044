#!/usr/bin/env python3
"""Per-workgroup cycles of the pipelined frame kernel (production template: the balance state of the workspace holds the cycles of
the last launch) against what each workgroup had to do under the balanced work cuts: a least-squares fit of candidate cost terms and
what the heaviest share would be if the cuts followed the fitted model.

    python tools/fit_pipe_cost.py [workload] [--cams=N]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import vfa_amd  # noqa: E402
from vfa_amd import _lib, ops  # noqa: E402
from vfa_amd.synthetic import make_workload  # noqa: E402

name = next((a for a in sys.argv[1:] if not a.startswith("--")), "multiviewc_156x156x5")
cams = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--cams=")), "0"))
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
    out = torch.empty(L * W, 256, device=dev)
    for _ in range(5):
        ops.pipe_collapse(integrals, [m.collapse.bias for m in mods], ws, (L, W), nl, out=out)
    torch.cuda.synchronize()
lay = ops.pipe_workspace_layout(n, L, W, nl, 3)
host = ws.cpu().numpy()
tiles = lay["tiles_l"] * lay["tiles_w"]
K = lay["n_chunks"]
start = host[lay["chunks"]:lay["chunks"] + 4 * (K + 1)].view(np.int32)
rank = host[lay["ranks"]:lay["ranks"] + 4 * (K + 1)].view(np.int32)
bal = host[lay["balance"]:lay["balance"] + 4096].view(np.int32)
nblk = int(bal[513])
bounds = bal[:nblk + 1]
cyc = host[lay["balance"] + 4096:lay["balance"] + 4096 + 8 * nblk].view(np.uint64).astype(np.float64)
live = [host[lay["live"][s]:lay["live"][s] + 4 * tiles].view(np.uint32) for s in range(3)]
hdrs = [host[lay["hdrs"][s]:lay["hdrs"][s] + tiles * nl * n * 32].view(np.uint32).reshape(tiles, nl, n, 8) for s in range(3)]
steps = 2 * nl * n * 3 * tiles // nblk
rt = (4 if steps >= 250 else 1) if n <= 2 else (4 if steps >= 1600 else (2 if steps >= 1200 else 1))
rt = int(os.environ.get("VFA_AMD_PIPE_RT", rt))
runs = (tiles + rt - 1) // rt


def groups_of_run(r):
    out = []
    for s in range(3):
        subs = [(t, v) for t in range(r * rt, min(tiles, (r + 1) * rt)) for v in range(n) if (int(live[s][t]) >> v) & 1]
        for g0 in range(0, len(subs), 4):
            out.append((s, subs[g0:g0 + 4]))
    return out


feat = np.zeros((nblk, 7))
for wg in range(nblk):
    c0, c1 = int(bounds[wg]), int(bounds[wg + 1])
    rb, kb, re, ke = int(start[c0]), int(rank[c0]), int(start[c1]), int(rank[c1])
    for r in range(rb, min(re + (1 if ke > 0 else 0), runs)):
        gs = groups_of_run(r)
        lo, hi = (kb if r == rb else 0), (ke if r == re else len(gs))
        for s, subs in gs[lo:hi]:
            sets = (len(subs) + 1) // 2
            feat[wg, 0] += 4 * nl * sets                      # steps with work
            feat[wg, 1] += 4 * nl * (2 - sets)                # empty steps
            feat[wg, 6] += 1                                  # groups
            for t, v in subs:
                h = hdrs[s][t, :, v]                          # (nl, 8)
                lv = (h[:, 0] & 1) == 1
                direct = lv & ((h[:, 0] & 2) == 2)
                feat[wg, 2] += 4 * float(h[lv & ~direct, 1].sum()) / 64.0   # window slots fetched (x 4 quarters), in units of 64
                feat[wg, 3] += 4 * int(direct.sum())          # sub-tile steps pooled from L2
                feat[wg, 4] += 4 * int((~lv).sum())           # sub-tile steps without a live box in the layer
                feat[wg, 5] += 4 * int(lv.sum())              # sub-tile steps with pooling
on = cyc > 0
A, y = feat[on], cyc[on]
names = ["step with work", "empty step", "64 window slots", "sub-tile step from L2", "dead sub-tile step", "live sub-tile step", "group"]
for cols, label in (([0, 1, 6], "steps + groups (the shipped model's terms, refitted)"), ([0, 1, 6, 3], "+ L2 sub-tile steps"), ([0, 1, 6, 3, 2], "+ window slots"),
                    ([0, 1, 6, 3, 2, 4], "+ dead sub-tile steps"), ([1, 6, 3, 2, 4, 5], "per live sub-tile step instead of per step")):
    coef, *_ = np.linalg.lstsq(A[:, cols], y, rcond=None)
    pred = A[:, cols] @ coef
    print(f"{name} n={n} rt={rt}: {label}: " + ", ".join(f"{names[c]} {v:.0f}" for c, v in zip(cols, coef))
          + f" | residual rms {np.sqrt(np.mean((pred - y) ** 2)) / y.mean():.3f}; measured max/mean {y.max() / y.mean():.3f}; "
          f"max/mean if the cuts had equalised THIS model {1 + (y - pred).max() / y.mean():.3f}")

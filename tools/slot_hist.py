"""Distribution of tap-window sizes (slots of 1 KiB) over the (view, tile, scale) items of the bench frame."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vfa_amd
from vfa_amd import _lib, ops
from vfa_amd.synthetic import make_workload
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "multiviewc_200x200x1"
wl = make_workload(name, channels=256, seed=0, device=dev)
n = wl["n_cam"]
mod = vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
zl, co = mod._kernel_geometry(dev)
hws = [tuple(wl["features"][0][s].shape[-2:]) for s in range(3)]
ws = ops.frame_records(wl["calibs"], wl["grid"], zl, co, _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1], hws)
L, W = wl["grid"].shape[1:3]
lay = ops.frame_workspace_layout(n, L, W, 3)
host = ws.cpu().numpy()
tiles = lay["tiles_l"] * lay["tiles_w"]
allv = []
for s in range(3):
    off = lay["hdrs"][s]
    hdr = host[off:off + n * tiles * 32].view(np.uint32).reshape(n * tiles, 8)
    live = (hdr[:, 0] & 1) == 1
    sl = hdr[:, 1][live].astype(np.int64)
    allv.append(sl)
    print(f"scale {s}: live {live.sum()}  " + "  ".join(f"<={c}: {np.mean(sl <= c):.3f}" for c in (24, 32, 40, 47, 48, 62, 64, 80, 94, 96, 126)) + f"  p50 {np.median(sl):.0f} p90 {np.percentile(sl, 90):.0f} max {sl.max()}")
    # consecutive live items of a tile (views in order): does the pair fit a ring of R slots?
    h2 = hdr.reshape(n, tiles, 8)
    for R in (94, 96):
        fit = tot = 0
        prev = None
        for t in range(tiles):
            for v in range(n):
                if h2[v, t, 0] & 1 and h2[v, t, 1] <= R:
                    if prev is not None:
                        tot += 1
                        fit += (prev + h2[v, t, 1]) <= R
                    prev = int(h2[v, t, 1])
        print(f"   ring {R}: consecutive pairs that fit together {fit / max(tot, 1):.3f}")

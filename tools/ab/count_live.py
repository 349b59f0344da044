import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import vfa_amd
from vfa_amd import _lib, ops
from vfa_amd.synthetic import make_workload
dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    wl = make_workload(name, channels=256, seed=0)
    n = wl["n_cam"]
    mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
    zl, co = mods[0]._kernel_geometry(dev)
    grid = wl["grid"].to(dev)
    L, W = grid.shape[1:3]
    nl = mods[0].num_grid_layer
    sizes = [tuple(s) for s in wl["feat_sizes"]]
    ws = ops.pipe_records(wl["calibs"].to(dev), grid, zl, co, _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1], sizes,
                          weights=[m.collapse.weight for m in mods])
    torch.cuda.synchronize()
    host = ws.cpu().numpy()
    lay = ops.pipe_workspace_layout(n, L, W, nl, 3)
    tiles = lay["tiles_l"] * lay["tiles_w"]
    tot_live = tot_proc = 0
    for s in range(3):
        hdr = host[lay["hdrs"][s]:lay["hdrs"][s] + tiles * nl * n * 32].view(np.uint32).reshape(tiles, nl, n, 8)
        live_item = (hdr[..., 0] & 1).astype(bool)              # (tile, layer, view)
        live_view = live_item.any(axis=1)                          # (tile, view): the group lists
        tot_live += int(live_item.sum())
        tot_proc += int(live_view.sum()) * nl
    print(f"{name}: live (tile, layer, view, scale) items {tot_live}, multiplied {tot_proc}: {1 - tot_live / tot_proc:.3f} of the row blocks are zeros")

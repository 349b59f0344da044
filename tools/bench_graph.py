#!/usr/bin/env python3
"""Eager vs hipGraph replay of the aggregate on a small (launch-bound) grid."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vfa_amd  # noqa: E402
from vfa_amd.graph import GraphedAggregate  # noqa: E402
from vfa_amd.synthetic import make_workload  # noqa: E402

dev = torch.device("cuda:0")
for crop in ((16, 16), (40, 40), (100, 100), (200, 200)):
    wl = make_workload("multiviewc_200x200x1", channels=256, seed=0, device=dev)
    grid = wl["grid"][:, :crop[0], :crop[1]].contiguous()
    torch.manual_seed(0)
    mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
    lats = [torch.cat([wl["features"][c][s] for c in range(7)]) for s in range(3)]
    g = GraphedAggregate(*mods, *lats, wl["calibs"], grid)

    def timeit(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    with torch.no_grad():
        te = timeit(lambda: vfa_amd.aggregate_views(*mods, *lats, wl["calibs"], grid))
        tg = timeit(lambda: g(*g.static_in))  # producers write straight into the static buffers: no copy
    print(f"grid {crop[0]}x{crop[1]}x1, 7 cameras: eager {te:.3f} ms/frame, hipGraph replay {tg:.3f} ms/frame ({te / tg:.2f}x)")

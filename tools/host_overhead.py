#!/usr/bin/env python3
"""Host time to queue one frame of the shipped path (no synchronisation inside the loop) against the GPU time of the frame: how far
ahead the launching thread runs.  With few cameras per rank (strong scaling over 8 GPUs: one camera each) the GPU part shrinks and the
host part does not."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfa_amd
from vfa_amd.synthetic import make_workload

dev = torch.device("cuda:0")
for n_cam in (7, 2, 1):
    wl = make_workload("multiviewc_200x200x1", channels=256, seed=0, n_cam=n_cam)
    mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
    lats = [torch.cat([wl["features"][c][s] for c in range(n_cam)]).to(dev) for s in range(3)]
    calibs, grid = wl["calibs"].to(dev), wl["grid"].to(dev)
    with torch.no_grad():
        for _ in range(200):
            vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
        torch.cuda.synchronize()
        gc.collect(); gc.disable()
        for reps in (20, 200):
            for _ in range(100):
                vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"{n_cam} cameras, {reps:3d} frames: host queues a frame in {(t1 - t0) / reps * 1e6:6.1f} us, the GPU finishes one every {(t2 - t0) / reps * 1e6:6.1f} us")
        gc.enable()

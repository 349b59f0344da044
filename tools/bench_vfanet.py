#!/usr/bin/env python3
"""End-to-end VFANet inference (ResNet-18 backbone + laterals in PyTorch/MIOpen, projector + aggregation in HIP, heads)
on synthetic 7 x 720 x 1280 images: where the frame time goes."""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vfa_amd  # noqa: E402
from vfa_amd.synthetic import make_workload  # noqa: E402
from vfa_amd.vfanet import VFANet  # noqa: E402

dev = torch.device("cuda:0")
wl = make_workload("multiviewc_200x200x1", channels=256, seed=0, device=dev)
args = SimpleNamespace(data="MultiviewC", image_size=(720, 1280))
torch.manual_seed(0)
net = VFANet(args, grid_height=wl["grid_height"], cube_size=wl["cube_size"], angle_range=360).to(dev).eval()
images = torch.rand(7, 3, 720, 1280, device=dev)
calibs, grid = wl["calibs"], wl["grid"]


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    lats = net.laterals(images)
    ortho = net.ortho_features(images, calibs, grid)
    t_all = timeit(lambda: net(images, calibs, grid))
    t_lat = timeit(lambda: net.laterals(images))
    t_proj = timeit(lambda: vfa_amd.aggregate_views(net.vfa8, net.vfa16, net.vfa32, *lats, calibs, grid))
    t_heads = timeit(lambda: (net.map_classifier(net.fuse(ortho)), net.tytx_pred(ortho), net.thtwtl_pred(ortho),
                              net.orient_pred(net.fuse(ortho))))
print(f"VFANet forward, 7 cameras 720x1280 -> 200x200 BEV: {t_all:.2f} ms/frame; backbone + laterals {t_lat:.2f} ms, "
      f"projection + aggregation (HIP) {t_proj:.2f} ms, BEV heads {t_heads:.2f} ms")

#!/usr/bin/env python3
"""Random small frames through the fused per-frame path against the multi-kernel window path (bit-exact pooling kernel + MFMA
collapse kernel on the same records): random rigs (cameras inside and outside the field, 1..12 of them), grid crops, feature sizes,
launch widths (reserved CUs), accumulate mode and row-slot budgets.  Prints the worst error / tolerance; exits non-zero on a miss.
Not part of the test suite (minutes of GPU time); run it after touching vfa_fused.hip."""
import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vfa_amd
from vfa_amd import _lib, ops
from vfa_amd.synthetic import look_at_camera
from vfa_amd.utils import make_grid

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
RTOL, ATOL_REL = 1e-4, 1e-5
worst = 0.0
for case in range(n_cases):
    image_size = (720, 1280)
    n = int(rng.integers(1, 13))
    cams = []
    for _ in range(n):
        inside = rng.random() < 0.3
        pos = (rng.uniform(200, 3500), rng.uniform(200, 3500), rng.uniform(150, 400)) if inside else \
              (rng.uniform(-1500, 5200), rng.uniform(-1500, 5200), rng.uniform(200, 900))
        tgt = (rng.uniform(800, 2900), rng.uniform(800, 2900), 0.0)
        cams.append(look_at_camera(pos, tgt, rng.uniform(400, 1100), (1280, 720)))
    calibs = torch.tensor(np.stack(cams), dtype=torch.float32).to(dev)
    cube = float(rng.choice([18.75, 37.5, 75.0, 150.0]))
    full = make_grid(world_size=(3750, 3750), cube_LW=[cube, cube], dataset="MultiviewC")
    L0, W0 = full.shape[:2]
    L, W = int(rng.integers(1, min(L0, 70) + 1)), int(rng.integers(1, min(W0, 90) + 1))
    l0, w0 = int(rng.integers(0, L0 - L + 1)), int(rng.integers(0, W0 - W + 1))
    grid = full[l0:l0 + L, w0:w0 + W].contiguous().to(dev)
    ns = int(rng.integers(1, 4))
    sizes = [(90, 160), (45, 80), (23, 40)][:ns] if rng.random() < 0.6 else [(int(rng.integers(5, 60)), int(rng.integers(5, 90))) for _ in range(ns)]
    args = SimpleNamespace(data="MultiviewC", image_size=image_size)
    torch.manual_seed(case)
    gh = float(rng.choice([160, 300]))
    mods = [vfa_amd.VFA(256, grid_height=gh, cube_size=(cube, cube, gh), args=args).to(dev) for _ in range(ns)]
    with torch.no_grad():
        for m in mods:
            m.collapse.weight.mul_(3.0)
            m.collapse.bias.uniform_(-0.3, 0.1)
    gen = torch.Generator().manual_seed(1000 + case)
    lats = [torch.relu(torch.randn(n, 256, h, w, generator=gen)).to(dev) for h, w in sizes]
    zl, co = mods[0]._kernel_geometry(dev)
    kind = _lib.CONV_KIND["MultiviewC"]
    weights = [m.layer_major_weight() for m in mods]
    biases = [m.collapse.bias for m in mods]
    row_slots = None if rng.random() < 0.7 else int(rng.integers(0, 6))
    reserved = int(rng.choice([0, 0, 8, 16, 100, 200]))
    accumulate = rng.random() < 0.3
    with torch.no_grad():
        ws = ops.frame_records(calibs, grid, zl, co, kind, image_size[::-1], sizes, weights=weights, row_slots=row_slots)
        integrals = ops.integral_images(lats)
        base = torch.randn(L * W, 256, device=dev) if accumulate else None
        got = ops.pool_collapse(integrals, biases, ws, (L, W), out=None if base is None else base.clone(), accumulate=accumulate,
                                reserved_cus=reserved)
        want = torch.zeros(L * W, 256, dtype=torch.float64, device=dev) if base is None else base.double()
        vox = torch.empty(n, L * W, 256, device=dev)
        for k in range(ns):
            ops.pool_windows(integrals[k], ws, (L, W), ns, k, out=vox)
            want += torch.relu(vox.double() @ weights[k].double().T + biases[k].double()).sum(0)
    torch.cuda.synchronize()
    scale = max(want.abs().max().item(), 1e-30)
    err = ((got.double() - want).abs() / (RTOL * want.abs() + ATOL_REL * scale)).max().item()
    lay = ops.frame_workspace_layout(n, L, W, ns)
    direct = int(ws[lay["counter"]:lay["counter"] + 4].cpu().numpy().view(np.uint32)[0])
    worst = max(worst, err)
    print(f"case {case:3d}: {n:2d} cameras, grid {L:3d} x {W:3d} (cube {cube}), {ns} scales {sizes}, direct items {direct:4d}, row slots {row_slots}, "
          f"reserved CUs {reserved:3d}, accumulate {int(accumulate)}: |err| / tol = {err:.3f}", flush=True)
    if not (err <= 1.0) or not torch.isfinite(got).all():
        print("MISMATCH")
        sys.exit(1)
print(f"worst |err| / tolerance over {n_cases} frames: {worst:.3f}")

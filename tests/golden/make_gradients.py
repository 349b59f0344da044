#!/usr/bin/env python3
"""Gradient fixtures: the reference's OWN ``backward()`` through the path (CPU, fp32) -- what ``trainer.py:41`` computes.

Runs only in the build container (the reference lives at /root/reference; nothing of it is copied: the script imports
``vfa.model.vfa_op.VFA`` with the stand-in modules of ``make_golden.py`` and records inputs + outputs).  Per case:

  * module level (``grad_*.npz``): one ``VFA.forward`` (vfa_op.py:61-125) with ``requires_grad`` on the feature map, ``collapse.weight``
    and ``collapse.bias``; loss = sum(ortho * probe); ``loss.backward()`` -> d feature, d weight, d bias, in fp32 (the reference as it
    runs) AND in float64 (the same module cast with ``.double()``: the reference's own arithmetic at twice the width).  The distance
    between the two is the rounding noise of the reference's fp32 gradient -- the yardstick of the tolerance in
    tests/test_reference_gradients.py (the gradient of the feature map inherits the cancellation of the integral image: vfa_op.py:
    172-173 backwards is two reverse cumsums of scattered tap weights);
  * frame level (``grad_frame_*.npz``): the camera loop of ``VFANet.forward`` (vfanet.py:64-82: three ``VFA`` modules per camera,
    ``f8 + f16 + f32``, ``ortho +=``) on 256-channel lateral maps, single-layer grid -- the shape the fused training node
    (``vfa_amd.vfa_op._FusedFrameTrain``) takes; same outputs per lateral map / module.

Usage:  python tests/golden/make_gradients.py      (writes tests/golden/grad_*.npz)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the stand-in modules, imports the reference)

import torch  # noqa: E402

from vfa_amd.synthetic import look_at_camera, ring_cameras  # noqa: E402

RefVFA, ref_make_grid = mg.RefVFA, mg.ref_make_grid
torch.set_num_threads(4)


def _backward(vfa, feature, calib, grid, probe, dtype):
    """One forward + backward of the reference module in `dtype`; returns ortho, grads and the visibility mask it used."""
    vfa = vfa.to(dtype)
    f = feature.to(dtype).clone().requires_grad_(True)
    cap = {}
    orig = mg.ref_op.torch.logical_and

    def land(a, b, *r, **k):
        out = orig(a, b, *r, **k)
        cap["visible"] = out.detach().clone()
        return out

    mg.ref_op.torch.logical_and = land
    try:
        vfa.zero_grad()
        ortho = vfa(f, calib.to(dtype), grid.to(dtype))
        (ortho * probe.to(dtype)).sum().backward()
    finally:
        mg.ref_op.torch.logical_and = orig
    return (ortho.detach().contiguous(), f.grad.detach(), vfa.collapse.weight.grad.detach().clone(), vfa.collapse.bias.grad.detach().clone(),
            cap["visible"])


def module_case(fname, data, image_size, cube_size, grid_height, grid, calib, C, feat_hw, seed, signed=False, wscale=1.0):
    torch.manual_seed(seed)
    args = types.SimpleNamespace(data=data, image_size=tuple(image_size))
    vfa = RefVFA(channel=C, grid_height=grid_height, cube_size=cube_size, args=args)
    with torch.no_grad():  # (larger weights, a bias on both sides of zero: the ReLU cuts a real share of the outputs)
        vfa.collapse.weight.mul_(wscale)
        vfa.collapse.bias.uniform_(-0.2, 0.1)
    feat = torch.randn(1, C, *feat_hw)
    if not signed:
        feat = torch.relu(feat)
    w32, b32 = vfa.collapse.weight.detach().clone(), vfa.collapse.bias.detach().clone()
    L, W = grid.shape[1:3]
    probe = torch.randn(1, C, L, W)
    o32, gf32, gw32, gb32, vis32 = _backward(vfa, feat, calib, grid, probe, torch.float32)
    o64, gf64, gw64, gb64, vis64 = _backward(vfa, feat, calib, grid, probe, torch.float64)
    assert torch.equal(vis32, vis64), f"{fname}: fp32 and float64 disagree on the visibility of a box; pick another seed"
    active = float((o32 > 0).float().mean())
    assert active > 0.05, (fname, active)
    noise = {k: float((a.double() - b).abs().max() / b.abs().max()) for k, a, b in
             (("feature", gf32, gf64), ("weight", gw32, gw64), ("bias", gb32, gb64), ("ortho", o32, o64))}
    np.savez_compressed(os.path.join(HERE, fname), data=data, image_size=np.array(image_size), cube_size=np.array(cube_size, dtype=np.float64),
                        grid_height=np.array(grid_height), seed=np.array(seed), feature=feat[0].numpy(), calib=calib.numpy(),
                        grid=grid[0].numpy(), weight=w32.numpy(), bias=b32.numpy(), probe=probe[0].numpy(), ortho=o32[0].numpy(),
                        d_feature=gf32[0].numpy(), d_weight=gw32.numpy(), d_bias=gb32.numpy(),
                        # (the float64 run, stored rounded to fp32: it is a yardstick for noise of 1e-6 and up, not a second reference)
                        d_feature64=gf64[0].float().numpy(), d_weight64=gw64.float().numpy(), d_bias64=gb64.float().numpy(),
                        ortho64=o64[0].float().numpy())
    print(f"{fname}: C {C} nl {vfa.collapse.in_features // C} grid {L}x{W}, {active:.0%} outputs active, visible {float(vis32.float().mean()):.2f}; "
          f"fp32 vs float64 of the reference (max / max): " + ", ".join(f"{k} {v:.2e}" for k, v in noise.items()))


def frame_case(fname, data, image_size, cube_size, grid_height, grid, calibs, feat_hws, seed, wscale=3.0):
    """vfanet.py:64-82 on given lateral maps: ortho = sum_cam (vfa8(lat8) + vfa16(lat16) + vfa32(lat32))."""
    C = 256
    torch.manual_seed(seed)
    args = types.SimpleNamespace(data=data, image_size=tuple(image_size))
    mods = [RefVFA(channel=C, grid_height=grid_height, cube_size=cube_size, args=args) for _ in range(3)]
    with torch.no_grad():
        for m in mods:
            m.collapse.weight.mul_(wscale)
            m.collapse.bias.uniform_(-0.3, 0.1)
    n = calibs.shape[0]
    lats = [torch.relu(torch.randn(n, C, *hw)) for hw in feat_hws]
    L, W = grid.shape[1:3]
    probe = torch.randn(1, C, L, W)

    def run(dtype):
        ms = [m.to(dtype) for m in mods]
        ls = [l.to(dtype).clone().requires_grad_(True) for l in lats]
        for m in ms:
            m.zero_grad()
        ortho = 0
        for cam in range(n):                                                       # vfanet.py:65
            f = [ms[s](ls[s][[cam], ...], calibs[cam].to(dtype), grid.to(dtype)) for s in range(3)]   # :76-78
            ortho = ortho + ((f[0] + f[1]) + f[2])                                 # :79, :82
        (ortho * probe.to(dtype)).sum().backward()
        return (ortho.detach().contiguous(), [l.grad.detach() for l in ls], [m.collapse.weight.grad.detach().clone() for m in ms],
                [m.collapse.bias.grad.detach().clone() for m in ms])

    w32 = [m.collapse.weight.detach().clone() for m in mods]
    b32 = [m.collapse.bias.detach().clone() for m in mods]
    o32, gl32, gw32, gb32 = run(torch.float32)
    o64, gl64, gw64, gb64 = run(torch.float64)
    out = dict(data=data, image_size=np.array(image_size), cube_size=np.array(cube_size, dtype=np.float64), grid_height=np.array(grid_height),
               seed=np.array(seed), calibs=calibs.numpy(), grid=grid[0].numpy(), probe=probe[0].numpy(), ortho=o32[0].numpy(),
               ortho64=o64[0].float().numpy())
    for s, name in enumerate((8, 16, 32)):
        out.update({f"lat{name}": lats[s].numpy(), f"weight{name}": w32[s].numpy(), f"bias{name}": b32[s].numpy(),
                    f"d_lat{name}": gl32[s].numpy(), f"d_weight{name}": gw32[s].numpy(), f"d_bias{name}": gb32[s].numpy(),
                    f"d_lat{name}_64": gl64[s].float().numpy(), f"d_weight{name}_64": gw64[s].float().numpy(),
                    f"d_bias{name}_64": gb64[s].float().numpy()})
    np.savez_compressed(os.path.join(HERE, fname), **out)
    worst = max(float((a.double() - b).abs().max() / b.abs().max()) for a, b in zip(gl32 + gw32 + gb32, gl64 + gw64 + gb64))
    print(f"{fname}: {n} cameras x 3 scales, grid {L}x{W}, {float((o32 > 0).float().mean()):.0%} outputs active; worst fp32 vs float64 gradient "
          f"of the reference {worst:.2e}")


def main():
    # ---- module level, two cases per dataset kind (small C: the fixtures stay small) -------------------------------------------
    g = ref_make_grid((3900, 3900), cube_LW=[300, 260], dataset="MultiviewC").unsqueeze(0)  # (1,13,15,3)
    cams = ring_cameras(3, (1950.0, 1950.0, 0.0), 2800.0, 600.0, 900.0, (1280, 720))
    module_case("grad_mc_s16.npz", "MultiviewC", (720, 1280), (25, 25, 32), 160, g, cams[1], 8, (45, 80), 51, wscale=3.0)
    inside = torch.tensor(look_at_camera((1500.0, 1700.0, 250.0), (2600.0, 2300.0, 0.0), 700.0, (1280, 720)), dtype=torch.float32)
    g2 = ref_make_grid((3900, 3900), cube_LW=[150, 175], dataset="MultiviewC").unsqueeze(0)
    module_case("grad_mc_inside_signed.npz", "MultiviewC", (720, 1280), (25, 25, 32), 160, g2, inside, 4, (45, 80), 52, signed=True, wscale=3.0)
    gw = ref_make_grid((480, 1440), cube_LW=[32, 60], dataset="Wildtrack").unsqueeze(0)
    wt_c = (480 * 2.5 / 2 - 300.0, 1440 * 2.5 / 2 - 900.0, 0.0)
    wcams = ring_cameras(3, wt_c, 0.45 * 1440 * 2.5, 400.0, 1100.0, (1920, 1080))
    module_case("grad_wt_s8.npz", "Wildtrack", (1080, 1920), (4, 4, 4), 32, gw, wcams[0], 4, (45, 80), 53, wscale=3.0)
    module_case("grad_wt_s32.npz", "Wildtrack", (1080, 1920), (4, 4, 4), 32, gw, wcams[1], 8, (23, 40), 54, wscale=3.0)
    gx = ref_make_grid((640, 1000), cube_LW=[50, 40], dataset="MultiviewX").unsqueeze(0)
    mx_cam = torch.tensor(look_at_camera((-5.0, 8.0, 3.0), (12.0, 8.0, 0.0), 1700.0, (1920, 1080)), dtype=torch.float32)
    module_case("grad_mx_s16.npz", "MultiviewX", (1080, 1920), (4, 4, 8), 64, gx, mx_cam, 8, (45, 80), 55, wscale=3.0)
    mx_cam2 = torch.tensor(look_at_camera((30.0, 20.0, 2.5), (12.0, 6.0, 0.0), 1400.0, (1920, 1080)), dtype=torch.float32)
    module_case("grad_mx_s32.npz", "MultiviewX", (1080, 1920), (4, 4, 8), 64, gx, mx_cam2, 8, (23, 40), 56, wscale=3.0)
    # ---- module level at C = 256: the MFMA products and the LDS-privatised scatter of the HIP path ------------------------------
    g3 = ref_make_grid((3750, 3750), cube_LW=[250, 375], dataset="MultiviewC").unsqueeze(0)  # (1,15,10,3)
    module_case("grad_mc_c256_nl1.npz", "MultiviewC", (720, 1280), (18.75, 18.75, 160), 160, g3, cams[1], 256, (12, 20), 57, wscale=3.0)
    # ---- frame level: the camera loop on 256-channel laterals, single-layer grid -----------------------------------------------
    gf = ref_make_grid((3750, 3750), cube_LW=[300, 250], dataset="MultiviewC").unsqueeze(0)
    fcams = ring_cameras(2, (1875.0, 1875.0, 0.0), 2700.0, 600.0, 900.0, (1280, 720))
    frame_case("grad_frame_mc_nl1.npz", "MultiviewC", (720, 1280), (18.75, 18.75, 160), 160, gf, fcams, [(12, 20), (6, 10), (3, 5)], 58)


if __name__ == "__main__":
    main()

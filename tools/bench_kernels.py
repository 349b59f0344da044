#!/usr/bin/env python3
"""Per-kernel timing of the HIP entry points on one workload (HIP events on the launch stream, interleaved rounds).

    python tools/bench_kernels.py [--workload multiviewc_200x200x1] [--rounds 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vfa_amd import _lib, ops  # noqa: E402
import vfa_amd  # noqa: E402
from vfa_amd.synthetic import make_workload  # noqa: E402


def timed(fn, rounds):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(rounds)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--workload", default="multiviewc_200x200x1")
    p.add_argument("--rounds", type=int, default=20)
    p.add_argument("--channels", type=int, default=256)
    a = p.parse_args()
    dev = torch.device("cuda:0")
    wl = make_workload(a.workload, channels=a.channels, seed=0)
    n = wl["n_cam"]
    mod = vfa_amd.VFA(a.channels, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
    zl, co = mod._kernel_geometry(dev)
    nl = zl.numel()
    grid_flat = wl["grid"].reshape(-1, 3).to(dev).contiguous()
    calibs = wl["calibs"].reshape(n, 12).to(dev).contiguous()
    kind = _lib.CONV_KIND[wl["args"].data]
    img_wh = wl["args"].image_size[::-1]
    cells = grid_flat.shape[0]
    C = a.channels
    print(f"workload {a.workload}: {n} cameras, grid {cells} cells x {nl} layers, C={C}")
    for s in range(3):
        lat = torch.cat([wl["features"][c][s] for c in range(n)]).to(dev)
        Hf, Wf = lat.shape[-2:]
        integral = ops.integral_image(lat)
        vox = torch.empty((n, cells, nl * C), device=dev)
        box, area, vis = ops.box_params(calibs, grid_flat, zl, co, kind, img_wh, (Hf, Wf))
        visfrac = vis.float().mean().item()
        for _ in range(3):
            ops.project_gather(integral, calibs, grid_flat, zl, co, kind, img_wh, out=vox)
        t_int = timed(lambda: ops.integral_image(lat), a.rounds)
        ops.project_gather(integral, calibs, grid_flat, zl, co, kind, img_wh, out=vox, kernel="direct")
        t_fused = timed(lambda: ops.project_gather(integral, calibs, grid_flat, zl, co, kind, img_wh, out=vox,
                                                   kernel="direct"), a.rounds)
        voxc = ops.project_gather(integral, calibs, grid_flat, zl, co, kind, img_wh, kernel="tap_cache")
        same_c = torch.equal(voxc.view(torch.int32), vox.view(torch.int32))
        t_cache = timed(lambda: ops.project_gather(integral, calibs, grid_flat, zl, co, kind, img_wh, out=voxc,
                                                   kernel="tap_cache"), a.rounds)
        del voxc
        ws = torch.empty(_lib.lib().vfa_gather_workspace_bytes(n, nl, cells), dtype=torch.uint8, device=dev)
        vox2 = ops.project_gather_ws(integral, calibs, grid_flat, zl, co, kind, img_wh, workspace=ws)
        same = torch.equal(vox2.view(torch.int32), vox.view(torch.int32))
        t_ws = timed(lambda: ops.project_gather_ws(integral, calibs, grid_flat, zl, co, kind, img_wh, out=vox2,
                                                   workspace=ws), a.rounds)
        del vox2
        w_lm = mod.layer_major_weight().detach()
        w_t = w_lm.t().contiguous()
        lin = ops.project_collapse(integral, calibs, grid_flat, zl, co, w_t, kind, img_wh)
        t_fc = timed(lambda: ops.project_collapse(integral, calibs, grid_flat, zl, co, w_t, kind, img_wh, out=lin), a.rounds)
        t_mm = timed(lambda: torch.matmul(vox.view(-1, nl * C), w_lm.t()), a.rounds)
        gflop = 2.0 * n * cells * nl * C * C / 1e9
        t_unf = timed(lambda: ops.gather(integral, box, area, vis), a.rounds)
        t_box = timed(lambda: ops.box_params(calibs, grid_flat, zl, co, kind, img_wh, (Hf, Wf)), a.rounds)
        nbox = n * cells * nl
        bytes_g = n * C * Hf * Wf * 4 + nbox * C * 4 + cells * 12
        bytes_i = 2 * n * C * Hf * Wf * 4
        print(f" scale {Hf}x{Wf}: visible {visfrac:.2f} | integral {t_int[0]:.1f} us ({bytes_i / t_int[0] / 1e3:.0f} GB/s alg) | "
              f"project_gather med {t_fused[0]:.1f} min {t_fused[1]:.1f} us = {nbox / t_fused[0] / 1e3:.2f} Gbox/s, "
              f"{bytes_g / t_fused[0] / 1e3:.0f} GB/s alg | TAP-CACHE med {t_cache[0]:.1f} min {t_cache[1]:.1f} us = {bytes_g / t_cache[0] / 1e3:.0f} GB/s alg bitwise_same={same_c} | ws-form med {t_ws[0]:.1f} min {t_ws[1]:.1f} us bitwise_same={same} | FUSED collapse med {t_fc[0]:.1f} us ({gflop / t_fc[0] * 1e3:.1f} TF) vs gather+GEMM {t_fused[0] + t_mm[0]:.1f} us (GEMM {t_mm[0]:.1f} us, {gflop / t_mm[0] * 1e3:.1f} TF) | gather(unfused) {t_unf[0]:.1f} us | box_params {t_box[0]:.1f} us")


if __name__ == "__main__":
    main()

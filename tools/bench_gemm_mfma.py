#!/usr/bin/env python3
"""Time `vfa_collapse_gemm_f32` against the fp32 library GEMM on the multi-layer shapes of the BASELINE configs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vfa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, M, K in (("multiviewc 156x156x5", 7 * 24336, 1280), ("wildtrack 120x360x8", 7 * 43200, 2048),
                   ("multiviewc 200x200x1", 7 * 40000, 256)):
    torch.manual_seed(0)
    vox = torch.rand(M, K, device=dev) * (torch.rand(M, 1, device=dev) > 0.3)
    w = (torch.rand(256, K, device=dev) - 0.5) * (2.0 / K ** 0.5)
    ref = vox[:20000].double() @ w.double().T
    flops = 2.0 * M * K * 256
    for label, f in (("library fp32", lambda: vox @ w.T), ("mfma bf16x3", lambda: ops.collapse_gemm(vox, w, terms=3))):
        out = f()
        err = (out[:20000].double() - ref).abs().max().item() / ref.abs().max().item()
        us = timeit(f)
        print(f"{name} M={M} K={K}: {label} {us:.0f} us, {flops / us / 1e6:.0f} TFLOP/s fp32-equivalent, {vox.numel() * 4 / us / 1e6:.2f} TB/s of vox, "
              f"max err / max|out| {err:.2e}")

#!/usr/bin/env python3
"""Time `vfa_collapse_relu_sum_f32` against library GEMM + epilogue on the bench workload's shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vfa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, M = 7, int(sys.argv[1]) if len(sys.argv) > 1 else 40000
torch.manual_seed(0)
vox = torch.rand(n, M, 256, device=dev)
if os.environ.get("MASK", "coherent") == "coherent":   # each view misses a contiguous third of the cells (like real visibility)
    for v in range(n):
        a0 = (v * M) // n
        idx = (torch.arange(M // 3, device=dev) + a0) % M
        vox[v, idx] = 0
else:
    vox *= (torch.rand(n, M, 1, device=dev) > 0.3)
w = (torch.rand(256, 256, device=dev) - 0.5) * 0.125
b = (torch.rand(256, device=dev) - 0.5) * 0.125
ref = torch.relu(vox.double() @ w.double().T + b.double()).sum(0)


def timeit(f, reps=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def library():
    lin = (vox.view(-1, 256) @ w.T).view(n, M, 256)
    return ops.bias_relu_accumulate(lin, b)


acc0 = torch.zeros(M, 256, device=dev)
for name, f in (("library fp32 GEMM + epilogue", library), ("collapse_relu_sum terms=3", lambda: ops.collapse_relu_sum(vox, w, b, terms=3)),
                ("collapse_relu_sum terms=4", lambda: ops.collapse_relu_sum(vox, w, b, terms=4)),
                ("collapse_relu_sum terms=3 accumulate", lambda: ops.collapse_relu_sum(vox, w, b, out=acc0.zero_(), accumulate=True, terms=3))):
    out = f()
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    us = timeit(f)
    print(f"{name}: {us:.0f} us  ({vox.numel() * 4 / us / 1e6:.2f} TB/s of vox, {2 * n * M * 65536 / us / 1e6:.0f} TFLOP/s fp32-equivalent), "
          f"max err / max|out| {err:.2e}")

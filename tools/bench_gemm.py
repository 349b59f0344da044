#!/usr/bin/env python3
"""fp32 collapse GEMM variants through torch (rocBLAS / hipBLASLt), with and without TunableOp."""
import sys
import time

import torch

tune = len(sys.argv) > 1 and sys.argv[1] == "tune"
if tune:
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(200)
    torch.cuda.tunable.set_filename("/tmp/bench_gemm_tune.csv")
dev = torch.device("cuda:0")
M, K, N = 7 * 40000, 256, 256
a = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev)
wt = w.t().contiguous()
out = torch.empty(M, N, device=dev)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return ts[len(ts) // 2]


fl = 2.0 * M * K * N
for name, fn in [("a @ w.t() (TN)", lambda: torch.matmul(a, w.t())), ("a @ wt (NN)", lambda: torch.matmul(a, wt)),
                 ("mm(out=) NN", lambda: torch.mm(a, wt, out=out)), ("F.linear", lambda: torch.nn.functional.linear(a, w))]:
    us = t(fn)
    print(f"{'tuned' if tune else 'default'} {name:18s}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")

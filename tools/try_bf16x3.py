#!/usr/bin/env python3
"""Experiment: error-compensated bf16 GEMM (3 or 6 products) vs the fp32 library GEMM for `collapse`."""
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, K, N = 280000, 256, 256
a = torch.rand(M, K, device=dev) * (torch.rand(M, 1, device=dev) > 0.3)   # non-negative vox features, 30 % masked rows
w = (torch.rand(N, K, device=dev) - 0.5) * (2.0 / K ** 0.5)               # nn.Linear default init range
ref = (a.double() @ w.double().T)
def split(x, n):
    out = []
    for _ in range(n):
        h = x.to(torch.bfloat16)
        out.append(h)
        x = x - h.float()
    return out
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def report(name, out, us):
    err = (out.double() - ref).abs()
    tol = 1e-4 * ref.abs() + 1e-5 * ref.abs().max()
    print(f"{name}: {us:.0f} us, max err {err.max().item():.3e} (max|ref| {ref.abs().max().item():.3f}), worst err/tol {(err / tol).max().item():.3f}")
f32 = lambda: a @ w.T
report("fp32 library", f32(), timeit(f32))
for terms, pairs in ((3, [(0, 0), (0, 1), (1, 0)]), (4, [(0, 0), (0, 1), (1, 0), (1, 1)]), (6, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)])):
    n = max(max(p) for p in pairs) + 1
    as_, ws_ = split(a, n), split(w, n)
    A = torch.cat([as_[i] for i, _ in pairs], 1).contiguous()
    W = torch.cat([ws_[j] for _, j in pairs], 1).contiguous()
    try:
        g = lambda: torch.mm(A, W.T, out_dtype=torch.float32)
        report(f"bf16x{terms} one GEMM K={A.shape[1]} (out_dtype fp32)", g(), timeit(g))
    except Exception as e:
        print("out_dtype path failed:", type(e).__name__, str(e)[:200])
        # accuracy only: fp32 matmuls of the bf16-rounded parts
        out = sum(as_[i].float() @ ws_[j].float().T for i, j in pairs)
        report(f"bf16x{terms} (emulated in fp32, accuracy only)", out, float("nan"))
bf = lambda: torch.mm(a.to(torch.bfloat16), w.to(torch.bfloat16).T)
print(f"plain bf16 GEMM incl. casts: {timeit(bf):.0f} us")
ab, wb = a.to(torch.bfloat16), w.to(torch.bfloat16)
bf2 = lambda: torch.mm(ab, wb.T)
print(f"plain bf16 GEMM K=256, bf16 out: {timeit(bf2):.0f} us")

import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfa_amd import ops
dev = torch.device('cuda:0')
shapes = [(90,160),(45,80),(23,40)]
feats = [torch.relu(torch.randn(7,256,h,w,device=dev)) for h,w in shapes]
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/n*1e3
print("batched 3 maps: %.1f us" % timeit(lambda: ops.integral_images(feats)))
for f in feats:
    print(tuple(f.shape), "per-map old: %.1f us; batched-1: %.1f us" % (timeit(lambda: ops.integral_image(f)), timeit(lambda: ops.integral_images([f]))))
x = torch.empty(272*1024*1024//4, device=dev); y = torch.empty_like(x)
print("copy 272MB r + 272MB w: %.1f us" % timeit(lambda: y.copy_(x)))

import os, sys
sys.path.insert(0, "/root/repo")
import torch
from vfa_amd import ops
dev = torch.device("cuda:0")
M, K = 7 * 24336, 1280
vox = torch.rand(M, K, device=dev); w = torch.rand(256, K, device=dev) - 0.5
for _ in range(6): ops.collapse_gemm(vox, w, terms=3)
torch.cuda.synchronize()

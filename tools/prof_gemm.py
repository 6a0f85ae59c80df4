import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from versecrafter_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 65520, 5120, 5120
a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
bias = torch.randn(N, device="cuda", generator=g).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(a, w, bias, out=out, tile=2)
torch.cuda.synchronize()

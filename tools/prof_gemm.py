import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from versecrafter_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
M, MP, N, K = 65520, 65536, 5120, 5120
a = torch.randn(MP, K, device="cuda", generator=g).bfloat16()[:M]        # rows up to the next multiple of 256 are readable
w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
bias = torch.randn(N, device="cuda", generator=g).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(a, w, bias, out=out, tile=4)                                  # 4: the ping-pong kernel (gemm_pp_kernel)
torch.cuda.synchronize()

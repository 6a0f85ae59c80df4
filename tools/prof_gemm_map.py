"""Micro-driver for counter passes: the ping-pong GEMM at one cfg-3 shape under a tile map.  argv: tile id (4 production map, 6 shared super-band), N, K."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from versecrafter_amd import ops
tile, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator(device="cuda").manual_seed(0)
M, MP = 65520, 65536
a = torch.randn(MP, K, device="cuda", generator=g).bfloat16()[:M]
w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
bias = torch.randn(N, device="cuda", generator=g).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(a, w, bias, out=out, tile=tile)
torch.cuda.synchronize()

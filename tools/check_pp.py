#!/usr/bin/env python3
"""Where does the ping-pong GEMM (tile 4) differ from the 2-stage kernel (tile 2)?  (GPU box only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from versecrafter_amd import ops
TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in ((1024, 256, 128), (1024, 256, 256), (1100, 512, 1024), (2048, 768, 384), (65520, 5120, 5120), (65520, 5120, 13824)):
    MP = (M + 255) // 256 * 256
    abuf = torch.randn(MP, K, device="cuda", generator=g).bfloat16()
    a = abuf[:M]
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    bias = torch.zeros(N, device="cuda").bfloat16()
    o2 = ops.gemm(a, w, bias, tile=2)
    o4 = ops.gemm(a, w, bias, tile=TILE)
    o4b = ops.gemm(a, w, bias, tile=TILE)
    torch.cuda.synchronize()
    ne = (o2 != o4)
    print(f"M={M} N={N} K={K}: mismatches {int(ne.sum())} / {ne.numel()}  rerun-identical {torch.equal(o4, o4b)} "
          f"max|d| {float((o2.float() - o4.float()).abs().max()):.4g}", flush=True)
    if ne.any():
        idx = ne.nonzero()
        rows, cols = idx[:, 0], idx[:, 1]
        print("   rows%256 hist(16-bins):", torch.bincount((rows % 256) // 16, minlength=16).tolist())
        print("   cols%256 hist(16-bins):", torch.bincount((cols % 256) // 16, minlength=16).tolist())
        if M <= 512:
            ref = (a.float() @ w.float().t())
            print("   err vs fp32: tile2 %.4g  tile4 %.4g" % (float((o2.float() - ref).abs().max()), float((o4.float() - ref).abs().max())))

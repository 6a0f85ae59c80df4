#!/usr/bin/env python3
"""Bitwise pre-flight of a GEMM kernel variant against tile 2 at awkward shapes (GPU box only, 3 repeats) before a timing A/B:
   python tools/ab_gemm_small.py 5"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from versecrafter_amd import ops

t = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator(device="cuda").manual_seed(0)
ok = True
for (M, N, K, epi) in ((1024, 256, 128, 0), (1024, 256, 256, 0), (1500, 512, 384, 1), (4096, 2048, 512, 0),
                       (20000, 1536, 1536, 3), (33000, 5120, 384, 0), (66000, 2560, 640, 1)):
    MP = (M + 255) // 256 * 256
    abuf = torch.randn(MP, K, device="cuda", generator=g).bfloat16()
    a = abuf[:M]
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g).bfloat16()
    resid = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi == 3 else None
    gate = torch.randn(2, N, device="cuda", generator=g).bfloat16() if epi == 3 else None
    ref = ops.gemm(a, w, bias, epilogue=epi, resid=resid, gate=gate, rows_per_batch=M // 2, tile=2)
    for rep in range(3):
        out = ops.gemm(a, w, bias, epilogue=epi, resid=resid, gate=gate, rows_per_batch=M // 2, tile=t)
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        ok = ok and same
        print(f"M={M} N={N} K={K} epi={epi} rep {rep}: tile {t} == tile 2 bitwise: {same}"
              + ("" if same else f"  max diff {(out.float() - ref.float()).abs().max().item():.4g}"), flush=True)
sys.exit(0 if ok else 1)

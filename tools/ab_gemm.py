#!/usr/bin/env python3
"""A/B of GEMM kernel variants at the cfg-3 shapes (GPU box only): bitwise comparison + interleaved timings.
   python tools/ab_gemm.py 2 4     (tile ids of vc_gemm_tile_override)"""
import os
import sys
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from versecrafter_amd import ops

tiles = [int(t) for t in sys.argv[1:]] or [2, 4]
g = torch.Generator(device="cuda").manual_seed(0)
M, MP = 65520, 65536
for (N, K, epi) in ((5120, 5120, 0), (13824, 5120, 1), (5120, 13824, 3)):
    abuf = torch.randn(MP, K, device="cuda", generator=g).bfloat16()
    a = abuf[:M]
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g).bfloat16()
    resid = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi == 3 else None
    gate = torch.randn(2, N, device="cuda", generator=g).bfloat16() if epi == 3 else None
    outs = {}
    for t in tiles:
        outs[t] = ops.gemm(a, w, bias, epilogue=epi, resid=resid, gate=gate, rows_per_batch=M // 2, tile=t)
    torch.cuda.synchronize()
    for t in tiles[1:]:
        same = torch.equal(outs[t], outs[tiles[0]])
        print(f"N={N} K={K} epi={epi}: tile {t} == tile {tiles[0]} bitwise: {same}", flush=True)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    times = {t: [] for t in tiles}
    for _ in range(5):
        for t in tiles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                ops.gemm(a, w, bias, epilogue=epi, resid=resid, gate=gate, rows_per_batch=M // 2, out=out, tile=t)
            e1.record()
            torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) / 2)
    fl = 2.0 * M * N * K
    for t in tiles:
        med, mn = statistics.median(times[t]), min(times[t])
        print(f"  tile {t}: median {med:.3f} ms ({fl / med / 1e9:.0f} TF)  min {mn:.3f} ms ({fl / mn / 1e9:.0f} TF)", flush=True)

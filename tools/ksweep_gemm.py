#!/usr/bin/env python3
"""Per-tile overhead of a GEMM kernel variant: time against K at fixed M, N (GPU box only); a line fit gives the cost of one
K-tile (slope) and of the tile prologue + epilogue (intercept).   python tools/ksweep_gemm.py 4 5"""
import os
import sys
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from versecrafter_amd import ops

tiles = [int(t) for t in sys.argv[1:]] or [4, 5]
M, N = 65536, 5120
g = torch.Generator(device="cuda").manual_seed(0)
ks = tuple(int(k) for k in os.environ.get("KS", "1024,2048,4096,8192,16384").split(","))
res = {t: [] for t in tiles}
for K in ks:
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for t in tiles:
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm(a, w, None, out=out, tile=t)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
        res[t].append(statistics.median(ts))
    del a, w
per_cu = (M // 256) * (N // 256) / 256          # tiles each CU works through
for t in tiles:
    xs = [k / 64 for k in ks]
    ys = [ms * 1e3 / per_cu for ms in res[t]]   # us per tile
    n = len(xs)
    mx, my = sum(xs) / n, sum(ys) / n
    slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    icpt = my - slope * mx
    print(f"tile {t}: " + "  ".join(f"K={k}: {ms:.3f} ms ({2.0 * M * N * k / ms / 1e9:.0f} TF)" for k, ms in zip(ks, res[t])))
    print(f"   per K-tile {slope:.3f} us, per output tile {icpt:.2f} us", flush=True)

"""Micro-driver for counter passes and timing: the FP8 instantiation of the ping-pong GEMM at one cfg-3 shape.  argv: N K [ksweep]
   ksweep: instead of one shape, times K = 1024 .. K in steps and fits  time = tiles x (K-tiles x a + b)  (a: us per 128-byte K-tile, b: us per
   256 x 256 output tile) -- the same fit tools/ksweep_gemm.py makes for the bf16 kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from versecrafter_amd import ops
N, K = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(0)
M, MP = 65520, 65536


def operands(k):
    a = torch.randn(MP, k, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, k, device="cuda", generator=g) * k ** -0.5).bfloat16()
    aq, asc = ops.quantize_rows_fp8(a)
    wq, wsc = ops.quantize_rows_fp8(w)
    return aq[:M], asc[:M], wq, wsc


bias = torch.randn(N, device="cuda", generator=g).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
if len(sys.argv) > 3 and sys.argv[3] == "ksweep":
    import numpy as np
    ks, ts = [], []
    for k in range(1024, K + 1, 1024 if K <= 8192 else 2560):
        aq, asc, wq, wsc = operands(k)
        for _ in range(2):
            ops.gemm_fp8(aq, asc, wq, wsc, bias, out=out, a_rows_padded=True)
        torch.cuda.synchronize()
        best = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm_fp8(aq, asc, wq, wsc, bias, out=out, a_rows_padded=True)
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) / 3)
        best.sort()
        ks.append(k); ts.append(best[2])
        print(f"  K = {k:6d}: {best[2]:.3f} ms  {2.0 * M * N * k / best[2] / 1e9:.0f} TFLOP/s")
    tiles_per_cu = (MP // 256) * (N // 256) / 256.0
    A = np.stack([np.array(ks) / 128.0, np.ones(len(ks))], 1) * tiles_per_cu
    (a_, b_), *_ = np.linalg.lstsq(A, np.array(ts) * 1e3, rcond=None)
    print(f"fp8 ping-pong GEMM, N = {N}: a = {a_:.3f} us per 128-byte K-tile (in-loop {2.0 * 256 * 256 * 128 * 256 / a_ / 1e6:.0f} TFLOP/s), b = {b_:.2f} us per output tile")
else:
    aq, asc, wq, wsc = operands(K)
    for _ in range(3):
        ops.gemm_fp8(aq, asc, wq, wsc, bias, out=out, a_rows_padded=True)
    torch.cuda.synchronize()

#!/usr/bin/env python3
"""fp8 GEMM (vc_op_gemm_fp8) against the dequantised fp32 product of the same e4m3 operands, and its speed next to the bf16 kernel.
   python tools/fp8_gemm_check.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from versecrafter_amd import ops


def main():
    torch.manual_seed(0)
    dev = "cuda"
    for (M, N, K) in [(256, 256, 256), (512, 768, 1024), (1024, 5120, 5120)]:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        bias = torch.randn(N, device=dev).bfloat16()
        aq, asc = ops.quantize_rows_fp8(a)
        wq, wsc = ops.quantize_rows_fp8(w)
        # quantiser against torch's own e4m3 cast
        ref_sc = a.float().abs().amax(1) / 448.0
        ref_q = (a.float() * (1.0 / ref_sc)[:, None]).to(torch.float8_e4m3fn)
        same = (ref_q.view(torch.uint8) == aq).float().mean().item()
        print(f"[{M}x{N}x{K}] quantiser: scale max rel diff {((asc - ref_sc).abs() / ref_sc).max().item():.2e}, bytes equal to torch's cast {same:.6f}")
        deq_a = aq.view(torch.float8_e4m3fn).float() * asc[:, None]
        deq_w = wq.view(torch.float8_e4m3fn).float() * wsc[:, None]
        want = deq_a @ deq_w.T + bias.float()
        got = ops.gemm_fp8(aq, asc, wq, wsc, bias=bias).float()
        torch.cuda.synchronize()
        e = ((got - want).norm() / want.norm()).item()
        full = a.float() @ w.float().T + bias.float()
        eq = ((want - full).norm() / full.norm()).item()
        print(f"    kernel vs dequantised fp32 product: rel L2 {e:.3e} (bf16 output rounding ~2e-3);  quantisation itself vs bf16 operands: {eq:.3e}")
    # speed at the engine's shapes
    for (M, N, K, name) in [(65536, 5120, 5120, "5120^2"), (65536, 13824, 5120, "5120->13824"), (65536, 5120, 13824, "13824->5120")]:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        bias = torch.randn(N, device=dev).bfloat16()
        aq, asc = ops.quantize_rows_fp8(a)
        wq, wsc = ops.quantize_rows_fp8(w)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def timeit(fn, n=10):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.time() - t0) / n
        t8 = timeit(lambda: ops.gemm_fp8(aq, asc, wq, wsc, bias=bias, out=out))
        t16 = timeit(lambda: ops.gemm(a, w, bias=bias, out=out))
        tq = timeit(lambda: ops.quantize_rows_fp8(a))
        fl = 2.0 * M * N * K
        print(f"{name}: fp8 {t8 * 1e3:.3f} ms ({fl / t8 / 1e12:.0f} TF)   bf16 {t16 * 1e3:.3f} ms ({fl / t16 / 1e12:.0f} TF)   "
              f"row quantiser of A {tq * 1e3:.3f} ms ({M * K * 3 / tq / 1e9:.0f} GB/s)")


if __name__ == "__main__":
    main()

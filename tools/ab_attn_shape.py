#!/usr/bin/env python3
"""MFMA-shape A/B of the self-attention kernel at the cfg-3 shape (B=2, 40 heads, 32760 tokens), random gaussian data:
v_mfma_f32_32x32x16_bf16 (attention.hip) against v_mfma_f32_16x16x32_bf16 (attention16.hip), same output tile per wave.
Interleaved rounds in ONE process on ONE device (cdna_hip_programming.md rule 24); median and min of HIP-event times.
   python tools/ab_attn_shape.py [rounds]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from versecrafter_amd import ops


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    g = torch.Generator(device="cuda").manual_seed(0)
    B, H, L, d = 2, 40, 32760, 5120
    qkv = torch.randn(B, L, 3 * d, device="cuda", generator=g).bfloat16()
    q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 128)) for i in range(3))
    out = {s: torch.empty(B, L, H, 128, device="cuda", dtype=torch.bfloat16) for s in (32, 16)}
    fl = 4.0 * B * H * L * L * 128
    for s in (32, 16):                                   # warm-up (also sets the kernels' LDS attribute)
        ops.attention(q, k, v, k_len=L, out=out[s], variant=s)
    torch.cuda.synchronize()
    d32, d16 = out[32].float(), out[16].float()
    print(f"16 vs 32: rel L2 {((d32 - d16).norm() / d32.norm()).item():.3e}, max abs {(d32 - d16).abs().max().item():.3e}", flush=True)
    ts = {32: [], 16: []}
    for r in range(rounds):
        for s in ((32, 16) if r % 2 == 0 else (16, 32)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.attention(q, k, v, k_len=L, out=out[s], variant=s)
            b.record()
            torch.cuda.synchronize()
            ts[s].append(a.elapsed_time(b))
    for s in (32, 16):
        med, mn = statistics.median(ts[s]), min(ts[s])
        print(f"shape {s}: median {med:.3f} ms ({fl / med / 1e9:.0f} TF) min {mn:.3f} ms ({fl / mn / 1e9:.0f} TF)  "
              f"all {[round(x, 2) for x in ts[s]]}", flush=True)
    print(f"ratio 16/32 (median time): {statistics.median(ts[16]) / statistics.median(ts[32]):.4f}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels in isolation at the cfg-3 shapes (GPU box only).
   python tools/bench_kernels.py [gemm] [attn] [row]
Interleaved rounds in ONE process, median and min of HIP-event times, random (gaussian) operands."""
import os
import sys
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from versecrafter_amd import ops


def timeit(fn, rounds=5, inner=2):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    return statistics.median(ts), min(ts)


def bench_gemm(tiles=(2,)):
    g = torch.Generator(device="cuda").manual_seed(0)
    M = 65520
    for (N, K, epi) in ((5120, 5120, 0), (13824, 5120, 1), (5120, 13824, 3)):
        a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
        bias = torch.randn(N, device="cuda", generator=g).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        resid = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi == 3 else None
        gate = torch.randn(2, N, device="cuda", generator=g).bfloat16() if epi == 3 else None
        for tile in tiles:
            med, mn = timeit(lambda: ops.gemm(a, w, bias, epilogue=epi, resid=resid, gate=gate,
                                              rows_per_batch=M // 2, out=out, tile=tile))
            fl = 2.0 * M * N * K
            print(f"gemm M={M} N={N} K={K} epi={epi} tile={tile}: median {med:.3f} ms ({fl / med / 1e9:.0f} TF) "
                  f"min {mn:.3f} ms ({fl / mn / 1e9:.0f} TF)", flush=True)


def bench_attn():
    g = torch.Generator(device="cuda").manual_seed(0)
    B, H, L, d = 2, 40, 32760, 5120
    qkv = torch.randn(B, L, 3 * d, device="cuda", generator=g).bfloat16()
    q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 128)) for i in range(3))
    out = torch.empty(B, L, H, 128, device="cuda", dtype=torch.bfloat16)
    med, mn = timeit(lambda: ops.attention(q, k, v, k_len=L, out=out), rounds=4, inner=1)
    fl = 4.0 * B * H * L * L * 128
    print(f"attn self B={B} H={H} L={L}: median {med:.3f} ms ({fl / med / 1e9:.0f} TF) min {mn:.3f} ms "
          f"({fl / mn / 1e9:.0f} TF)", flush=True)
    kc = torch.randn(B, 512, H, 128, device="cuda", generator=g).bfloat16()
    vc = torch.randn(B, 512, H, 128, device="cuda", generator=g).bfloat16()
    qc = torch.randn(B, L, H, 128, device="cuda", generator=g).bfloat16()
    med, mn = timeit(lambda: ops.attention(qc, kc, vc, out=out), rounds=4, inner=2)
    fl = 4.0 * B * H * L * 512 * 128
    print(f"attn cross Lk=512: median {med:.3f} ms ({fl / med / 1e9:.0f} TF)", flush=True)
    med, mn = timeit(lambda: ops.attention_padmerge(qc, kc, vc, [60, 77]), rounds=4, inner=2)
    by = 2.0 * 2 * B * H * L * 128
    print(f"attn cross, padded keys folded (60 / 77-token prompts, the engine's call): median {med:.3f} ms min {mn:.3f} ms "
          f"({by / mn / 1e6:.0f} GB/s of Q read + O write)", flush=True)


def bench_attn_seg():
    """Self-attention of one rank of a P = 8 / P = 2 sequence-parallel run (Ulysses receive layout, heads / P)."""
    g = torch.Generator(device="cuda").manual_seed(0)
    for S, Ls, H in ((8, 4095, 5), (2, 16380, 20)):
        B = 2
        q, k, v = (torch.randn(S, B, Ls, H, 128, device="cuda", generator=g).bfloat16() for _ in range(3))
        med, mn = timeit(lambda: ops.attention_segmented(q, k, v), rounds=4, inner=2)
        fl = 4.0 * B * H * (S * Ls) ** 2 * 128
        print(f"attn segmented S={S} Ls={Ls} H={H}: median {med:.3f} ms ({fl / med / 1e9:.0f} TF) min {mn:.3f} ms "
              f"({fl / mn / 1e9:.0f} TF)", flush=True)


def bench_row():
    g = torch.Generator(device="cuda").manual_seed(0)
    M, d = 65520, 5120
    x = torch.randn(M, d, device="cuda", generator=g).bfloat16()
    mod = torch.randn(2, 6, d, device="cuda", generator=g).bfloat16()
    med, mn = timeit(lambda: ops.layernorm_modulate(x, mod[:, 1], mod[:, 0], M // 2))
    print(f"layernorm_modulate [{M},{d}]: median {med:.3f} ms ({4.0 * M * d / med / 1e6:.0f} GB/s)", flush=True)
    from oracle import wan_oracle as O
    qkv = torch.randn(M, 3 * d, device="cuda", generator=g).bfloat16()
    wq, wk = torch.ones(d, device="cuda").bfloat16(), torch.ones(d, device="cuda").bfloat16()
    tab = ops.rope_table_device(O.rope_table(128), "cuda")
    grid = (21, 30, 52)
    med, mn = timeit(lambda: ops.qkv_front(qkv, wq, wk, tab, grid, rows_per_batch=M // 2))
    print(f"qkv_front in place (q and k of [{M},{3 * d}]): median {med:.3f} ms ({8.0 * M * d / med / 1e6:.0f} GB/s)", flush=True)
    med, mn = timeit(lambda: (ops.rmsnorm_rope_(qkv[:, :d], wq, 1e-6, tab, grid, rows_per_batch=M // 2),
                              ops.rmsnorm_rope_(qkv[:, d:2 * d], wk, 1e-6, tab, grid, rows_per_batch=M // 2)))
    print(f"two rmsnorm_rope launches (same bytes): median {med:.3f} ms ({8.0 * M * d / med / 1e6:.0f} GB/s)", flush=True)
    Mp = M // 8
    med, mn = timeit(lambda: ops.qkv_front(qkv[:Mp], wq, wk, tab, grid, rows_per_batch=Mp // 2, P=8, pack=True))
    print(f"qkv_front packed, P=8 rank share [{Mp},{3 * d}]: median {med:.3f} ms ({12.0 * Mp * d / med / 1e6:.0f} GB/s)", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["gemm", "attn", "row"]
    if "gemm" in what:
        bench_gemm()
    if "attn" in what:
        bench_attn()
    if "attnseg" in what:
        bench_attn_seg()
    if "row" in what:
        bench_row()

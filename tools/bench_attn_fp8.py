"""Times the fp8 self-attention (quantiser and attention kernels separately) against the bf16 kernel at the bench shapes, same box, interleaved
rounds (cdna_hip_programming.md rule 24).  usage: python tools/bench_attn_fp8.py [L=32760] [H=40] [B=2] [rounds=5]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versecrafter_amd import ops


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 32760
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, L, 3 * H * 128, generator=g, device="cuda").bfloat16()
    d = H * 128
    q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 128)) for i in range(3))
    out = torch.empty(B, L, H, 128, device="cuda", dtype=torch.bfloat16)
    _, ws = ops.attention_fp8(q, k, v, out=out, return_workspace=True)
    flops = 4.0 * B * H * L * L * 128

    def timed(fn, n=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    variants = {
        "bf16 attn_fwd_pipe": lambda: ops.attention(q, k, v, out=out),
        "fp8 quantiser": lambda: ops.attention_fp8(q, k, v, out=out, workspace=ws, stage=1),
        "fp8 attention pmode1": lambda: ops.attention_fp8(q, k, v, out=out, workspace=ws, stage=2, pmode=1),
        "fp8 attention pmode0": lambda: ops.attention_fp8(q, k, v, out=out, workspace=ws, stage=2, pmode=0),
    }
    res = {n: [] for n in variants}
    for _ in range(rounds):
        for n, fn in variants.items():
            res[n].append(timed(fn))
    print(f"B={B} H={H} L={L}: algorithmic {flops / 1e12:.2f} TFLOP per launch; workspace {ws.numel() / 2**20:.0f} MiB")
    for n, ts in res.items():
        ts = sorted(ts)
        med = ts[len(ts) // 2]
        extra = f"  {flops / med / 1e9:.0f} TFLOP/s (median)" if "quant" not in n else f"  {(3 * B * L * d * 2 + 3 * B * L * d) / med / 1e6:.0f} GB/s"
        print(f"  {n:24s} median {med:8.3f} ms  min {ts[0]:8.3f} ms{extra}")
    ref = ops.attention(q, k, v)
    for pm in (1, 0):
        o = ops.attention_fp8(q, k, v, pmode=pm)
        print(f"  pmode {pm}: rel L2 vs the bf16 kernel {((o.float() - ref.float()).norm() / ref.float().norm()).item():.4g}")


if __name__ == "__main__":
    main()

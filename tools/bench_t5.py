#!/usr/bin/env python3
"""Time the HIP umT5-XXL text encoder (24 layers, dim 4096, 64 heads, ffn 10240; random weights) on a prompt pair padded to
512 tokens -- the once-per-video cost of PIPE.py:273 (GPU box only).   python tools/bench_t5.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from versecrafter_amd.models import WanT5EncoderModel

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = WanT5EncoderModel(param_device=dev)                       # umT5-XXL defaults (wan_civitai.yaml:14-26)
with torch.no_grad():
    for n, p in m.named_parameters():
        if p.dim() == 1:
            p.fill_(1.0)
        else:
            p.normal_(0.0, p.shape[-1] ** -0.5)
ids = torch.randint(0, 256384, (2, 512), device=dev)
mask = torch.zeros(2, 512, dtype=torch.long, device=dev)
mask[0, :60] = 1
mask[1, :77] = 1
out = m(ids, attention_mask=mask)[0]
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    out = m(ids, attention_mask=mask)[0]
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
nparam = sum(p.numel() for p in m.parameters())
L, d, f = 1024, 4096, 10240
flops = 24 * (2 * L * d * d * 4 + 2 * L * d * f * 3 + 4 * 2 * 64 * 512 * 512 * 64)
print(f"umT5-XXL encoder: {nparam / 1e9:.2f} B parameters, 2 x 512 tokens: min {min(ts) * 1e3:.1f} ms, median {sorted(ts)[2] * 1e3:.1f} ms "
      f"({flops / min(ts) / 1e12:.0f} TFLOP/s over {flops / 1e12:.1f} TFLOP), finite={bool(torch.isfinite(out.float()).all())}, "
      f"workspace {m.workspace_bytes() / 2 ** 20:.0f} MiB")

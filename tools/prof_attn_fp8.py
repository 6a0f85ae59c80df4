"""Micro-driver for counter passes: the fp8 self-attention (quantiser + attention kernel) at the cfg-3 shape.  argv[1]: pmode (default 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from versecrafter_amd import ops
pmode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator(device="cuda").manual_seed(0)
B, H, L, d = 2, 40, 32760, 5120
qkv = torch.randn(B, L, 3 * d, device="cuda", generator=g).bfloat16()
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 128)) for i in range(3))
out = torch.empty(B, L, H, 128, device="cuda", dtype=torch.bfloat16)
_, ws = ops.attention_fp8(q, k, v, out=out, pmode=pmode, return_workspace=True)
for _ in range(2):
    ops.attention_fp8(q, k, v, out=out, pmode=pmode, workspace=ws)
torch.cuda.synchronize()

"""PMC driver of the MFMA-shape A/B: two launches of each attention variant at the cfg-3 shape (tools/ab_attn_shape.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from versecrafter_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
B, H, L, d = 2, 40, 32760, 5120
qkv = torch.randn(B, L, 3 * d, device="cuda", generator=g).bfloat16()
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 128)) for i in range(3))
out = torch.empty(B, L, H, 128, device="cuda", dtype=torch.bfloat16)
for _ in range(2):
    for s in (32, 16):
        ops.attention(q, k, v, k_len=L, out=out, variant=s)
torch.cuda.synchronize()

"""Times only the fp8 attention kernel (pmode 1) at the cfg-3 shape with the library named by VC_ENGINE_LIB (ablation builds: results are wrong)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from versecrafter_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
B, H, L, d = 2, 40, 32760, 5120
qkv = torch.randn(B, L, 3 * d, device="cuda", generator=g).bfloat16()
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 128)) for i in range(3))
out = torch.empty(B, L, H, 128, device="cuda", dtype=torch.bfloat16)
_, ws = ops.attention_fp8(q, k, v, out=out, return_workspace=True)
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.attention_fp8(q, k, v, out=out, workspace=ws, stage=2, pmode=1)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 3)
ts.sort()
print(f"{os.environ.get('VC_ENGINE_LIB', 'production')}: median {ts[2]:.3f} ms  min {ts[0]:.3f} ms")

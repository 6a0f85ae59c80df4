#!/usr/bin/env python3
"""Where a wave of attn_fp8_kernel<1> spends its time (GPU box only; needs a library built with -DF8_TRACE:
   tools/build_variant.sh f8trace -DF8_TRACE && VC_ENGINE_LIB=$PWD/versecrafter_amd/libvcengine_f8trace.so python tools/trace_attn_fp8.py).
Every wave sums s_memtime deltas over its key tiles for five sections of a beat: [0] counted vmcnt + workgroup barrier, [1] LDS-DMA issue
(addresses, M0, 2-5 pieces), [2] phase 1 (rescale test, exponent, 4 QK^T MFMAs || P conversion), [3] phase 2 (row-sum + 4 PV MFMAs || row
maximum), [4] loop control between them.  Printed per wave group (waves 0-3: phase 1 then 2; waves 4-7: staggered) as clock ticks per tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from versecrafter_amd import _lib, ops
lib = _lib.load()
lib.vc_debug_set_attn_fp8_trace.argtypes = [ctypes.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
B, H, L, d = 2, 40, 32760, 5120
qkv = torch.randn(B, L, 3 * d, device="cuda", generator=g).bfloat16()
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 128)) for i in range(3))
out = torch.empty(B, L, H, 128, device="cuda", dtype=torch.bfloat16)
_, ws = ops.attention_fp8(q, k, v, out=out, return_workspace=True)
nwg = (B * H * ((L + 255) // 256) + 7) // 8 * 8
buf = torch.zeros(nwg * 8, 8, dtype=torch.int64, device="cuda")
for _ in range(2):
    ops.attention_fp8(q, k, v, out=out, workspace=ws, stage=2, pmode=1)
torch.cuda.synchronize()
assert lib.vc_debug_set_attn_fp8_trace(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(int(os.environ.get('WARM', '0'))):       # the guide's clock check: stamp after >= 2 s of back-to-back launches
    ops.attention_fp8(q, k, v, out=out, workspace=ws, stage=2, pmode=1)
e0.record()
ops.attention_fp8(q, k, v, out=out, workspace=ws, stage=2, pmode=1)
e1.record()
torch.cuda.synchronize()
lib.vc_debug_set_attn_fp8_trace(None)
t = buf.cpu().double().view(nwg, 8, 8)
t = t[t[:, 0, 6] > 0]
nt = t[:, :, 6]
print(f"traced launch {e0.elapsed_time(e1):.2f} ms; {t.shape[0]} workgroups, {int(nt[0, 0])} key tiles each")
clk = (t[:, :, 5] / t[:, :, 7].clamp(min=1)) * 100e6
print(f"in-kernel clock (s_memtime / s_memrealtime around the loop, median over waves): {clk.median().item() / 1e9:.3f} GHz "
      f"(10th / 90th percentile {clk.flatten().kthvalue(max(1, int(0.1 * clk.numel()))).values.item() / 1e9:.3f} / "
      f"{clk.flatten().kthvalue(int(0.9 * clk.numel())).values.item() / 1e9:.3f})")
names = ["wait+barrier", "DMA issue", "phase 1", "phase 2", "loop control"]
for grp, sl in (("waves 0-3 (phase 1, phase 2)", slice(0, 4)), ("waves 4-7 (phase 2 of t-1, phase 1 of t)", slice(4, 8)), ("wave 0", slice(0, 1)), ("wave 1", slice(1, 2)), ("wave 2", slice(2, 3))):
    per = (t[:, sl, :5] / nt[:, sl, None]).mean(dim=(0, 1))
    tot = (t[:, sl, 5] / nt[:, sl]).mean()
    print(f"{grp}: {tot:.0f} ticks per tile = " + ", ".join(f"{n} {x:.0f} ({100 * x / tot:.0f} %)" for n, x in zip(names, per.tolist())))

#!/usr/bin/env python3
"""The clock the chip holds under the bf16 self-attention kernel (MI355X_MICROARCH.md, DVFS give-back item 6): a -DVC_ATTN_CLOCK build stamps
s_memtime (shader cycles) and s_memrealtime (100 MHz) around the tile loop of attn_fwd_pipe_kernel; clock = d(memtime) / d(memrealtime) x 100 MHz,
median over waves, after WARM back-to-back launches on gaussian data.
   tools/build_variant.sh aclock -DVC_ATTN_CLOCK && VC_ENGINE_LIB=$PWD/versecrafter_amd/libvcengine_aclock.so python tools/clock_attn.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from versecrafter_amd import _lib, ops
lib = _lib.load()
lib.vc_debug_set_attn_clock.argtypes = [ctypes.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
B, H, L, d = 2, 40, 32760, 5120
qkv = torch.randn(B, L, 3 * d, device="cuda", generator=g).bfloat16()
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 128)) for i in range(3))
out = torch.empty(B, L, H, 128, device="cuda", dtype=torch.bfloat16)
nwg = (B * H * ((L + 255) // 256) + 7) // 8 * 8
buf = torch.zeros(nwg * 8, 8, dtype=torch.int64, device="cuda")
ops.attention(q, k, v, out=out)
torch.cuda.synchronize()
assert lib.vc_debug_set_attn_clock(buf.data_ptr()) == 0
for _ in range(int(os.environ.get("WARM", "60"))):
    ops.attention(q, k, v, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.attention(q, k, v, out=out)
e1.record()
torch.cuda.synchronize()
lib.vc_debug_set_attn_clock(None)
t = buf.cpu().double()
t = t[t[:, 1] > 0]
clk = t[:, 0] / t[:, 1] * 100e6
ms = e0.elapsed_time(e1)
tf = 4.0 * B * H * L * L * 128 / (ms * 1e-3) / 1e12
ghz = clk.median().item() / 1e9
print(f"attn_fwd_pipe_kernel B=2 H=40 L=32760 after {os.environ.get('WARM', '60')} warm launches: {ms:.2f} ms = {tf:.0f} TFLOP/s; in-kernel clock {ghz:.3f} GHz "
      f"(10th / 90th percentile {clk.kthvalue(max(1, int(0.1 * clk.numel()))).values.item() / 1e9:.3f} / {clk.kthvalue(int(0.9 * clk.numel())).values.item() / 1e9:.3f}); "
      f"dense bf16 peak at that clock {2500 * ghz / 2.4:.0f} TFLOP/s -> fraction {tf / (2500 * ghz / 2.4):.3f}")
if t[:, 7].max() > 0:            # a -DVC_ATTN_TRACE build: s_memtime sums per section (the fences perturb hipcc's schedule: indicative only)
    nt = t[:, 7].clamp(min=1)
    per = (t[:, 2:7] / nt[:, None]).mean(0).tolist()
    tot = (t[:, 0] / nt).mean().item()
    names = ["wait + barrier", "K / V staging (4 LDS-DMA pieces)", "rescale test + phase 1", "phase 2 + row maximum", "loop control"]
    print(f"per tile and wave: {tot:.0f} clocks = " + ", ".join(f"{n} {x:.0f} ({100 * x / tot:.0f} %)" for n, x in zip(names, per)))

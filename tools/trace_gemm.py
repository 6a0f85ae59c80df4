#!/usr/bin/env python3
"""Where a GEMM output tile's time goes (GPU box only; needs a library built with -DVC_PP_TRACE, e.g.
   tools/build_variant.sh trace -DVC_PP_TRACE && VC_ENGINE_LIB=versecrafter_amd/libvcengine_trace.so python tools/trace_gemm.py).
Each workgroup of gemm_pp_kernel stamps the 100 MHz clock at entry, before its first K pair, after its last, after issuing its
stores and after they are acknowledged; grouped per CU this gives the prologue, loop, epilogue and the gap between one
workgroup's end and the next one's start on the same CU."""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from versecrafter_amd import _lib, ops

lib = _lib.load()
lib.vc_debug_set_gemm_trace.argtypes = [ctypes.c_void_p]
M, N = 65536, 5120
EPI = int(os.environ.get("EPI", "-1"))        # -1: no bias, plain store; 0 / 1 / 3: the engine's epilogues with bias (3: + gate, residual in place)
for K in (int(k) for k in (sys.argv[1:] or ["5120"])):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    out = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    kw = {}
    if EPI >= 0:
        kw = dict(bias=torch.randn(N, device="cuda", generator=g).bfloat16(), epilogue=EPI)
        if EPI == 3:
            kw.update(resid=out, gate=torch.randn(2, N, device="cuda", generator=g).bfloat16(), rows_per_batch=M // 2)
    ntiles = (M // 256) * (N // 256)
    grid = (ntiles + 7) // 8 * 8
    buf = torch.zeros(grid, 8, dtype=torch.int64, device="cuda")
    def run():
        if EPI >= 0:
            b_ = kw["bias"]
            ops.gemm(a, w, b_, out=out, tile=4, **{k: v for k, v in kw.items() if k != "bias"})
        else:
            ops.gemm(a, w, None, out=out, tile=4)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert lib.vc_debug_set_gemm_trace(buf.data_ptr()) == 0
    run()
    torch.cuda.synchronize()
    lib.vc_debug_set_gemm_trace(None)
    t = buf.cpu().numpy()
    t = t[t[:, 0] > 0]
    us = lambda x: x / 100.0                      # 100 MHz ticks -> us
    pro, loop, epi, ack = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2]), us(t[:, 7] - t[:, 3])
    cu = (t[:, 5] & 0xF) * 4096 + (t[:, 4] & 0xFF00) // 256 * 16 + ((t[:, 4] >> 13) & 7) * 2 + ((t[:, 4] >> 12) & 1) * 1 + 0
    key = (t[:, 5] & 0xF) * 65536 + (t[:, 4] & 0xFF00)          # (xcc, se/sh/cu bits 8..15)
    gaps, per_cu = [], {}
    for i in range(len(t)):
        per_cu.setdefault(int(key[i]), []).append((int(t[i, 0]), int(t[i, 7])))
    for k, v in per_cu.items():
        v.sort()
        for (s0, e0), (s1, e1) in zip(v, v[1:]):
            gaps.append(us(s1 - e0))
    span = us(t[:, 7].max() - t[:, 0].min())
    med = statistics.median
    print(f"K={K} EPI={EPI}: {len(t)} workgroups on {len(per_cu)} CUs, kernel span {span:.1f} us")
    print(f"  prologue (entry -> first K pair)      median {med(pro):6.2f} us   p90 {sorted(pro)[int(.9 * len(pro))]:6.2f}")
    print(f"  main loop                             median {med(loop):6.2f} us   p90 {sorted(loop)[int(.9 * len(loop))]:6.2f}"
          f"   ({med(loop) / (K // 64):.3f} us per K-tile)")
    print(f"  epilogue (loop end -> stores issued)  median {med(epi):6.2f} us   p90 {sorted(epi)[int(.9 * len(epi))]:6.2f}")
    print(f"  stores issued -> acknowledged (wave 0) median {med(ack):6.2f} us   p90 {sorted(ack)[int(.9 * len(ack))]:6.2f}")
    if gaps:
        print(f"  gap: previous workgroup's end -> next one's entry on the same CU  median {med(gaps):6.2f} us   "
              f"p90 {sorted(gaps)[int(.9 * len(gaps))]:6.2f}   (n = {len(gaps)})")
    # how synchronised are the CUs?  spread of the entry times of the k-th workgroup of each CU
    rounds = {}
    for k, v in per_cu.items():
        for r, (s0, e0) in enumerate(v):
            rounds.setdefault(r, []).append(s0)
    sp = [us(max(x) - min(x)) for r, x in sorted(rounds.items()) if len(x) > 200]
    print("  spread of entry times across CUs, by round: " + " ".join(f"{x:.0f}" for x in sp[:24]) + " us")

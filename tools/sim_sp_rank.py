#!/usr/bin/env python3
"""What-if timing of ONE rank's share of a P-way sequence-parallel denoise step on a single GPU (no peers).

    python tools/sim_sp_rank.py P [steps] [--gbps G] [--dual 0|1] [--json out.json]

The engine's simulated transport (include/vcengine.h: vc_sp_init_sim) replaces every exchange by a local copy plus one idle
wave that holds the chain's stream for (bytes leaving the rank) / G -- so the number is the per-rank time of an N = P run:
GEMMs at M = B*L/P rows, attention over the full sequence with heads/P, pack / unpack passes, the two-chain schedule, and the
part of the wire time the other chain's kernels do NOT hide.  Results are wrong by construction (no peer data): timing only.

Default G: xGMI is point-to-point, 7 links x ~153 GB/s per GPU.  An all-to-all among P ranks of one node uses P-1 of this
rank's links, one peer per link: G = (P-1) * 153 GB/s * 0.8 (P=8: ~857 GB/s, P=2: ~122 GB/s).  --gbps 0 = no wire time
(pure compute share)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from versecrafter_amd import _lib
from versecrafter_amd.models import VerseCrafterWanTransformer3DModel


class SimSP:
    """SequenceParallel interface (world_size, rank, attach) on the engine's simulated transport."""

    def __init__(self, P, gbps):
        self.world_size, self.rank, self.error, self.gbps = P, 0, None, gbps

    def attach(self, lib, handle):
        _lib.check(lib.vc_sp_init_sim(handle, self.world_size, self.rank, C.c_double(self.gbps)), handle)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("P", type=int, nargs="?", default=8)
    ap.add_argument("steps", type=int, nargs="?", default=2)
    ap.add_argument("--gbps", type=float, default=None, help="egress GB/s of this rank during an exchange; 0 = no wire time")
    ap.add_argument("--dual", type=int, default=None, help="force the two-chain schedule on (1) / off (0)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--batch", type=int, default=2, choices=(1, 2),
                    help="samples on this rank: 2 = the CFG pair (Ulysses over all ranks), 1 = one sample (cfg_degree 2: an "
                         "N-GPU run is then two groups of P = N/2)")
    args = ap.parse_args()
    P = args.P
    gbps = args.gbps if args.gbps is not None else max(1, P - 1) * 153.0 * 0.8
    if args.dual is not None:
        os.environ["VC_DUAL_LANE"] = str(args.dual)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = VerseCrafterWanTransformer3DModel(geoada_in_dim=128, param_device=dev, param_dtype=torch.bfloat16,
                                              dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, skip_init=True)
    model.init_weights(zero_init_outputs=False)
    if P > 1:
        model.enable_multi_gpus_inference(SimSP(P, gbps if gbps > 0 else 1e9))
    T, h, w = 21, 60, 104
    g = torch.Generator().manual_seed(2025)
    x = torch.randn(1, 16, T, h, w, generator=g).to(dev, torch.bfloat16).repeat(2, 1, 1, 1, 1)
    geo = torch.randn(1, 128, T, h, w, generator=g).to(dev, torch.bfloat16).repeat(2, 1, 1, 1, 1)
    ctx = [torch.randn(60, 4096, generator=g).to(dev, torch.bfloat16), torch.randn(77, 4096, generator=g).to(dev, torch.bfloat16)]
    t = torch.tensor([900.0, 900.0], device=dev)
    if args.batch == 1:
        x, geo, ctx, t = x[1:], geo[1:], ctx[1:], t[1:]
    L = T * (h // 2) * (w // 2)
    model(x, t, geo, ctx, L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if args.batch == 2:
            model.assert_cfg_pair(x)
        model(x, t, geo, ctx, L)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    model.profile_enable(True)
    if args.batch == 2:
        model.assert_cfg_pair(x)
    model(x, t, geo, ctx, L)
    torch.cuda.synchronize()
    prof = model.profile_read()
    # bytes leaving the rank per step: 60 blocks x (q|k|v + o) x (P-1)/P of [B*L/P, d] bf16
    Lloc = (L + P - 1) // P
    egress = 60 * 4 * args.batch * Lloc * 5120 * 2 * (P - 1) / P if P > 1 else 0.0
    wire_ms = egress / (gbps * 1e9) * 1e3 if gbps > 0 and P > 1 else 0.0
    out = {"P": P, "batch": args.batch, "ms_per_forward": dt * 1e3, "ideal_steps_per_s_at_N": 1 / dt, "egress_GB_per_step": egress / 1e9,
           "egress_gbps": gbps, "injected_wire_ms_per_step": wire_ms, "dual_lane": os.environ.get("VC_DUAL_LANE", "auto"),
           "classes": {k: {"ms": v["ms"], "tflops": v["flops"] / (v["ms"] / 1e3) / 1e12 if v["flops"] and v["ms"] else None}
                       for k, v in prof.items() if v["launches"]}}
    print(json.dumps(out))
    if args.json:
        with open(args.json, "a") as f:
            f.write(json.dumps(out) + "\n")


if __name__ == "__main__":
    main()

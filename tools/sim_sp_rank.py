#!/usr/bin/env python3
"""What-if timing of ONE rank's share of a P-way sequence-parallel step on a single GPU (no peers): the collective
callbacks just copy send -> recv locally (same bytes, HBM speed), so the number is the per-rank COMPUTE time of an
N = P run (GEMMs at M = B*L/P rows, attention over the full sequence with heads/P, pack/unpack kernels).
Results are wrong by construction (no real exchange) -- timing only.   python tools/sim_sp_rank.py 8"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from versecrafter_amd import _lib
from versecrafter_amd.dist import alias_device_bytes
from versecrafter_amd.models import VerseCrafterWanTransformer3DModel


class LoopbackSP:
    def __init__(self, P):
        self.world_size, self.rank, self.error = P, 0, None
        self.c_all_to_all = _lib.ALL_TO_ALL_FN(self._a2a)
        self.c_all_gather = _lib.ALL_GATHER_FN(self._ag)

    @staticmethod
    def _on(stream):
        return torch.cuda.stream(torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream())

    def _a2a(self, ctx, send, recv, bpp, stream):
        n = bpp * self.world_size
        with self._on(stream):
            alias_device_bytes(recv, n, "cuda").copy_(alias_device_bytes(send, n, "cuda"))
        return 0

    def _ag(self, ctx, send, recv, n, stream):
        with self._on(stream):
            r = alias_device_bytes(recv, n * self.world_size, "cuda")
            s = alias_device_bytes(send, n, "cuda")
            for i in range(self.world_size):
                r[i * n:(i + 1) * n].copy_(s)
        return 0


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = VerseCrafterWanTransformer3DModel(geoada_in_dim=128, param_device=dev, param_dtype=torch.bfloat16,
                                              dim=5120, ffn_dim=13824, num_heads=40, num_layers=40)
    model.init_weights(zero_init_outputs=False)
    if P > 1:
        model.enable_multi_gpus_inference(LoopbackSP(P))
    T, h, w = 21, 60, 104
    g = torch.Generator().manual_seed(2025)
    x = torch.randn(2, 16, T, h, w, generator=g).to(dev, torch.bfloat16)
    geo = torch.randn(2, 128, T, h, w, generator=g).to(dev, torch.bfloat16)
    ctx = [torch.randn(60, 4096, generator=g).to(dev, torch.bfloat16), torch.randn(77, 4096, generator=g).to(dev, torch.bfloat16)]
    t = torch.tensor([900.0, 900.0], device=dev)
    L = T * (h // 2) * (w // 2)
    model(x, t, geo, ctx, L)
    torch.cuda.synchronize()
    model.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        model(x, t, geo, ctx, L)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = model.profile_read()
    print(f"P={P}: {dt * 1e3:.1f} ms per forward of one rank ->  ideal N={P} rate {1 / dt:.3f} steps/s "
          f"(1-GPU-equivalent efficiency needs the N=1 time)")
    for k, v in prof.items():
        if v["launches"]:
            print(f"  {k:10s} {v['ms'] / steps:8.1f} ms/step  {v['flops'] / (v['ms'] / 1e3) / 1e12 if v['flops'] else 0:7.0f} TF")


if __name__ == "__main__":
    main()

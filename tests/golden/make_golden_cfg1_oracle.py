"""Records the CPU oracle's fp32 output of BASELINE config 1 at FULL depth (Wan2.1-1.3B + GeoAdapter: 30 + 15 blocks, 1920 tokens, CFG pair)
on the seeded inputs of tests/test_gpu_forward.py::test_cfg1_full_depth_forward_vs_oracle_and_four_step_sampler, so that the GPU suite
does not spend 85 s of CPU time on it every run (the oracle is oracle/wan_oracle.py, itself pinned against the reference's own WT.py /
VC.py by tests/golden/make_golden.py).  The weights and inputs are regenerated from the same torch seeds in the test; only the expected
output travels.  Run from the repo root:  python tests/golden/make_golden_cfg1_oracle.py      (minutes on 8 cores)"""
import os
import sys
import time

import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import wan_oracle as O  # noqa: E402


def prod_weights(cfg, seed=3):
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, shp in O.state_dict_shapes(cfg).items():
        if k.endswith("modulation"):
            w = torch.randn(shp, generator=g) / cfg.dim ** 0.5
        elif "norm" in k and k.endswith("weight"):
            w = 1 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            w = 0.02 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for s_ in shp[1:]:
                fan_in *= s_
            a = (6.0 / (fan_in + shp[0])) ** 0.5
            w = (torch.rand(shp, generator=g) * 2 - 1) * a
        W[k] = w.bfloat16()
    return W, g


def main():
    cfgk = dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, geoada_in_dim=128, in_dim=16, out_dim=16, text_dim=4096,
                text_len=512, freq_dim=256)
    cfg = O.Config(**cfgk)
    W, g = prod_weights(cfg)
    T, h, w_ = 3, 40, 64
    x = torch.randn(2, 16, T, h, w_, generator=g).bfloat16()
    geo = torch.randn(2, 128, T, h, w_, generator=g).bfloat16()
    ctx = [torch.randn(60, 4096, generator=g).bfloat16(), torch.randn(77, 4096, generator=g).bfloat16()]
    t = torch.tensor([700.0, 700.0])
    L = O.seq_len_for((16, T, h, w_))
    Wf = {k: v.float() for k, v in W.items()}
    t0 = time.time()
    want = O.forward(Wf, cfg, x.float(), t, geo.float(), [c.float() for c in ctx], L)
    print(f"oracle forward: {time.time() - t0:.1f} s", flush=True)
    # check sums of the inputs: the test regenerates them and must see the same numbers before it trusts the recorded output.  Integer sums
    # of the bf16 BIT PATTERNS: exact and independent of the summation order (an fp32 sum differs between hosts with different thread counts)
    bits = lambda v: v.contiguous().view(torch.int16).to(torch.int64).sum().reshape(1)
    save_file({"want": want.contiguous(), "x_bits": bits(x), "geo_bits": bits(geo), "w_bits": sum(bits(v) for v in W.values())},
              os.path.join(ROOT, "tests", "golden", "cfg1_full_depth_oracle.safetensors"))


if __name__ == "__main__":
    main()

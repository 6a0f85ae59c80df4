#!/usr/bin/env python3
"""Time the HIP Wan2.1 VAE at the bench clip's size (81 frames 480x832 -> latent [16,21,60,104]), random weights (GPU box):
one control-video encode and one decode.   python tools/bench_vae.py [frames H W [chunk]]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from versecrafter_amd.models.wan_vae import AutoencoderKLWan, vae_state_dict_shapes


def conv_flops(F, H, W):
    """2 * MACs of every convolution of the encoder / decoder walk (attention and norms not counted)."""
    shapes = vae_state_dict_shapes()
    enc = dec = 0.0
    # walk the geometry exactly as the engine does
    def px(t, h, w): return t * h * w
    T2, T4 = 1 + (F - 1) // 2, 1 + (F - 1) // 4
    geo_enc = {"encoder.conv1": px(F, H, W)}
    lvl = [(F, H, W), (F, H // 2, W // 2), (T2, H // 4, W // 4), (T4, H // 8, W // 8)]
    idx = 0
    for i in range(4):
        for _ in range(2):
            geo_enc[f"encoder.downsamples.{idx}."] = px(*lvl[i]); idx += 1
        if i != 3:
            geo_enc[f"encoder.downsamples.{idx}.resample"] = px(lvl[i][0], lvl[i][1] // 2, lvl[i][2] // 2)
            geo_enc[f"encoder.downsamples.{idx}.time_conv"] = px(*lvl[i + 1]); idx += 1
    for k in ("encoder.middle.0.", "encoder.middle.1.", "encoder.middle.2.", "encoder.head.2", "conv1"):
        geo_enc[k] = px(*lvl[3])
    dl = [(T4, H // 8, W // 8), (T2, H // 4, W // 4), (F, H // 2, W // 2), (F, H, W)]
    geo_dec = {"conv2": px(*dl[0]), "decoder.conv1": px(*dl[0]), "decoder.middle.0.": px(*dl[0]), "decoder.middle.1.": px(*dl[0]),
               "decoder.middle.2.": px(*dl[0]), "decoder.head.2": px(*dl[3])}
    idx = 0
    for i in range(4):
        for _ in range(3):
            geo_dec[f"decoder.upsamples.{idx}."] = px(*dl[i]); idx += 1
        if i != 3:
            geo_dec[f"decoder.upsamples.{idx}.time_conv"] = px(dl[i][0] - 1, dl[i][1], dl[i][2])
            geo_dec[f"decoder.upsamples.{idx}.resample"] = px(dl[i + 1][0], dl[i + 1][1], dl[i + 1][2]); idx += 1
    for key, shp in shapes.items():
        if not key.endswith("weight"):
            continue
        macs = 1
        for v in shp:
            macs *= v
        for table, is_enc in ((geo_enc, True), (geo_dec, False)):
            hit = [p for p in table if key.startswith(p)]
            if hit:
                p = max(hit, key=len)
                if is_enc: enc += 2.0 * macs * table[p]
                else: dec += 2.0 * macs * table[p]
    return enc, dec


def main():
    F, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (81, 480, 832)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = AutoencoderKLWan(param_device=dev)
    for k, p in m.named_parameters():
        if k.endswith("gamma"):
            p.data.fill_(1.0)
        elif k.endswith("bias"):
            p.data.zero_()
        else:
            fan = p[0].numel()
            p.data.normal_(0, 1.2 / fan ** 0.5)
    x = (torch.rand(1, 3, F, H, W, device=dev) * 2 - 1).bfloat16()
    fe, fd = conv_flops(F, H, W)
    chunk = int(sys.argv[4]) if len(sys.argv) > 4 else -1        # -1 automatic (chunks above 40 GB), 0 whole sequence, n frames per chunk
    m.set_time_chunk(chunk)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        z = m.encode(x)[0].mode()
        torch.cuda.synchronize(); te = time.perf_counter() - t0
        t0 = time.perf_counter()
        y = m.decode(z).sample
        torch.cuda.synchronize(); td = time.perf_counter() - t0
    print(f"VAE {F}x{H}x{W}: encode {te * 1e3:.0f} ms ({fe / te / 1e12:.0f} TFLOP/s of {fe / 1e12:.1f} conv TFLOP), decode {td * 1e3:.0f} ms "
          f"({fd / td / 1e12:.0f} TFLOP/s of {fd / 1e12:.1f}), time chunk {m.last_time_chunk()}, workspace {m.workspace_bytes() / 2**30:.1f} GiB, latent {tuple(z.shape)}, "
          f"finite {bool(torch.isfinite(y.float()).all())}")


if __name__ == "__main__":
    main()

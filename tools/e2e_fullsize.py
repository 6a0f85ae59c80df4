#!/usr/bin/env python3
"""Whole flow at the bench clip's size on ONE MI355X, synthetic everything (no checkpoint is reachable offline):
  1. inference/rendering_4D_control_maps.py  -- synthetic scene (PNG + depth npz + object mask + 81-frame camera trajectory + ellipsoid
     json, the file set of the reference's demo_data folders) -> the five control videos (.mp4, the package's own I_PCM writer)
  2. inference/versecrafter_inference.py     -- Wan2.1-14B + GeoAdapter (random weights), production-width Wan VAE (random weights),
     random prompt embeddings, N denoise steps, VAE decode -> generated_video_0.mp4
   python tools/e2e_fullsize.py [workdir] [steps] [H] [W] [moe] [fp8]      moe: a second 14B expert runs the high-noise steps (BASELINE config 5's
   pair); fp8: --fp8_linear 1 --fp8_attention 1 (config 5's dtype)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
from safetensors.torch import save_file


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/vc_e2e"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 480
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 832
    moe = "moe" in sys.argv[5:]
    fp8 = "fp8" in sys.argv[5:]
    F_ = 81
    os.makedirs(work, exist_ok=True)
    rs = np.random.RandomState(0)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx + yy) % 256)], -1).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(work, "0001.png"))
    depth = (3.0 + 0.5 * np.sin(xx / 40.0) + 0.3 * rs.rand(H, W)).astype(np.float32)
    np.savez(os.path.join(work, "0001.npz"), depth=depth, intrinsic=np.array([[0.9, 0, 0.5], [0, 0.9 * W / H, 0.5], [0, 0, 1]], dtype=np.float32))
    os.makedirs(os.path.join(work, "masks"), exist_ok=True)
    m = np.zeros((H, W), dtype=np.uint8)
    m[H // 3:2 * H // 3, W // 3:W // 2] = 255
    Image.fromarray(m).save(os.path.join(work, "masks", "obj1.png"))
    c2w = np.tile(np.eye(4), (F_, 1, 1))
    c2w[:, :3, :3] = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64)          # Blender camera looking along world +Y
    c2w[:, 0, 3] = np.linspace(0, 0.4, F_)
    np.savez(os.path.join(work, "custom_camera_trajectory.npz"), extrinsics=c2w)
    doc = {"metadata": {"num_frames": F_, "num_objects": 2, "obj_id_to_color_idx": {"1": 2, "2": 6}},
           "frames": [{"frame_index": f, "objects": [
               {"object_id": 1, "gaussian_3d": {"mean": [0.01 * f, 2.2, 0.0], "covariance": [[0.05, 0, 0], [0, 0.03, 0], [0, 0, 0.08]]}},
               {"object_id": 2, "gaussian_3d": {"mean": [-0.5, 2.8 - 0.005 * f, 0.2], "covariance": [[0.02, 0.01, 0], [0.01, 0.04, 0], [0, 0, 0.02]]}}]}
               for f in range(F_)]}
    json.dump(doc, open(os.path.join(work, "ell.json"), "w"))
    maps = os.path.join(work, "rendering_4D_maps")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "inference", "rendering_4D_control_maps.py"), "--png_path", os.path.join(work, "0001.png"),
                        "--npz_path", os.path.join(work, "0001.npz"), "--mask_dir", os.path.join(work, "masks"), "--trajectory_npz",
                        os.path.join(work, "custom_camera_trajectory.npz"), "--ellipsoid_json", os.path.join(work, "ell.json"), "--output_dir", maps],
                       capture_output=True, text=True)
    print(f"[1] renderer: rc {r.returncode}, {time.time() - t0:.1f} s wall (incl. python start-up)")
    print("    " + "\n    ".join(r.stderr.strip().splitlines()[-3:]))
    if r.returncode:
        raise SystemExit(r.stderr[-3000:])
    for n in sorted(os.listdir(maps)):
        a = np.load(os.path.join(maps, n)) if n.endswith(".npy") else None
        if n.endswith(".mp4"):                               # the package's own I_PCM writer
            from versecrafter_amd.utils import mp4_pcm
            a = mp4_pcm.read_mp4(os.path.join(maps, n)).cpu().numpy()
        print(f"    {n}: {os.path.getsize(os.path.join(maps, n)) >> 20} MiB {None if a is None else (a.shape, a.dtype, round(float(a.mean()), 2))}")

    from oracle import vae_oracle as V                       # random weights of the published VAE architecture (names + shapes)
    cfg = V.Config(dim=96, z_dim=16)
    save_file({k: v.bfloat16() for k, v in V.random_weights(cfg, 3).items()}, os.path.join(work, "vae.safetensors"))
    g = torch.Generator().manual_seed(0)
    save_file({"prompt_embeds": torch.randn(77, 4096, generator=g).bfloat16(), "negative_prompt_embeds": torch.randn(60, 4096, generator=g).bfloat16()},
              os.path.join(work, "embeds.safetensors"))
    out_dir = os.path.join(work, "out")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "inference", "versecrafter_inference.py"), "--rendering_maps_path", maps, "--prompt",
                        "a car drives along a road", "--input_image_path", os.path.join(work, "0001.png"), "--ulysses_degree", "1", "--ring_degree", "1",
                        "--num_inference_steps", str(steps), "--sample_size", f"{H},{W}", "--video_length", str(F_), "--save_path", out_dir,
                        "--synthetic_model", "14b", "--vae_path", os.path.join(work, "vae.safetensors"), "--prompt_embeds_path",
                        os.path.join(work, "embeds.safetensors"), "--output_latents", "1"] +
                       (["--synthetic_high_noise_expert", "--shift", "12"] if moe else []) +
                       (["--fp8_linear", "1", "--fp8_attention", "1"] if fp8 else []), capture_output=True, text=True)
    print(f"[2] inference CLI: rc {r.returncode}, {time.time() - t0:.1f} s wall ({'two 14B experts' if moe else '14B'}{', fp8 linear layers + fp8 self-attention' if fp8 else ''} random init + 4 VAE encodes + {steps} steps + decode)")
    print("    " + "\n    ".join((r.stdout.strip().splitlines() or [""])[-3:]))
    if r.returncode:
        raise SystemExit(r.stderr[-3000:])
    for n in sorted(os.listdir(out_dir)):
        pth = os.path.join(out_dir, n)
        if n.endswith(".npy"):
            a = np.load(pth)
            print(f"    {n}: {a.shape} {a.dtype} mean {a.mean():.2f} std {a.std():.2f}")
            assert a.shape == (F_, H, W, 3) and a.dtype == np.uint8 and a.std() > 0
        elif n.endswith(".mp4"):
            from versecrafter_amd.utils import mp4_pcm
            a = mp4_pcm.read_mp4(pth).cpu().numpy()
            print(f"    {n}: {os.path.getsize(pth) >> 20} MiB, decodes to {a.shape} {a.dtype} mean {a.mean():.2f} std {a.std():.2f}")
            assert a.shape == (F_, H, W, 3) and a.std() > 0
        else:
            print(f"    {n}: {os.path.getsize(pth)} bytes")
    print("e2e ok")


if __name__ == "__main__":
    main()

"""Wan2.1 VAE on the HIP engine (csrc/vae.hip behind vc_vae_*, Python mirror models/wan_vae.AutoencoderKLWan) against
oracle/vae_oracle.py -- fp32 PyTorch restatement of the published architecture, PARITY UNPINNED against upstream (the class
and its weights are absent from the reference tree; see the oracle's header).  Tolerance: the engine keeps activations in bf16
between layers (as the reference does: CLI.py:223 casts the VAE to bf16) with fp32 accumulation; relative L2 against the fp32
oracle on bf16-rounded weights must stay below 3e-2 through the ~30 (encoder) / ~45 (decoder) convolutions."""
import pytest
import torch

from oracle import vae_oracle as V

pytestmark = pytest.mark.gpu

TINY = dict(dim=32, z_dim=16)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def make(cfgk, seed):
    from versecrafter_amd.models.wan_vae import AutoencoderKLWan
    cfg = V.Config(**cfgk)
    W = {k: v.bfloat16() for k, v in V.random_weights(cfg, seed).items()}
    m = AutoencoderKLWan(latent_channels=cfg.z_dim, dim=cfg.dim, dim_mult=tuple(cfg.dim_mult), num_res_blocks=cfg.num_res_blocks)
    missing, unexpected = m.load_state_dict({"model." + k: v for k, v in W.items()})       # VideoX-Fun's prefix is accepted
    assert not missing and not unexpected
    return cfg, {k: v.float() for k, v in W.items()}, m.to("cuda")


@pytest.mark.parametrize("frames,H,W", [(1, 32, 32), (5, 32, 48), (9, 48, 32), (13, 32, 32)])
def test_encode_matches_oracle(frames, H, W):
    cfg, Wf, m = make(TINY, 3)
    g = torch.Generator().manual_seed(frames)
    x = (torch.rand(2, 3, frames, H, W, generator=g) * 2 - 1).bfloat16()
    got = m.encode(x.cuda())[0].mode()
    torch.cuda.synchronize()
    with torch.no_grad():
        want = V.encode(Wf, cfg, x.float())
    assert got.shape == want.shape == (2, 16, 1 + (frames - 1) // 4, H // 8, W // 8)
    assert torch.isfinite(got.float()).all()
    e = rel(got, want)
    print(f"encode {frames}x{H}x{W}: rel L2 {e:.4g}")
    assert e < 3e-2


@pytest.mark.parametrize("T,h,w", [(1, 4, 4), (2, 4, 6), (3, 6, 4), (4, 4, 4)])
def test_decode_matches_oracle(T, h, w):
    cfg, Wf, m = make(TINY, 5)
    g = torch.Generator().manual_seed(T * 7 + h)
    z = torch.randn(2, 16, T, h, w, generator=g).bfloat16()
    got = m.decode(z.cuda()).sample
    torch.cuda.synchronize()
    with torch.no_grad():
        want = V.decode(Wf, cfg, z.float())
    assert got.shape == want.shape == (2, 3, 1 + 4 * (T - 1), 8 * h, 8 * w)
    assert torch.isfinite(got.float()).all() and got.float().abs().max() <= 1.0
    e = rel(got, want)
    print(f"decode {T}x{h}x{w}: rel L2 {e:.4g}")
    assert e < 3e-2


def test_errors_and_contract():
    cfg, Wf, m = make(TINY, 3)
    assert m.config.latent_channels == 16 and m.temporal_compression_ratio == 4 and m.spatial_compression_ratio == 8
    assert m.dtype == torch.bfloat16
    with pytest.raises(ValueError):
        m.encode(torch.zeros(1, 3, 6, 32, 32, device="cuda", dtype=torch.bfloat16))      # F != 1 + 4n
    with pytest.raises(ValueError):
        m.encode(torch.zeros(1, 3, 5, 40, 32, device="cuda", dtype=torch.bfloat16))      # H % 16
    with pytest.raises(RuntimeError):
        m.encode(torch.zeros(1, 3, 5, 32, 32, dtype=torch.bfloat16))                     # CPU tensor
    a = m.encode(torch.zeros(1, 3, 5, 32, 32, device="cuda", dtype=torch.bfloat16))
    assert a.latent_dist.mode().shape == (1, 16, 2, 4, 4) and m.workspace_bytes() > 0


def test_production_width_small_clip_vs_oracle():
    """The published widths (dim 96 -> 96 / 192 / 384 / 384 channels, 127 M parameters: channel padding 96 -> 128, 64- and
    128-column tiles, q|k|v of 3 x 384, time-doubling convs of 768 columns) on a 5-frame 64x64 clip the CPU oracle finishes
    in seconds: encode, decode, and the encode -> decode round trip shape contract."""
    cfg, Wf, m = make(dict(dim=96, z_dim=16), 11)
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(1, 3, 5, 64, 64, generator=g) * 2 - 1).bfloat16()
    got = m.encode(x.cuda())[0].mode()
    with torch.no_grad():
        want = V.encode(Wf, cfg, x.float())
    e1 = rel(got, want)
    z = torch.randn(1, 16, 2, 8, 8, generator=g).bfloat16()
    dec = m.decode(z.cuda()).sample
    with torch.no_grad():
        wdec = V.decode(Wf, cfg, z.float())
    e2 = rel(dec, wdec)
    print(f"production width: encode rel L2 {e1:.4g}, decode rel L2 {e2:.4g}")
    assert got.shape == (1, 16, 2, 8, 8) and dec.shape == (1, 3, 5, 64, 64)
    assert e1 < 3e-2 and e2 < 3e-2


@pytest.mark.parametrize("chunk", [1, 3, 4, 8])
def test_time_chunked_full_resolution_stage_is_bit_identical(chunk):
    """The full-resolution stage in time chunks (two frames of history per causal convolution -- upstream's own execution order)
    against the whole-sequence walk: encode of 13 frames and decode of 4 latent frames (13 output frames), chunk sizes that
    divide the clip, leave a ragged tail, or are a single frame; the arena of the chunked walk must be the smaller one."""
    cfg, Wf, m = make(TINY, 3)
    g = torch.Generator().manual_seed(chunk)
    x = (torch.rand(1, 3, 13, 32, 48, generator=g) * 2 - 1).bfloat16().cuda()
    z = torch.randn(1, 16, 4, 4, 6, generator=g).bfloat16().cuda()
    m.set_time_chunk(0)
    e0, d0 = m.encode(x)[0].mode(), m.decode(z).sample
    assert m.last_time_chunk() == 0
    torch.cuda.synchronize()
    ws0 = m.workspace_bytes()
    m.release_workspace()
    m.set_time_chunk(chunk)
    e1 = m.encode(x)[0].mode()
    assert m.last_time_chunk() == chunk
    d1 = m.decode(z).sample
    torch.cuda.synchronize()
    assert torch.equal(e0, e1) and torch.equal(d0, d1)
    assert m.workspace_bytes() < ws0
    m.release_workspace()


def test_config4_frame_size_encode_decode_properties():
    """BASELINE config 4's clip, 81 frames of 720 x 1280, through the production-width VAE (random weights; the CPU oracle is
    out of reach at this size -- ~1.5 PFLOP): shapes, finiteness, the decoder's clamp, and two size-independent properties --
    the encoder is CAUSAL in time (the first 41 frames alone give the first 11 latent frames of the whole clip bit for bit), and the
    time-chunked walk the default policy picks at this size (workspace < 40 GB) equals the whole-sequence walk (137 GiB) bit for bit.
    Round 3: this size first exposed launches of more than 2^32 threads, which HIP refuses -- silently, until every launch was checked."""
    cfg, Wf, m = make(dict(dim=96, z_dim=16), 11)
    g = torch.Generator().manual_seed(4)
    F, H, W = 81, 720, 1280
    x = (torch.rand(1, 3, F, H, W, generator=g) * 2 - 1).bfloat16().cuda()
    lat = m.encode(x)[0].mode()
    torch.cuda.synchronize()
    ws_enc = m.workspace_bytes()
    assert lat.shape == (1, 16, 21, H // 8, W // 8) and torch.isfinite(lat.float()).all()
    head = m.encode(x[:, :, :41].contiguous())[0].mode()
    torch.cuda.synchronize()
    assert head.shape == (1, 16, 11, H // 8, W // 8) and torch.equal(head, lat[:, :, :11])
    del head
    z = torch.randn(1, 16, 21, H // 8, W // 8, generator=g).bfloat16().cuda()
    vid = m.decode(z).sample
    torch.cuda.synchronize()
    ws_dec = m.workspace_bytes()
    chunk = m.last_time_chunk()
    assert vid.shape == (1, 3, F, H, W) and torch.isfinite(vid.float()).all() and vid.float().abs().max() <= 1.0
    assert vid.float().std() > 0
    # default policy at this size: the full-resolution stage runs in time chunks and the arena stays below 40 GB ...
    assert chunk == 8 and ws_enc < 40e9 and ws_dec < 40e9
    # ... and gives the bytes of the whole-sequence walk (five times the memory)
    m.release_workspace()
    m.set_time_chunk(0)
    vid_whole = m.decode(z).sample
    torch.cuda.synchronize()
    ws_whole = m.workspace_bytes()
    assert m.last_time_chunk() == 0 and torch.equal(vid, vid_whole)
    del vid_whole
    m.release_workspace()
    lat_whole = m.encode(x)[0].mode()
    torch.cuda.synchronize()
    ws_whole_enc = m.workspace_bytes()
    assert torch.equal(lat, lat_whole)
    print(f"config-4 clip: VAE workspace chunked (8 frames) {ws_enc / 2**30:.1f} GiB encode / {ws_dec / 2**30:.1f} GiB decode; "
          f"whole sequence {ws_whole_enc / 2**30:.1f} / {ws_whole / 2**30:.1f} GiB")
    m.release_workspace()


def test_cli_full_flow_with_vae_and_frame_dumps(tmp_path):
    """inference/versecrafter_inference.py as the reference runs it (CLI.py:187-465), on tiny random models: control maps and
    mask read from frame dumps next to the (absent) .mp4 names, first frame from a .png, VAE-encoded on the engine, denoised,
    VAE-decoded, written as a video (frame dump: the image has no codec)."""
    import json
    import os
    import subprocess
    import sys
    import numpy as np
    from PIL import Image
    from safetensors.torch import save_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    maps = tmp_path / "rendering_4D_maps"
    maps.mkdir()
    rs = np.random.RandomState(0)
    F_, H, W = 9, 64, 96
    for name in ("background_RGB", "background_depth", "3D_gaussian_RGB", "3D_gaussian_depth"):
        np.save(maps / f"{name}.npy", rs.randint(0, 255, (F_, H, W, 3), dtype=np.uint8))
    np.save(maps / "merged_mask.npy", (rs.rand(F_, H, W) < 0.5).astype(np.uint8) * 255)
    Image.fromarray(rs.randint(0, 255, (H, W, 3), dtype=np.uint8)).save(tmp_path / "0001.png")
    cfg = V.Config(dim=32, z_dim=16)
    save_file({k: v.bfloat16() for k, v in V.random_weights(cfg, 3).items()}, str(tmp_path / "vae.safetensors"))
    g = torch.Generator().manual_seed(0)
    save_file({"prompt_embeds": torch.randn(33, 4096, generator=g).bfloat16(),
               "negative_prompt_embeds": torch.randn(20, 4096, generator=g).bfloat16()}, str(tmp_path / "embeds.safetensors"))
    out_dir = tmp_path / "out"
    cmd = [sys.executable, os.path.join(root, "inference", "versecrafter_inference.py"), "--rendering_maps_path", str(maps),
           "--prompt", "a car drives", "--input_image_path", str(tmp_path / "0001.png"), "--ulysses_degree", "1", "--ring_degree", "1",
           "--num_inference_steps", "4", "--sample_size", f"{H},{W}", "--video_length", str(F_), "--save_path", str(out_dir),
           "--synthetic_model", "tiny", "--num_skip_start_steps", "2", "--vae_path", str(tmp_path / "vae.safetensors"),
           "--vae_kwargs", json.dumps({"dim": 32}), "--prompt_embeds_path", str(tmp_path / "embeds.safetensors"), "--output_latents", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    frames = np.load(out_dir / "generated_video_0.npy") if (out_dir / "generated_video_0.npy").exists() else None
    assert frames is not None or (out_dir / "generated_video_0.mp4").exists(), os.listdir(out_dir)
    if frames is not None:
        assert frames.shape == (F_, H, W, 3) and frames.dtype == np.uint8 and frames.std() > 0
    from safetensors.torch import load_file
    lat = load_file(str(out_dir / "generated_latents_0.safetensors"))["latents"]
    assert lat.shape == (1, 16, 3, H // 8, W // 8) and torch.isfinite(lat).all()

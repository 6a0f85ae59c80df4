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
    from versecrafter_amd import _lib
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

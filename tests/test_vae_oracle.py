"""oracle/vae_oracle.py (Wan2.1 VAE restatement, PARITY UNPINNED against upstream -- see its header): the upstream execution
order (frame 0, then 4-frame chunks with per-conv feature caches; decode one latent frame at a time with the 'Rep' rule of
upsample3d) and the whole-sequence form the HIP implementation follows must agree, and the published size contract holds:
F = 1 + 4n frames <-> 1 + n latent frames, 8x spatial, 16 latent channels (config/wan2.1/wan_civitai.yaml:8-13)."""
import pytest
import torch

from oracle import vae_oracle as V


@pytest.mark.parametrize("frames,H,W", [(1, 16, 16), (5, 16, 24), (9, 32, 16), (13, 16, 16)])
def test_chunked_and_whole_sequence_forms_agree(frames, H, W):
    torch.manual_seed(frames)
    cfg = V.Config(dim=8, z_dim=4)
    Wt = V.random_weights(cfg, 3)
    x = torch.rand(2, 3, frames, H, W) * 2 - 1
    with torch.no_grad():
        a, b = V.encode(Wt, cfg, x), V.encode_chunked(Wt, cfg, x)
    n = (frames - 1) // 4
    assert a.shape == (2, 4, 1 + n, H // 8, W // 8) and torch.allclose(a, b, atol=2e-5, rtol=1e-4), (a - b).abs().max()
    z = torch.randn(2, 4, 1 + n, H // 8, W // 8)
    with torch.no_grad():
        c, d = V.decode(Wt, cfg, z), V.decode_chunked(Wt, cfg, z)
    assert c.shape == (2, 3, frames, H, W) and torch.allclose(c, d, atol=2e-5, rtol=1e-4), (c - d).abs().max()
    assert c.abs().max() <= 1.0


def test_encoder_is_causal_in_time():
    """Latent frame k depends on video frames 0 .. 4k only (causal convolutions): perturbing later frames leaves it unchanged."""
    cfg = V.Config(dim=8, z_dim=4)
    Wt = V.random_weights(cfg, 5)
    x = torch.rand(1, 3, 13, 16, 16) * 2 - 1
    y = x.clone()
    y[:, :, 9:] += 0.3
    with torch.no_grad():
        a, b = V.encode(Wt, cfg, x), V.encode(Wt, cfg, y)
    assert torch.equal(a[:, :, :3], b[:, :, :3]) and not torch.allclose(a[:, :, 3], b[:, :, 3])


def test_production_state_dict_inventory():
    """dim 96, z 16, dim_mult (1,2,4,4), 2 / 3 res blocks per level, time down at levels 1 and 2: 127 M parameters
    (the size of the published Wan2.1_VAE checkpoint)."""
    cfg = V.Config()
    shapes = V.state_dict_shapes(cfg)
    n = sum(int(torch.tensor(s).prod()) for s in shapes.values())
    assert 126e6 < n < 128e6, n
    assert shapes["encoder.downsamples.5.time_conv.weight"] == (192, 192, 3, 1, 1)
    assert shapes["decoder.upsamples.3.time_conv.weight"] == (768, 384, 3, 1, 1)
    assert shapes["decoder.upsamples.3.resample.1.weight"] == (192, 384, 3, 3)
    assert shapes["decoder.upsamples.4.shortcut.weight"] == (384, 192, 1, 1, 1)
    assert shapes["decoder.head.2.weight"] == (3, 96, 3, 3, 3)


def test_mirror_inventory_equals_oracle_inventory_and_no_gpu_is_loud():
    """The Python mirror's parameter inventory (what it registers and hands to vc_vae_load_weight) is the oracle's, key by key;
    without a HIP device the engine refuses to construct (no CPU path)."""
    import ctypes
    from versecrafter_amd import _lib
    from versecrafter_amd.models.wan_vae import AutoencoderKLWan, vae_state_dict_shapes
    for cfgk in (dict(dim=96, z_dim=16), dict(dim=32, z_dim=16), dict(dim=64, z_dim=8)):
        want = V.state_dict_shapes(V.Config(**cfgk))
        assert vae_state_dict_shapes(cfgk["dim"], cfgk["z_dim"]) == want
    m = AutoencoderKLWan(dim=32, param_device="meta")
    assert {k: tuple(p.shape) for k, p in m.named_parameters()} == V.state_dict_shapes(V.Config(dim=32, z_dim=16))
    assert m.config.latent_channels == 16 and m.spacial_compression_ratio == 8
    if not torch.cuda.is_available():
        lib = _lib.load()
        cfg = _lib.vc_vae_config()
        cfg.dim, cfg.z_dim, cfg.num_res_blocks = 32, 16, 2
        for i, v in enumerate((1, 2, 4, 4)):
            cfg.dim_mult[i] = v
        for i, v in enumerate((0, 1, 1)):
            cfg.temporal_downsample[i] = v
        h = ctypes.c_void_p()
        assert lib.vc_vae_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.VC_E_HIP
        assert b"no HIP device" in lib.vc_vae_last_error(None)

"""CPU ORACLE (test infrastructure) for the Wan2.1 video VAE -- the per-video stage on either side of the denoise loop:
`vae.encode(frames)[0].mode()` for the four control videos (versecrafter/pipeline/pipeline_wan_versecrafter.py:397-438) and
`decode_latents` (PIPE.py:550-555), constructed at inference/versecrafter_inference.py:220-236 from
config/wan2.1/wan_civitai.yaml:8-13 (temporal ratio 4, spatial ratio 8, 16 latent channels).

PARITY UNPINNED.  The class the reference uses (videox_fun.models.AutoencoderKLWan) is an un-vendored submodule and its
weights are not in the reference tree; nothing importable in this container implements it.  This file restates the PUBLISHED
Wan2.1 VAE architecture (Wan2.1 `wan/modules/vae.py`, which VideoX-Fun wraps) from its description:
  * CausalConv3d: zero padding of 2 frames in FRONT of the time axis, symmetric spatial padding;
  * RMS_norm: L2-normalise over channels * sqrt(C) * gamma;  ResidualBlock: norm-SiLU-conv3-norm-SiLU-conv3 (+1x1x1 shortcut);
  * AttentionBlock: single-head self-attention over the h*w positions of each frame;
  * Resample: downsample2d = ZeroPad2d(0,1,0,1) + Conv2d(3, stride 2); downsample3d adds a (3,1,1) stride-2 time conv that the
    FIRST frame bypasses; upsample2d = nearest 2x + Conv2d(C -> C/2, 3); upsample3d first doubles the frames of every frame but
    the first through a causal (3,1,1) conv to 2C channels (the first frame is not part of its history);
  * encoder: conv 3->dim, [2 res blocks (+ resample)] x 4 levels (dim_mult 1,2,4,4; time down at levels 1,2), middle
    (res, attn, res), head norm-SiLU-conv -> 2z; conv1 (1x1x1) -> (mu, logvar); mu normalised with the published per-channel
    mean / std;  decoder mirrored with 3 res blocks per level.
Two formulations are given and tested against each other (tests/test_vae_oracle.py): the upstream EXECUTION ORDER -- frame 0
alone, then chunks of 4 frames (encode) / 1 latent frame (decode) with per-conv feature caches -- and the WHOLE-SEQUENCE form
the HIP implementation uses (every op once over the full clip).  Their agreement pins the restatement's internal
consistency, not its agreement with the upstream weights' behaviour."""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
CACHE_T = 2

# published per-channel statistics of the Wan2.1 latent space (wan/modules/vae.py); (mu - mean) / std on encode
LATENT_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
               0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
LATENT_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
              3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


@dataclass
class Config:
    dim: int = 96
    z_dim: int = 16
    dim_mult: List[int] = field(default_factory=lambda: [1, 2, 4, 4])
    num_res_blocks: int = 2
    temporal_downsample: List[bool] = field(default_factory=lambda: [False, True, True])

    @property
    def enc_dims(self):
        return [self.dim * m for m in [1] + list(self.dim_mult)]

    @property
    def dec_dims(self):
        return [self.dim * m for m in [self.dim_mult[-1]] + list(self.dim_mult[::-1])]


# ------------------------------------------------------------------------------------------ layer inventory
def encoder_layers(cfg: Config):
    """[(key prefix, kind, in_dim, out_dim)] of encoder.downsamples in upstream order."""
    out, dims = [], cfg.enc_dims
    idx = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(cfg.num_res_blocks):
            out.append((f"encoder.downsamples.{idx}.", "res", cin, cout))
            cin = cout
            idx += 1
        if i != len(cfg.dim_mult) - 1:
            out.append((f"encoder.downsamples.{idx}.", "down3d" if cfg.temporal_downsample[i] else "down2d", cout, cout))
            idx += 1
    return out


def decoder_layers(cfg: Config):
    out, dims = [], cfg.dec_dims
    up = list(cfg.temporal_downsample[::-1])
    idx = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(cfg.num_res_blocks + 1):
            out.append((f"decoder.upsamples.{idx}.", "res", cin, cout))
            cin = cout
            idx += 1
        if i != len(cfg.dim_mult) - 1:
            out.append((f"decoder.upsamples.{idx}.", "up3d" if up[i] else "up2d", cout, cout // 2))
            idx += 1
    return out


def state_dict_shapes(cfg: Config) -> Dict[str, tuple]:
    """Key -> shape of the upstream Wan2.1_VAE state dict (VideoX-Fun prefixes every key with "model.")."""
    s = {}

    def res(p, cin, cout):
        s[p + "residual.0.gamma"] = (cin, 1, 1, 1)
        s[p + "residual.2.weight"] = (cout, cin, 3, 3, 3)
        s[p + "residual.2.bias"] = (cout,)
        s[p + "residual.3.gamma"] = (cout, 1, 1, 1)
        s[p + "residual.6.weight"] = (cout, cout, 3, 3, 3)
        s[p + "residual.6.bias"] = (cout,)
        if cin != cout:
            s[p + "shortcut.weight"] = (cout, cin, 1, 1, 1)
            s[p + "shortcut.bias"] = (cout,)

    def attn(p, c):
        s[p + "norm.gamma"] = (c, 1, 1)
        s[p + "to_qkv.weight"] = (3 * c, c, 1, 1)
        s[p + "to_qkv.bias"] = (3 * c,)
        s[p + "proj.weight"] = (c, c, 1, 1)
        s[p + "proj.bias"] = (c,)

    d0, top = cfg.dim, cfg.dim * cfg.dim_mult[-1]
    s["encoder.conv1.weight"] = (d0, 3, 3, 3, 3)
    s["encoder.conv1.bias"] = (d0,)
    for p, kind, cin, cout in encoder_layers(cfg):
        if kind == "res":
            res(p, cin, cout)
        else:
            s[p + "resample.1.weight"] = (cin, cin, 3, 3)
            s[p + "resample.1.bias"] = (cin,)
            if kind == "down3d":
                s[p + "time_conv.weight"] = (cin, cin, 3, 1, 1)
                s[p + "time_conv.bias"] = (cin,)
    res("encoder.middle.0.", top, top)
    attn("encoder.middle.1.", top)
    res("encoder.middle.2.", top, top)
    s["encoder.head.0.gamma"] = (top, 1, 1, 1)
    s["encoder.head.2.weight"] = (2 * cfg.z_dim, top, 3, 3, 3)
    s["encoder.head.2.bias"] = (2 * cfg.z_dim,)
    s["conv1.weight"] = (2 * cfg.z_dim, 2 * cfg.z_dim, 1, 1, 1)
    s["conv1.bias"] = (2 * cfg.z_dim,)
    s["conv2.weight"] = (cfg.z_dim, cfg.z_dim, 1, 1, 1)
    s["conv2.bias"] = (cfg.z_dim,)
    s["decoder.conv1.weight"] = (top, cfg.z_dim, 3, 3, 3)
    s["decoder.conv1.bias"] = (top,)
    res("decoder.middle.0.", top, top)
    attn("decoder.middle.1.", top)
    res("decoder.middle.2.", top, top)
    for p, kind, cin, cout in decoder_layers(cfg):
        if kind == "res":
            res(p, cin, cout)
        else:
            s[p + "resample.1.weight"] = (cout, cin, 3, 3)
            s[p + "resample.1.bias"] = (cout,)
            if kind == "up3d":
                s[p + "time_conv.weight"] = (2 * cin, cin, 3, 1, 1)
                s[p + "time_conv.bias"] = (2 * cin,)
    s["decoder.head.0.gamma"] = (cfg.dim, 1, 1, 1)
    s["decoder.head.2.weight"] = (3, cfg.dim, 3, 3, 3)
    s["decoder.head.2.bias"] = (3,)
    return s


def random_weights(cfg: Config, seed: int = 0) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, shp in state_dict_shapes(cfg).items():
        if k.endswith("gamma"):
            W[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            W[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for v in shp[1:]:
                fan_in *= v
            W[k] = torch.randn(shp, generator=g) * (1.5 / math.sqrt(fan_in))
    return W


# ------------------------------------------------------------------------------------------ primitives (x: [B,C,T,H,W])
def causal_conv3d(x: Tensor, w: Tensor, b: Tensor, stride=(1, 1, 1), cache: Optional[Tensor] = None) -> Tensor:
    """CausalConv3d.forward: time padding 2*pad_t in FRONT (zeros, or the cached last frames of the previous chunk)."""
    kt, kh, kw = w.shape[2:]
    pt = kt - 1                      # 2 * padding[0] with padding = (kt - 1) / 2 ... upstream passes padding=1 for kt=3, 0 for 1
    if cache is not None and pt > 0:
        x = torch.cat([cache, x], dim=2)
        pt -= cache.shape[2]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, max(pt, 0), 0))
    return F.conv3d(x, w, b, stride=stride)


def rms_norm(x: Tensor, gamma: Tensor) -> Tensor:
    """RMS_norm (channel_first): F.normalize(x, dim=1) * sqrt(C) * gamma."""
    c = x.shape[1]
    return F.normalize(x, dim=1) * (c ** 0.5) * gamma.reshape(1, c, *([1] * (x.dim() - 2)))


def attention_block(W, p: str, x: Tensor) -> Tensor:
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = rms_norm(y, W[p + "norm.gamma"])
    qkv = F.conv2d(y, W[p + "to_qkv.weight"], W[p + "to_qkv.bias"]).reshape(b * t, 1, 3 * c, h * w).permute(0, 1, 3, 2)
    q, k, v = qkv.chunk(3, dim=-1)
    y = F.scaled_dot_product_attention(q, k, v).squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
    y = F.conv2d(y, W[p + "proj.weight"], W[p + "proj.bias"])
    return y.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4) + x


def _spatial(x: Tensor, fn) -> Tensor:
    b, c, t, h, w = x.shape
    y = fn(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w))
    return y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def down_spatial(W, p, x):
    return _spatial(x, lambda u: F.conv2d(F.pad(u, (0, 1, 0, 1)), W[p + "resample.1.weight"], W[p + "resample.1.bias"], stride=2))


def up_spatial(W, p, x):
    return _spatial(x, lambda u: F.conv2d(F.interpolate(u.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").type_as(u),
                                          W[p + "resample.1.weight"], W[p + "resample.1.bias"], padding=1))


# ------------------------------------------------------------------------------------------ whole-sequence form
def res_block(W, p: str, x: Tensor) -> Tensor:
    h = causal_conv3d(x, W[p + "shortcut.weight"], W[p + "shortcut.bias"]) if (p + "shortcut.weight") in W else x
    y = F.silu(rms_norm(x, W[p + "residual.0.gamma"]))
    y = causal_conv3d(y, W[p + "residual.2.weight"], W[p + "residual.2.bias"])
    y = F.silu(rms_norm(y, W[p + "residual.3.gamma"]))
    y = causal_conv3d(y, W[p + "residual.6.weight"], W[p + "residual.6.bias"])
    return y + h


def down_temporal(W, p, x):
    """downsample3d after the spatial part: frame 0 passes; y_k = conv3(x_{2k-2}, x_{2k-1}, x_{2k}) for k >= 1."""
    if x.shape[2] == 1:
        return x
    rest = F.conv3d(x, W[p + "time_conv.weight"], W[p + "time_conv.bias"], stride=(2, 1, 1))     # windows start at 0, 2, 4 ...
    return torch.cat([x[:, :, :1], rest], dim=2)


def up_temporal(W, p, x):
    """upsample3d before the spatial part: frame 0 passes; frames 1.. go through a causal (3,1,1) conv to 2C channels whose
    history does NOT include frame 0, and each becomes two frames (channel halves interleaved in time)."""
    if x.shape[2] == 1:
        return x
    b, c, t, h, w = x.shape
    y = causal_conv3d(x[:, :, 1:], W[p + "time_conv.weight"], W[p + "time_conv.bias"])            # [b, 2c, t-1, h, w]
    y = y.reshape(b, 2, c, t - 1, h, w)
    y = torch.stack((y[:, 0], y[:, 1]), 3).reshape(b, c, 2 * (t - 1), h, w)
    return torch.cat([x[:, :, :1], y], dim=2)


def encode(W: Dict[str, Tensor], cfg: Config, x: Tensor, normalize: bool = True) -> Tensor:
    """x [B,3,F,H,W] in [-1,1], F = 1 + 4n  ->  mu [B,z,1+n,H/8,W/8] (the `.mode()` of the posterior)."""
    y = causal_conv3d(x, W["encoder.conv1.weight"], W["encoder.conv1.bias"])
    for p, kind, cin, cout in encoder_layers(cfg):
        if kind == "res":
            y = res_block(W, p, y)
        else:
            y = down_spatial(W, p, y)
            if kind == "down3d":
                y = down_temporal(W, p, y)
    y = res_block(W, "encoder.middle.0.", y)
    y = attention_block(W, "encoder.middle.1.", y)
    y = res_block(W, "encoder.middle.2.", y)
    y = F.silu(rms_norm(y, W["encoder.head.0.gamma"]))
    y = causal_conv3d(y, W["encoder.head.2.weight"], W["encoder.head.2.bias"])
    mu, _logvar = causal_conv3d(y, W["conv1.weight"], W["conv1.bias"]).chunk(2, dim=1)
    if normalize:
        mean = torch.tensor(LATENT_MEAN[:cfg.z_dim], dtype=mu.dtype).view(1, -1, 1, 1, 1)
        std = torch.tensor(LATENT_STD[:cfg.z_dim], dtype=mu.dtype).view(1, -1, 1, 1, 1)
        mu = (mu - mean) * (1.0 / std)
    return mu


def decode(W: Dict[str, Tensor], cfg: Config, z: Tensor, normalize: bool = True) -> Tensor:
    """z [B,z,T,h,w] -> video [B,3,1+4(T-1),8h,8w] clamped to [-1,1]."""
    if normalize:
        mean = torch.tensor(LATENT_MEAN[:cfg.z_dim], dtype=z.dtype).view(1, -1, 1, 1, 1)
        std = torch.tensor(LATENT_STD[:cfg.z_dim], dtype=z.dtype).view(1, -1, 1, 1, 1)
        z = z / (1.0 / std) + mean
    y = causal_conv3d(z, W["conv2.weight"], W["conv2.bias"])
    y = causal_conv3d(y, W["decoder.conv1.weight"], W["decoder.conv1.bias"])
    y = res_block(W, "decoder.middle.0.", y)
    y = attention_block(W, "decoder.middle.1.", y)
    y = res_block(W, "decoder.middle.2.", y)
    for p, kind, cin, cout in decoder_layers(cfg):
        if kind == "res":
            y = res_block(W, p, y)
        else:
            if kind == "up3d":
                y = up_temporal(W, p, y)
            y = up_spatial(W, p, y)
    y = F.silu(rms_norm(y, W["decoder.head.0.gamma"]))
    y = causal_conv3d(y, W["decoder.head.2.weight"], W["decoder.head.2.bias"])
    return y.clamp(-1, 1)


# ------------------------------------------------------------------------------------------ upstream execution order
class _Cache:
    def __init__(self):
        self.slots: Dict[int, object] = {}
        self.idx = 0


def _cached_conv(x, w, b, cache: _Cache):
    """ResidualBlock / Encoder3d / Decoder3d pattern: keep the last CACHE_T input frames of this conv for the next chunk."""
    i = cache.idx
    prev = cache.slots.get(i)
    keep = x[:, :, -CACHE_T:].clone()
    if keep.shape[2] < 2 and prev is not None:
        keep = torch.cat([prev[:, :, -1:], keep], dim=2)
    y = causal_conv3d(x, w, b, cache=prev)
    cache.slots[i] = keep
    cache.idx += 1
    return y


def _res_block_chunk(W, p, x, cache):
    h = causal_conv3d(x, W[p + "shortcut.weight"], W[p + "shortcut.bias"]) if (p + "shortcut.weight") in W else x
    y = F.silu(rms_norm(x, W[p + "residual.0.gamma"]))
    y = _cached_conv(y, W[p + "residual.2.weight"], W[p + "residual.2.bias"], cache)
    y = F.silu(rms_norm(y, W[p + "residual.3.gamma"]))
    y = _cached_conv(y, W[p + "residual.6.weight"], W[p + "residual.6.bias"], cache)
    return y + h


def _encoder_chunk(W, cfg, x, cache: _Cache):
    cache.idx = 0
    y = _cached_conv(x, W["encoder.conv1.weight"], W["encoder.conv1.bias"], cache)
    for p, kind, cin, cout in encoder_layers(cfg):
        if kind == "res":
            y = _res_block_chunk(W, p, y, cache)
        else:
            y = down_spatial(W, p, y)
            if kind == "down3d":
                i = cache.idx
                if cache.slots.get(i) is None:
                    cache.slots[i] = y.clone()                       # first chunk: no time conv
                else:
                    keep = y[:, :, -1:].clone()
                    y = F.conv3d(torch.cat([cache.slots[i][:, :, -1:], y], 2), W[p + "time_conv.weight"], W[p + "time_conv.bias"],
                                 stride=(2, 1, 1))
                    cache.slots[i] = keep
                cache.idx += 1
    y = _res_block_chunk(W, "encoder.middle.0.", y, cache)
    y = attention_block(W, "encoder.middle.1.", y)
    y = _res_block_chunk(W, "encoder.middle.2.", y, cache)
    y = F.silu(rms_norm(y, W["encoder.head.0.gamma"]))
    return _cached_conv(y, W["encoder.head.2.weight"], W["encoder.head.2.bias"], cache)


def encode_chunked(W, cfg: Config, x: Tensor, normalize: bool = True) -> Tensor:
    cache = _Cache()
    t = x.shape[2]
    outs = []
    for i in range(1 + (t - 1) // 4):
        chunk = x[:, :, :1] if i == 0 else x[:, :, 1 + 4 * (i - 1):1 + 4 * i]
        outs.append(_encoder_chunk(W, cfg, chunk, cache))
    y = torch.cat(outs, dim=2)
    mu, _ = causal_conv3d(y, W["conv1.weight"], W["conv1.bias"]).chunk(2, dim=1)
    if normalize:
        mean = torch.tensor(LATENT_MEAN[:cfg.z_dim], dtype=mu.dtype).view(1, -1, 1, 1, 1)
        std = torch.tensor(LATENT_STD[:cfg.z_dim], dtype=mu.dtype).view(1, -1, 1, 1, 1)
        mu = (mu - mean) * (1.0 / std)
    return mu


def _decoder_chunk(W, cfg, x, cache: _Cache):
    cache.idx = 0
    y = _cached_conv(x, W["decoder.conv1.weight"], W["decoder.conv1.bias"], cache)
    y = _res_block_chunk(W, "decoder.middle.0.", y, cache)
    y = attention_block(W, "decoder.middle.1.", y)
    y = _res_block_chunk(W, "decoder.middle.2.", y, cache)
    for p, kind, cin, cout in decoder_layers(cfg):
        if kind == "res":
            y = _res_block_chunk(W, p, y, cache)
            continue
        if kind == "up3d":
            i = cache.idx
            prev = cache.slots.get(i)
            if prev is None:
                cache.slots[i] = "Rep"                               # first chunk: frames are not doubled
            else:
                b, c, t, h, w = y.shape
                keep = y[:, :, -CACHE_T:].clone()
                if keep.shape[2] < 2 and not isinstance(prev, str):
                    keep = torch.cat([prev[:, :, -1:], keep], dim=2)
                if keep.shape[2] < 2 and isinstance(prev, str):
                    keep = torch.cat([torch.zeros_like(keep), keep], dim=2)
                tc = causal_conv3d(y, W[p + "time_conv.weight"], W[p + "time_conv.bias"],
                                   cache=None if isinstance(prev, str) else prev)
                cache.slots[i] = keep
                tc = tc.reshape(b, 2, c, t, h, w)
                y = torch.stack((tc[:, 0], tc[:, 1]), 3).reshape(b, c, 2 * t, h, w)
            cache.idx += 1
        y = up_spatial(W, p, y)
    y = F.silu(rms_norm(y, W["decoder.head.0.gamma"]))
    return _cached_conv(y, W["decoder.head.2.weight"], W["decoder.head.2.bias"], cache)


def decode_chunked(W, cfg: Config, z: Tensor, normalize: bool = True) -> Tensor:
    if normalize:
        mean = torch.tensor(LATENT_MEAN[:cfg.z_dim], dtype=z.dtype).view(1, -1, 1, 1, 1)
        std = torch.tensor(LATENT_STD[:cfg.z_dim], dtype=z.dtype).view(1, -1, 1, 1, 1)
        z = z / (1.0 / std) + mean
    x = causal_conv3d(z, W["conv2.weight"], W["conv2.bias"])
    cache = _Cache()
    outs = [_decoder_chunk(W, cfg, x[:, :, i:i + 1], cache) for i in range(z.shape[2])]
    return torch.cat(outs, dim=2).clamp(-1, 1)

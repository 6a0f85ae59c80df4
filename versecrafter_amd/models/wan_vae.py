"""Wan2.1 video VAE on the HIP engine: host-side mirror of `AutoencoderKLWan`, the class the reference's CLI builds
(inference/versecrafter_inference.py:220-236, `vae_kwargs` of config/wan2.1/wan_civitai.yaml:8-13) and its pipeline uses on
either side of the denoise loop:

    vae.encode(frames)[0].mode()          pipeline_wan_versecrafter.py:420, 432   (control videos -> control latents)
    vae.decode(latents).sample            pipeline_wan_versecrafter.py:551        (final latents -> frames in [-1, 1])
    vae.config.latent_channels / .spatial_compression_ratio / .temporal_compression_ratio, vae.latent_channels, vae.dtype

The class lives in the un-vendored `videox_fun.models` (origin: Wan2.1 wan/modules/vae.py) and its weights are not in the
reference tree: PARITY UNPINNED.  The arithmetic follows oracle/vae_oracle.py (a restatement of the published architecture);
parameter names are the upstream checkpoint's (`Wan2.1_VAE.pth`; VideoX-Fun's "model." prefix is accepted and dropped).

The module holds parameters only; encode / decode run in libvcengine (csrc/vae.hip).  There is no CPU path.
"""
import ctypes as C
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import _lib


def vae_state_dict_shapes(dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temporal_downsample=(False, True, True)):
    """Upstream key -> shape (the inventory csrc/vae.hip expects; mirrored by oracle/vae_oracle.state_dict_shapes)."""
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

    top = dim * dim_mult[-1]
    s["encoder.conv1.weight"] = (dim, 3, 3, 3, 3)
    s["encoder.conv1.bias"] = (dim,)
    dims = [dim * m for m in [1] + list(dim_mult)]
    idx = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(num_res_blocks):
            res(f"encoder.downsamples.{idx}.", cin, cout)
            cin = cout
            idx += 1
        if i != len(dim_mult) - 1:
            p = f"encoder.downsamples.{idx}."
            s[p + "resample.1.weight"] = (cout, cout, 3, 3)
            s[p + "resample.1.bias"] = (cout,)
            if temporal_downsample[i]:
                s[p + "time_conv.weight"] = (cout, cout, 3, 1, 1)
                s[p + "time_conv.bias"] = (cout,)
            idx += 1
    res("encoder.middle.0.", top, top)
    attn("encoder.middle.1.", top)
    res("encoder.middle.2.", top, top)
    s["encoder.head.0.gamma"] = (top, 1, 1, 1)
    s["encoder.head.2.weight"] = (2 * z_dim, top, 3, 3, 3)
    s["encoder.head.2.bias"] = (2 * z_dim,)
    s["conv1.weight"] = (2 * z_dim, 2 * z_dim, 1, 1, 1)
    s["conv1.bias"] = (2 * z_dim,)
    s["conv2.weight"] = (z_dim, z_dim, 1, 1, 1)
    s["conv2.bias"] = (z_dim,)
    s["decoder.conv1.weight"] = (top, z_dim, 3, 3, 3)
    s["decoder.conv1.bias"] = (top,)
    res("decoder.middle.0.", top, top)
    attn("decoder.middle.1.", top)
    res("decoder.middle.2.", top, top)
    ddims = [dim * m for m in [dim_mult[-1]] + list(dim_mult[::-1])]
    up = list(temporal_downsample[::-1])
    idx = 0
    for i, (cin, cout) in enumerate(zip(ddims[:-1], ddims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(num_res_blocks + 1):
            res(f"decoder.upsamples.{idx}.", cin, cout)
            cin = cout
            idx += 1
        if i != len(dim_mult) - 1:
            p = f"decoder.upsamples.{idx}."
            s[p + "resample.1.weight"] = (cout // 2, cout, 3, 3)
            s[p + "resample.1.bias"] = (cout // 2,)
            if up[i]:
                s[p + "time_conv.weight"] = (2 * cout, cout, 3, 1, 1)
                s[p + "time_conv.bias"] = (2 * cout,)
            idx += 1
    s["decoder.head.0.gamma"] = (dim, 1, 1, 1)
    s["decoder.head.2.weight"] = (3, dim, 3, 3, 3)
    s["decoder.head.2.bias"] = (3,)
    return s


class DiagonalGaussianDistribution:
    """What `vae.encode(x)[0]` / `.latent_dist` is upstream; the engine returns the mean only (the pipeline calls .mode())."""

    def __init__(self, mean):
        self.mean = mean

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        raise NotImplementedError("the HIP encoder returns the posterior mean (the reference's pipeline only calls .mode(), "
                                  "pipeline_wan_versecrafter.py:420)")


class AutoencoderKLOutput(tuple):
    def __new__(cls, dist):
        self = super().__new__(cls, (dist,))
        self.latent_dist = dist
        return self


class DecoderOutput(tuple):
    def __new__(cls, sample):
        self = super().__new__(cls, (sample,))
        self.sample = sample
        return self


class AutoencoderKLWan(nn.Module):
    def __init__(self, latent_channels=16, temporal_compression_ratio=4, spatial_compression_ratio=8, dim=96,
                 dim_mult=(1, 2, 4, 4), num_res_blocks=2, temporal_downsample=(False, True, True), param_device=None,
                 param_dtype=torch.bfloat16, **unused):
        super().__init__()
        if temporal_compression_ratio != 4 or spatial_compression_ratio != 8:
            raise ValueError("the Wan2.1 VAE compresses 4x in time and 8x in space (wan_civitai.yaml:11-12)")
        self.latent_channels = latent_channels
        self.temporal_compression_ratio, self.spatial_compression_ratio = temporal_compression_ratio, spatial_compression_ratio
        self.spacial_compression_ratio = spatial_compression_ratio                 # upstream spelling
        self.dim, self.dim_mult, self.num_res_blocks = dim, tuple(dim_mult), num_res_blocks
        self.temporal_downsample = tuple(bool(v) for v in temporal_downsample)
        self.config = SimpleNamespace(latent_channels=latent_channels, temporal_compression_ratio=temporal_compression_ratio,
                                      spatial_compression_ratio=spatial_compression_ratio)
        shapes = vae_state_dict_shapes(dim, latent_channels, self.dim_mult, num_res_blocks, self.temporal_downsample)
        for key, shape in shapes.items():
            holder = self
            parts = key.split(".")
            for part in parts[:-1]:
                if part not in holder._modules:
                    holder.add_module(part, nn.Module())
                holder = holder._modules[part]
            holder.register_parameter(parts[-1], nn.Parameter(torch.empty(shape, device=param_device, dtype=param_dtype),
                                                              requires_grad=False))
        self._engine = None
        self._loaded = {}

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def load_state_dict(self, state_dict, strict=True, assign=False):
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in state_dict.items()}
        return super().load_state_dict(sd, strict=strict, assign=assign)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, additional_kwargs=None, torch_dtype=torch.bfloat16):
        """CLI.py:220-223: `pretrained_model_path` is the checkpoint FILE (Wan2.1_VAE.pth, read with weights_only=True, or a
        .safetensors file); `additional_kwargs` = the yaml's vae_kwargs."""
        kw = dict(additional_kwargs or {})
        kw.pop("vae_subpath", None)
        if not os.path.isfile(pretrained_model_path):
            raise RuntimeError(f"{pretrained_model_path} is not a checkpoint file")
        if pretrained_model_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(pretrained_model_path)
        else:
            sd = torch.load(pretrained_model_path, map_location="cpu", weights_only=True, mmap=True)
        sd = sd.get("state_dict", sd)
        model = cls(param_dtype=torch_dtype, **kw)
        missing, unexpected = model.load_state_dict({k: v.to(torch_dtype) for k, v in sd.items()}, strict=False)
        if missing:
            raise RuntimeError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
        return model

    # ------------------------------------------------------------------ engine plumbing
    def _handle(self):
        if self._engine is None:
            lib = _lib.load()
            cfg = _lib.vc_vae_config()
            cfg.dim, cfg.z_dim, cfg.num_res_blocks = self.dim, self.latent_channels, self.num_res_blocks
            for i in range(4):
                cfg.dim_mult[i] = self.dim_mult[i]
            for i in range(3):
                cfg.temporal_downsample[i] = int(self.temporal_downsample[i])
            h = C.c_void_p()
            rc = lib.vc_vae_create(C.byref(cfg), C.byref(h))
            if rc != 0:
                raise ValueError("vc_vae_create: " + (lib.vc_vae_last_error(None) or b"").decode())
            self._engine = h
        return self._engine

    def _sync(self):
        lib, h = _lib.load(), self._handle()
        for key, p in self.named_parameters():
            if not p.is_cuda:
                raise RuntimeError(f"parameter {key} is on {p.device}: move the VAE to the GPU (versecrafter_amd has no CPU path)")
            if p.dtype != torch.bfloat16:
                raise TypeError(f"parameter {key} is {p.dtype}; the VAE computes in bf16 (CLI.py:223: .to(weight_dtype))")
            if not p.is_contiguous():
                p.data = p.data.contiguous()
            tag = (p.data_ptr(), p._version)
            if self._loaded.get(key) != tag:
                shape = (C.c_int64 * p.dim())(*p.shape)
                rc = lib.vc_vae_load_weight(h, key.encode(), C.c_void_p(p.data_ptr()), p.dim(), shape)
                if rc != 0:
                    raise RuntimeError("vc_vae_load_weight: " + (lib.vc_vae_last_error(h) or b"").decode())
                self._loaded[key] = tag
        return h

    @staticmethod
    def _raise(lib, h, rc, what):
        msg = (lib.vc_vae_last_error(h) or b"").decode()
        if rc == _lib.VC_E_INVALID:
            raise ValueError(f"{what}: {msg}")
        raise _lib.VcError(rc, f"{what}: {msg}")

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        """x [B, 3, F, H, W] in [-1, 1] (F = 1 + 4n) -> output whose [0] / .latent_dist has .mode() = [B, 16, 1 + n, H/8, W/8]."""
        if not x.is_cuda:
            raise RuntimeError("x must be a CUDA (HIP) tensor: versecrafter_amd has no CPU path")
        if x.dim() != 5 or x.shape[1] != 3:
            raise ValueError(f"expected [B, 3, F, H, W], got {tuple(x.shape)}")
        lib, h = _lib.load(), self._sync()
        B, _, F, H, W = x.shape
        xb = x.to(torch.bfloat16).contiguous()
        out = torch.empty(B, self.latent_channels, 1 + (F - 1) // 4, H // 8, W // 8, dtype=torch.bfloat16, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        with torch.cuda.device(x.device):
            for b in range(B):
                rc = lib.vc_vae_encode(h, C.c_void_p(xb[b].data_ptr()), C.c_void_p(out[b].data_ptr()), F, H, W, stream)
                if rc != 0:
                    self._raise(lib, h, rc, "vc_vae_encode")
        return AutoencoderKLOutput(DiagonalGaussianDistribution(out.to(x.dtype) if x.dtype.is_floating_point else out))

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        """z [B, 16, T, h, w] -> output with .sample = [B, 3, 1 + 4 (T - 1), 8h, 8w] in [-1, 1]."""
        if not z.is_cuda:
            raise RuntimeError("z must be a CUDA (HIP) tensor: versecrafter_amd has no CPU path")
        if z.dim() != 5 or z.shape[1] != self.latent_channels:
            raise ValueError(f"expected [B, {self.latent_channels}, T, h, w], got {tuple(z.shape)}")
        lib, h = _lib.load(), self._sync()
        B, _, T, hh, ww = z.shape
        zb = z.to(torch.bfloat16).contiguous()
        out = torch.empty(B, 3, 1 + 4 * (T - 1), 8 * hh, 8 * ww, dtype=torch.bfloat16, device=z.device)
        stream = C.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
        with torch.cuda.device(z.device):
            for b in range(B):
                rc = lib.vc_vae_decode(h, C.c_void_p(zb[b].data_ptr()), C.c_void_p(out[b].data_ptr()), T, hh, ww, stream)
                if rc != 0:
                    self._raise(lib, h, rc, "vc_vae_decode")
        return DecoderOutput(out.to(z.dtype))

    def set_time_chunk(self, frames: int = -1):
        """Frames per time chunk of the full-resolution stage (vc_vae_set_time_chunk): -1 automatic (chunks of 8 when the
        whole-sequence workspace would exceed 40 GB), 0 never, n > 0 always.  Results are bit-identical either way."""
        _lib.check(_lib.load().vc_vae_set_time_chunk(self._sync(), int(frames)))
        return self

    def last_time_chunk(self) -> int:
        return 0 if self._engine is None else int(_lib.load().vc_vae_last_time_chunk(self._engine))

    def workspace_bytes(self):
        return 0 if self._engine is None else int(_lib.load().vc_vae_workspace_bytes(self._engine))

    def release_workspace(self):
        """Give the engine's activation buffers (tens of GB at production sizes) back; the next encode / decode re-allocates."""
        if self._engine is not None:
            _lib.load().vc_vae_release_workspace(self._engine)

    def __del__(self):
        try:
            if self._engine is not None:
                _lib.load().vc_vae_destroy(self._engine)
                self._engine = None
        except Exception:
            pass

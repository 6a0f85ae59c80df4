"""umT5 text encoder on the HIP engine: host-side mirror of `WanT5EncoderModel`, the class the reference's CLI builds
(inference/versecrafter_inference.py:243-249) and its pipeline calls once per prompt
(pipeline_wan_versecrafter.py:221-282: `self.text_encoder(ids, attention_mask=mask)[0]`, `self.text_encoder.dtype`).

The class is imported by the reference from the un-vendored `videox_fun.models` (origin: Wan2.1 wan/modules/t5.py);
its source is not in the reference tree.  Constructor keywords follow config/wan2.1/wan_civitai.yaml:14-26
(`vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, shared_pos, dropout`), parameter names follow
the upstream checkpoint `models_t5_umt5-xxl-enc-bf16.pth` (token_embedding / blocks.N.{norm1,attn.{q,k,v,o},norm2,
ffn.{gate.0,fc1,fc2},pos_embedding.embedding} / norm).  The arithmetic is pinned against transformers'
UMT5EncoderModel (same published architecture) -- see oracle/t5_oracle.py and tests/test_t5_*.py.

The module holds parameters only; forward runs in libvcengine (csrc/t5.hip).  There is no CPU path.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from .. import _lib

# transformers / HF key -> upstream Wan key (for checkpoints in the HF umT5 layout)
_HF_LAYER = {
    "layer.0.layer_norm.weight": "norm1.weight",
    "layer.0.SelfAttention.q.weight": "attn.q.weight",
    "layer.0.SelfAttention.k.weight": "attn.k.weight",
    "layer.0.SelfAttention.v.weight": "attn.v.weight",
    "layer.0.SelfAttention.o.weight": "attn.o.weight",
    "layer.0.SelfAttention.relative_attention_bias.weight": "pos_embedding.embedding.weight",
    "layer.1.layer_norm.weight": "norm2.weight",
    "layer.1.DenseReluDense.wi_0.weight": "ffn.gate.0.weight",
    "layer.1.DenseReluDense.wi_1.weight": "ffn.fc1.weight",
    "layer.1.DenseReluDense.wo.weight": "ffn.fc2.weight",
}


def convert_hf_umt5_state_dict(sd):
    """transformers UMT5EncoderModel state dict -> the upstream Wan key names this class uses."""
    out = {}
    for k, v in sd.items():
        if k in ("shared.weight", "encoder.embed_tokens.weight"):
            out["token_embedding.weight"] = v
        elif k == "encoder.final_layer_norm.weight":
            out["norm.weight"] = v
        elif k.startswith("encoder.block."):
            n, rest = k[len("encoder.block."):].split(".", 1)
            if rest in _HF_LAYER:
                out[f"blocks.{n}.{_HF_LAYER[rest]}"] = v
    return out


class WanT5EncoderModel(nn.Module):
    def __init__(self, vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32,
                 shared_pos=False, dropout=0.0, max_distance=128, eps=1e-6, param_device=None,
                 param_dtype=torch.bfloat16, **unused):
        super().__init__()
        if shared_pos:
            raise NotImplementedError("shared_pos=True is not the umT5 configuration (wan_civitai.yaml:24)")
        if dim_attn != num_heads * 64:
            raise NotImplementedError(f"head dim {dim_attn // num_heads}: the HIP encoder is built for 64 (umT5-XXL: 64 x 64)")
        self.vocab, self.dim, self.dim_attn, self.dim_ffn = vocab, dim, dim_attn, dim_ffn
        self.num_heads, self.num_layers, self.num_buckets, self.max_distance, self.eps = (num_heads, num_layers, num_buckets,
                                                                                          max_distance, eps)
        kw = dict(device=param_device, dtype=param_dtype)
        P = lambda *shape: nn.Parameter(torch.empty(*shape, **kw), requires_grad=False)

        def add(key, *shape):
            # parameters are registered under their upstream dotted names (state_dict keys == checkpoint keys)
            holder = self
            parts = key.split(".")
            for part in parts[:-1]:
                if part not in holder._modules:
                    holder.add_module(part, nn.Module())
                holder = holder._modules[part]
            holder.register_parameter(parts[-1], P(*shape))
        add("token_embedding.weight", vocab, dim)
        for i in range(num_layers):
            p = f"blocks.{i}."
            add(p + "norm1.weight", dim)
            for n in "qkv":
                add(p + f"attn.{n}.weight", dim_attn, dim)
            add(p + "attn.o.weight", dim, dim_attn)
            add(p + "norm2.weight", dim)
            add(p + "ffn.gate.0.weight", dim_ffn, dim)
            add(p + "ffn.fc1.weight", dim_ffn, dim)
            add(p + "ffn.fc2.weight", dim, dim_ffn)
            add(p + "pos_embedding.embedding.weight", num_buckets, num_heads)
        add("norm.weight", dim)
        self._engine = None
        self._loaded = {}

    # ------------------------------------------------------------------ reference-facing surface
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_pretrained(cls, pretrained_model_path, additional_kwargs=None, low_cpu_mem_usage=False,
                        torch_dtype=torch.bfloat16):
        """CLI.py:243-247.  `pretrained_model_path`: a .safetensors file, or a .pth / .pt state dict (loaded with
        weights_only=True), in the upstream key layout or the transformers umT5 layout."""
        kw = dict(additional_kwargs or {})
        for k in ("text_encoder_subpath", "tokenizer_subpath", "text_length"):
            kw.pop(k, None)
        if not os.path.isfile(pretrained_model_path):
            raise RuntimeError(f"{pretrained_model_path} is not a checkpoint file")
        if pretrained_model_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(pretrained_model_path)
        else:
            sd = torch.load(pretrained_model_path, map_location="cpu", weights_only=True, mmap=True)
        if any(k.startswith("encoder.block.") for k in sd):
            sd = convert_hf_umt5_state_dict(sd)
        model = cls(param_device="meta" if low_cpu_mem_usage else None, param_dtype=torch_dtype, **kw)
        missing, unexpected = model.load_state_dict({k: v.to(torch_dtype) for k, v in sd.items()}, strict=False,
                                                    assign=low_cpu_mem_usage)
        if missing:
            raise RuntimeError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
        return model

    # ------------------------------------------------------------------ engine plumbing
    def _handle(self):
        if self._engine is None:
            lib = _lib.load()
            cfg = _lib.vc_t5_config(self.vocab, self.dim, self.dim_attn, self.dim_ffn, self.num_heads, self.num_layers,
                                    self.num_buckets, self.max_distance, float(self.eps))
            h = C.c_void_p()
            rc = lib.vc_t5_create(C.byref(cfg), C.byref(h))
            if rc != 0:
                raise ValueError("vc_t5_create: " + (lib.vc_t5_last_error(None) or b"").decode())
            self._engine = h
        return self._engine

    def _sync(self):
        lib, h = _lib.load(), self._handle()
        for key, p in self.named_parameters():
            if not p.is_cuda:
                raise RuntimeError(f"parameter {key} is on {p.device}: move the encoder to the GPU "
                                   "(versecrafter_amd has no CPU path)")
            if p.dtype != torch.bfloat16:
                raise TypeError(f"parameter {key} is {p.dtype}; the encoder computes in bf16")
            if not p.is_contiguous():
                p.data = p.data.contiguous()
            tag = (p.data_ptr(), p._version)
            if self._loaded.get(key) != tag:
                shape = (C.c_int64 * p.dim())(*p.shape)
                rc = lib.vc_t5_load_weight(h, key.encode(), C.c_void_p(p.data_ptr()), p.dim(), shape)
                if rc != 0:
                    raise RuntimeError("vc_t5_load_weight: " + (lib.vc_t5_last_error(h) or b"").decode())
                self._loaded[key] = tag

    @torch.no_grad()
    def forward(self, ids, attention_mask=None, **unused):
        """ids [B, L] integer tokens, attention_mask [B, L] (1 = token) -> ([B, L, dim] bf16,)"""
        if not ids.is_cuda:
            raise RuntimeError("ids must be a CUDA (HIP) tensor: versecrafter_amd has no CPU path")
        if ids.dim() != 2:
            raise ValueError("ids must be [B, L]")
        B, L = ids.shape
        lib = _lib.load()
        self._sync()
        h = self._engine
        ids32 = ids.to(torch.int32).contiguous()
        m32 = None if attention_mask is None else attention_mask.to(device=ids.device, dtype=torch.int32).contiguous()
        out = torch.empty(B, L, self.dim, dtype=torch.bfloat16, device=ids.device)
        stream = C.c_void_p(torch.cuda.current_stream(ids.device).cuda_stream)
        with torch.cuda.device(ids.device):
            rc = lib.vc_t5_encode(h, C.c_void_p(ids32.data_ptr()), C.c_void_p(0 if m32 is None else m32.data_ptr()),
                                  C.c_void_p(out.data_ptr()), B, L, stream)
        if rc != 0:
            msg = (lib.vc_t5_last_error(h) or b"").decode()
            if rc == _lib.VC_E_INVALID:
                raise ValueError("vc_t5_encode: " + msg)
            raise _lib.VcError(rc, msg)
        return (out,)

    def workspace_bytes(self):
        return 0 if self._engine is None else _lib.load().vc_t5_workspace_bytes(self._engine)

    def __del__(self):
        try:
            if self._engine is not None:
                _lib.load().vc_t5_destroy(self._engine)
                self._engine = None
        except Exception:
            pass

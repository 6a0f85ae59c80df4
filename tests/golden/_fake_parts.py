"""Deterministic stand-ins for the per-video models around the denoise loop (VAE, tokenizer, T5, DiT), used on BOTH sides
of the sampler-pipeline parity check: tests/golden/make_golden_pipe.py drives the REFERENCE's own
WanVerseCrafterPipeline.__call__ with them and records what the DiT is called with; tests/test_pipeline_golden.py drives
versecrafter_amd's pipeline with the same parts and must reproduce the record.  They are plain closed-form functions of their
inputs (no weights), so the fixtures pin the pipeline's own arithmetic: control-map / mask pre-processing, geoada_context
assembly, seq_len, CFG batch order, timestep broadcast, guidance combine, scheduler hand-over."""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F


class _Dist:
    def __init__(self, z):
        self._z = z

    def mode(self):
        return self._z


class FakeVAE:
    """encode(frames [B,3,F,H,W]) -> [dist]; dist.mode() [B,16,(F-1)//4+1,H/8,W/8]: frame 0, then the mean of each group of
    four frames, 8x8 average pooling, a fixed 3 -> 16 channel mix."""
    dtype = torch.float32
    latent_channels = 16
    temporal_compression_ratio = 4
    spatial_compression_ratio = 8
    config = SimpleNamespace(latent_channels=16, temporal_compression_ratio=4, spatial_compression_ratio=8)

    def __init__(self):
        i = torch.arange(16, dtype=torch.float32)[:, None]
        j = torch.arange(3, dtype=torch.float32)[None, :]
        self.mix = torch.sin(0.7 * i + 1.3 * j + 0.2)

    def encode(self, frames):
        frames = frames.float()
        b, c, f, h, w = frames.shape
        groups = [frames[:, :, :1]] + [frames[:, :, 1 + 4 * k:5 + 4 * k].mean(dim=2, keepdim=True) for k in range((f - 1) // 4)]
        t = torch.cat(groups, dim=2)
        t = F.avg_pool3d(t, kernel_size=(1, 8, 8))
        z = torch.einsum("oc,bcthw->bothw", self.mix, t)
        return [_Dist(z)]


class FakeTokenizer:
    """One token per character (capped), 1-padded mask; enough for _get_t5_prompt_embeds (PIPE.py:243-273)."""

    def __call__(self, prompt, padding=None, max_length=None, truncation=None, add_special_tokens=None, return_tensors=None):
        prompt = [prompt] if isinstance(prompt, str) else prompt
        longest = max(len(p) + 1 for p in prompt)
        width = max_length if padding == "max_length" else longest
        ids = torch.zeros(len(prompt), width, dtype=torch.long)
        mask = torch.zeros(len(prompt), width, dtype=torch.long)
        for r, p in enumerate(prompt):
            toks = [3 + (ord(ch) % 97) for ch in p][: width - 1] + [1]
            ids[r, :len(toks)] = torch.tensor(toks)
            mask[r, :len(toks)] = 1
        return SimpleNamespace(input_ids=ids, attention_mask=mask)

    def batch_decode(self, ids):
        return ["" for _ in ids]


class FakeTextEncoder:
    dtype = torch.float32

    def __init__(self, dim=64):
        self.dim = dim

    def __call__(self, ids, attention_mask=None):
        pos = torch.arange(ids.shape[1], dtype=torch.float32)[None, :, None]
        ch = torch.arange(self.dim, dtype=torch.float32)[None, None, :]
        emb = torch.sin(ids[..., None].float() * 0.37 + ch * 0.11 + pos * 0.05)
        return (emb,)

    def to(self, *a, **k):
        return self


class FakeTransformer:
    """Records every call; the 'noise prediction' is a closed-form function of ALL its inputs so that any mix-up of batch
    order, timestep, control maps, prompt or scale changes the sampler trajectory."""
    config = SimpleNamespace(patch_size=(1, 2, 2))

    def __init__(self):
        self.calls = []
        self.num_inference_steps = None
        self.current_steps = 0

    def parameters(self):
        yield torch.zeros(1)

    def __call__(self, x, context, t, geoada_context, seq_len, geoada_context_scale=1.0):
        g = torch.stack(list(geoada_context)) if isinstance(geoada_context, (list, tuple)) else geoada_context
        self.calls.append(dict(x=x.detach().clone().float(), t=t.detach().clone().float(), geoada=g.detach().clone().float(),
                               ctx_lens=[int(u.shape[0]) for u in context],
                               ctx_sums=[float(u.float().sum()) for u in context], seq_len=int(seq_len),
                               scale=float(geoada_context_scale), step=int(self.current_steps)))
        c = torch.stack([torch.tanh(u.float().mean() * 5.0) for u in context]).view(-1, 1, 1, 1, 1)
        tt = (t.float() / 1000.0).view(-1, 1, 1, 1, 1)
        gg = g.float()
        y = 0.3 * x.float() + 0.05 * geoada_context_scale * (gg[:, :16] - gg[:, 64:80]) + 0.1 * c + 0.02 * tt * x.float().roll(1, 2)
        return y.to(x.dtype)

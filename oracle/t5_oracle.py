"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain PyTorch fp32) of the umT5 encoder the reference's pipeline calls as
`self.text_encoder(ids, attention_mask=mask)[0]` (versecrafter/pipeline/pipeline_wan_versecrafter.py:273; class
WanT5EncoderModel built at inference/versecrafter_inference.py:243-249 from config/wan2.1/wan_civitai.yaml:14-26).

The class lives in the un-vendored videox_fun package (origin: Wan2.1 wan/modules/t5.py), so this restates the published
umT5 / T5 v1.1 encoder.  PINNED against transformers' UMT5EncoderModel (an independent implementation of the same
architecture, importable in the build container): tests/golden/make_golden_t5.py records its outputs on seeded inputs
into tests/golden/t5_tiny.safetensors and tests/test_t5_oracle.py checks this file against them.  Parity with the
reference's own class is UNPINNED (its source is absent); key names follow the upstream checkpoint.

Nothing under versecrafter_amd/ imports this module.
"""
import math

import torch
import torch.nn.functional as F


def relative_position_bucket(rel: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """Bidirectional T5 bucket of rel = key - query."""
    nb = num_buckets // 2
    ret = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(n < max_exact, n, large)


def t5_layer_norm(x, w, eps=1e-6):
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def gelu_tanh(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def encode(W, ids, mask, num_heads, num_buckets=32, max_distance=128, eps=1e-6):
    """W: state dict in the upstream key layout (fp32).  ids [B, L] long, mask [B, L] (1 = token) or None -> [B, L, dim]."""
    B, L = ids.shape
    x = W["token_embedding.weight"][ids]
    n_layers = 1 + max(int(k.split(".")[1]) for k in W if k.startswith("blocks."))
    pos = torch.arange(L)
    bucket = relative_position_bucket(pos[None, :] - pos[:, None], num_buckets, max_distance)       # [Lq, Lk]
    neg = torch.zeros(B, 1, 1, L)
    if mask is not None:
        neg = neg.masked_fill(mask[:, None, None, :] == 0, torch.finfo(torch.float32).min)
    for i in range(n_layers):
        p = f"blocks.{i}."
        t = t5_layer_norm(x, W[p + "norm1.weight"], eps)
        q, k, v = (F.linear(t, W[p + f"attn.{n}.weight"]).view(B, L, num_heads, -1).transpose(1, 2) for n in "qkv")
        bias = W[p + "pos_embedding.embedding.weight"][bucket].permute(2, 0, 1)[None]                # [1, N, Lq, Lk]
        s = q @ k.transpose(-1, -2) + bias + neg                                                    # no 1/sqrt(d) scaling
        a = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, L, -1)
        x = x + F.linear(a, W[p + "attn.o.weight"])
        t = t5_layer_norm(x, W[p + "norm2.weight"], eps)
        x = x + F.linear(gelu_tanh(F.linear(t, W[p + "ffn.gate.0.weight"])) * F.linear(t, W[p + "ffn.fc1.weight"]),
                         W[p + "ffn.fc2.weight"])
    return t5_layer_norm(x, W["norm.weight"], eps)


def random_weights(vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
    W = {"token_embedding.weight": r(vocab, dim), "norm.weight": 1.0 + 0.1 * r(dim)}
    for i in range(num_layers):
        p = f"blocks.{i}."
        W[p + "norm1.weight"] = 1.0 + 0.1 * r(dim)
        W[p + "norm2.weight"] = 1.0 + 0.1 * r(dim)
        for n in "qkv":
            W[p + f"attn.{n}.weight"] = r(dim_attn, dim, scale=dim ** -0.5 * (0.35 if n != "v" else 1.0))
        W[p + "attn.o.weight"] = r(dim, dim_attn, scale=dim_attn ** -0.5)
        W[p + "ffn.gate.0.weight"] = r(dim_ffn, dim, scale=dim ** -0.5)
        W[p + "ffn.fc1.weight"] = r(dim_ffn, dim, scale=dim ** -0.5)
        W[p + "ffn.fc2.weight"] = r(dim, dim_ffn, scale=dim_ffn ** -0.5)
        W[p + "pos_embedding.embedding.weight"] = r(num_buckets, num_heads)
    return W

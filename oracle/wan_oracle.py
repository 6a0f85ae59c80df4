"""CPU ORACLE for the VerseCrafter denoise hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional PyTorch (CPU, fp32) restatement of the reference's per-step
Wan2.1-DiT + GeoAdapter forward.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module, and only as the checker / the timed CPU baseline.
The product path (versecrafter_amd/) never imports it and has no CPU fallback.

Every function cites the reference file:line it follows; abbreviations:
  WT.py   = /root/reference/versecrafter/models/wan_transformer3d.py
  VC.py   = /root/reference/versecrafter/models/wan_transformer3d_versecrafter.py
  PIPE.py = /root/reference/versecrafter/pipeline/pipeline_wan_versecrafter.py

Pinning: the reference holds no tests or golden vectors.  This oracle is
pinned against outputs of the reference's own WT.py/VC.py executed in the build container
(tests/golden/make_golden.py -> tests/golden/*.safetensors; tests/test_oracle_golden.py).
Third-party arithmetic that is NOT under /root/reference (un-vendored submodule
third_party/VideoX-Fun @ unknown commit: `attention`, `TeaCache`, `FlowUniPCMultistepScheduler`,
dist `usp_attn_forward`) is restated from the call sites and is "parity unpinned":
  * attention()            : softmax(q k^T / sqrt(D)) v, keys >= k_lens masked.
  * TeaCache               : rel-L1 gate, np.poly1d rescale (WT.py:205-245 is the pinned part).
  * FlowUniPCMultistepScheduler : oracle/unipc_oracle.py.

Precision modes
---------------
`mode="fp32"` computes everything in fp32 (the mathematical reference).
`mode="bf16"` rounds to bfloat16 at the points where the reference, run under
`torch.cuda.amp.autocast(dtype=bfloat16)` with bf16 weights (PIPE.py:893, CLI weight_dtype),
materialises a bf16 tensor: every nn.Linear / Conv3d output, LayerNorm output (WT.py:339-346),
RMSNorm (stats in fp32, WT.py:323), each elementwise modulate/gate op, RoPE output
(WT.py:172), attention output, e and e0 (VC.py:353-354).  It is used to size test tolerances:
a HIP kernel's distance to the fp32 result must be of the order of the bf16 reference's own.
"""
import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _rounder(mode: str):
    if mode == "fp32":
        return lambda t: t
    if mode == "bf16":
        return lambda t: t.to(torch.bfloat16).to(torch.float32)
    raise ValueError(mode)


# --------------------------------------------------------------------------------------
# embeddings
# --------------------------------------------------------------------------------------
def sinusoidal_embedding_1d(dim: int, position: Tensor) -> Tensor:
    """WT.py:39-49.  fp64 [B, dim] = cat[cos(t*w), sin(t*w)], w_i = 10000^(-i/half)."""
    assert dim % 2 == 0
    half = dim // 2
    position = position.to(torch.float64)
    w = torch.pow(torch.tensor(10000.0, dtype=torch.float64),
                  -torch.arange(half, dtype=torch.float64) / half)
    s = torch.outer(position, w)
    return torch.cat([torch.cos(s), torch.sin(s)], dim=1)


def rope_params(max_seq_len: int, dim: int, theta: float = 10000.0) -> Tensor:
    """WT.py:52-60.  complex128 [max_seq_len, dim/2] = exp(i * p * theta^(-2j/dim))."""
    assert dim % 2 == 0
    inv = 1.0 / torch.pow(torch.tensor(theta, dtype=torch.float64),
                          torch.arange(0, dim, 2, dtype=torch.float64) / dim)
    ang = torch.outer(torch.arange(max_seq_len, dtype=torch.float64), inv)
    return torch.polar(torch.ones_like(ang), ang)


def rope_table(head_dim: int, max_seq_len: int = 1024) -> Tensor:
    """WT.py:785-795.  complex128 [1024, head_dim/2]; columns [d-4(d//6) | 2(d//6) | 2(d//6)]/2."""
    d = head_dim
    return torch.cat([rope_params(max_seq_len, d - 4 * (d // 6)),
                      rope_params(max_seq_len, 2 * (d // 6)),
                      rope_params(max_seq_len, 2 * (d // 6))], dim=1)


def rope_table_riflex(head_dim: int, k: int, L_test: int, L_test_scale: Optional[float] = None,
                      max_seq_len: int = 1024) -> Tensor:
    """WT.py:63-121, 873-888: temporal frequency k-1 replaced by 0.9*2pi/L_test (/scale)."""
    d = head_dim
    dt = d - 4 * (d // 6)
    inv = 1.0 / torch.pow(torch.tensor(10000.0, dtype=torch.float64),
                          torch.arange(0, dt, 2, dtype=torch.float64) / dt)
    inv[k - 1] = 0.9 * 2 * math.pi / L_test
    if L_test_scale is not None:
        inv[k - 1] = inv[k - 1] / L_test_scale
    ang = torch.outer(torch.arange(max_seq_len, dtype=torch.float64), inv)
    t = torch.polar(torch.ones_like(ang), ang)
    return torch.cat([t, rope_params(max_seq_len, 2 * (d // 6)),
                      rope_params(max_seq_len, 2 * (d // 6))], dim=1)


def rope_apply(x: Tensor, grid_sizes: Sequence[Sequence[int]], freqs: Tensor,
               token_offset: int = 0, mode: str = "fp32") -> Tensor:
    """WT.py:143-172.  x [B, L, N, D]; rotates adjacent pairs (2j, 2j+1) of each head by the
    per-token (f, h, w) entry of the cis table; tokens >= f*h*w are passed through.

    token_offset: global index of x[:, 0] under sequence parallelism (the third-party dist
    rope offsets the table lookup by the rank's chunk start).
    """
    rnd = _rounder(mode)
    B, L, N, D = x.shape
    c = D // 2
    ft, fh, fw = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    out = []
    for i, (f, h, w) in enumerate(grid_sizes):
        seq_len = f * h * w
        fi = torch.cat([ft[:f].view(f, 1, 1, -1).expand(f, h, w, -1),
                        fh[:h].view(1, h, 1, -1).expand(f, h, w, -1),
                        fw[:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(seq_len, 1, -1)
        lo = token_offset
        hi = min(token_offset + L, seq_len)
        n_valid = max(hi - lo, 0)
        xi = torch.view_as_complex(x[i, :n_valid].to(torch.float32).reshape(n_valid, N, -1, 2))
        yi = torch.view_as_real(xi * fi[lo:lo + n_valid]).flatten(2)   # complex128 product
        yi = torch.cat([yi, x[i, n_valid:].to(yi.dtype)])
        out.append(yi)
    return rnd(torch.stack(out).to(torch.float32))


# --------------------------------------------------------------------------------------
# norms
# --------------------------------------------------------------------------------------
def rms_norm(x: Tensor, weight: Tensor, eps: float = 1e-6, mode: str = "fp32") -> Tensor:
    """WT.py:307-323.  x * rsqrt(mean(x^2) + eps) * w, reduction over the FULL last dim.
    bf16 mode: stats fp32, rsqrt cast to bf16, then two bf16 multiplies (WT.py:320-323)."""
    rnd = _rounder(mode)
    inv = rnd(torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps))
    return rnd(rnd(x * inv) * weight)


def layer_norm(x: Tensor, weight: Optional[Tensor] = None, bias: Optional[Tensor] = None,
               eps: float = 1e-6, mode: str = "fp32") -> Tensor:
    """WT.py:326-346.  fp32 LayerNorm (biased variance), result cast back to x.dtype."""
    rnd = _rounder(mode)
    y = F.layer_norm(x.float(), (x.shape[-1],),
                     None if weight is None else weight.float(),
                     None if bias is None else bias.float(), eps)
    return rnd(y)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor], mode: str = "fp32") -> Tensor:
    """nn.Linear under autocast: fp32 accumulate, output rounded once."""
    return _rounder(mode)(F.linear(x, w, b))


def gelu_tanh(x: Tensor, mode: str = "fp32") -> Tensor:
    """nn.GELU(approximate='tanh') (WT.py:558, 761)."""
    return _rounder(mode)(F.gelu(x, approximate="tanh"))


# --------------------------------------------------------------------------------------
# attention (third-party contract; parity unpinned)
# --------------------------------------------------------------------------------------
def attention(q: Tensor, k: Tensor, v: Tensor, k_lens: Optional[Sequence[int]] = None,
              mode: str = "fp32") -> Tensor:
    """videox_fun.models.attention_utils.attention as called at WT.py:394-399, 425-430.
    [B, L, N, D] in/out; softmax(q k^T / sqrt(D)) v, non-causal; keys >= k_lens[b] masked."""
    rnd = _rounder(mode)
    B, Lq, N, D = q.shape
    Lk = k.shape[1]
    qh, kh, vh = (t.transpose(1, 2).float() for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / math.sqrt(D)
    if k_lens is not None:
        mask = torch.arange(Lk)[None, :] >= torch.as_tensor(list(k_lens))[:, None]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    o = torch.softmax(s, dim=-1) @ vh
    return rnd(o.transpose(1, 2).contiguous())


# --------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------
def self_attention(W: Dict[str, Tensor], p: str, x: Tensor, seq_lens, grid_sizes, freqs,
                   num_heads: int, eps: float, token_offset: int = 0, mode: str = "fp32",
                   attn_fn=None) -> Tensor:
    """WT.py:373-405.  q,k,v Linear -> full-dim RMSNorm(q,k) -> RoPE -> attention -> o."""
    B, L, C = x.shape
    n, d = num_heads, C // num_heads
    q = rms_norm(linear(x, W[p + "q.weight"], W[p + "q.bias"], mode), W[p + "norm_q.weight"], eps, mode)
    k = rms_norm(linear(x, W[p + "k.weight"], W[p + "k.bias"], mode), W[p + "norm_k.weight"], eps, mode)
    v = linear(x, W[p + "v.weight"], W[p + "v.bias"], mode)
    q = rope_apply(q.view(B, L, n, d), grid_sizes, freqs, token_offset, mode)
    k = rope_apply(k.view(B, L, n, d), grid_sizes, freqs, token_offset, mode)
    v = v.view(B, L, n, d)
    if attn_fn is None:
        o = attention(q, k, v, k_lens=seq_lens, mode=mode)
    else:                                   # sequence-parallel tests inject the Ulysses exchange
        o = attn_fn(q, k, v, seq_lens)
    return linear(o.flatten(2), W[p + "o.weight"], W[p + "o.bias"], mode)


def cross_attention(W: Dict[str, Tensor], p: str, x: Tensor, context: Tensor, num_heads: int,
                    eps: float, mode: str = "fp32") -> Tensor:
    """WT.py:410-436 (t2v).  q from x, k/v from the 512-row text context, no key mask
    (context_lens=None, VC.py:357), no RoPE."""
    B, L, C = x.shape
    n, d = num_heads, C // num_heads
    q = rms_norm(linear(x, W[p + "q.weight"], W[p + "q.bias"], mode), W[p + "norm_q.weight"], eps, mode)
    k = rms_norm(linear(context, W[p + "k.weight"], W[p + "k.bias"], mode), W[p + "norm_k.weight"], eps, mode)
    v = linear(context, W[p + "v.weight"], W[p + "v.bias"], mode)
    o = attention(q.view(B, L, n, d), k.view(B, -1, n, d), v.view(B, -1, n, d), None, mode)
    return linear(o.flatten(2), W[p + "o.weight"], W[p + "o.bias"], mode)


def attention_block(W: Dict[str, Tensor], p: str, x: Tensor, e0: Tensor, seq_lens, grid_sizes,
                    freqs, context: Tensor, num_heads: int, eps: float = 1e-6,
                    token_offset: int = 0, mode: str = "fp32", attn_fn=None) -> Tensor:
    """WT.py:564-611 (adaLN-Zero block).  x [B,L,C]; e0 [B,6,C]."""
    rnd = _rounder(mode)
    e = rnd(W[p + "modulation"] + e0).chunk(6, dim=1)                           # WT.py:588
    t = rnd(rnd(layer_norm(x, None, None, eps, mode) * rnd(1 + e[1])) + e[0])    # WT.py:591
    y = self_attention(W, p + "self_attn.", t, seq_lens, grid_sizes, freqs, num_heads, eps,
                       token_offset, mode, attn_fn)
    x = rnd(x + rnd(y * e[2]))                                                    # WT.py:595
    n3 = layer_norm(x, W[p + "norm3.weight"], W[p + "norm3.bias"], eps, mode)    # WT.py:548-550,600
    x = rnd(x + cross_attention(W, p + "cross_attn.", n3, context, num_heads, eps, mode))
    t = rnd(rnd(layer_norm(x, None, None, eps, mode) * rnd(1 + e[4])) + e[3])    # WT.py:603
    h = gelu_tanh(linear(t, W[p + "ffn.0.weight"], W[p + "ffn.0.bias"], mode), mode)
    y = linear(h, W[p + "ffn.2.weight"], W[p + "ffn.2.bias"], mode)
    return rnd(x + rnd(y * e[5]))                                                 # WT.py:607


def patch_embed(x: Tensor, w: Tensor, b: Tensor, mode: str = "fp32") -> Tensor:
    """WT.py:758-759 / VC.py:199-201 + flatten(2).transpose(1,2) (VC.py:263, 343).
    Conv3d kernel=stride=(1,2,2): x [C,T,H,W] -> [T*(H/2)*(W/2), dim]."""
    y = F.conv3d(x.unsqueeze(0), w, b, stride=(1, 2, 2))
    return _rounder(mode)(y.flatten(2).transpose(1, 2))[0]


def head(W: Dict[str, Tensor], x: Tensor, e: Tensor, eps: float = 1e-6, mode: str = "fp32") -> Tensor:
    """WT.py:631-644.  x [B,L,C], e [B,C] -> [B,L,out*4]."""
    rnd = _rounder(mode)
    m = rnd(W["head.modulation"] + e.unsqueeze(1)).chunk(2, dim=1)
    t = rnd(rnd(layer_norm(x, None, None, eps, mode) * rnd(1 + m[1])) + m[0])
    return linear(t, W["head.head.weight"], W["head.head.bias"], mode)


def unpatchify(x: Tensor, grid_sizes, out_dim: int = 16, patch=(1, 2, 2)) -> List[Tensor]:
    """WT.py:1127-1150.  x [B,L,out*4] -> list of [out, F, 2H, 2W]."""
    out = []
    for u, v in zip(x, grid_sizes):
        u = u[:math.prod(v)].view(*v, *patch, out_dim)
        u = torch.einsum("fhwpqrc->cfphqwr", u)
        out.append(u.reshape(out_dim, *[i * j for i, j in zip(v, patch)]))
    return out


def time_embed(W: Dict[str, Tensor], t: Tensor, dim: int, freq_dim: int = 256, mode: str = "fp32"):
    """VC.py:347-354.  fp32 MLPs; e [B,C] and e0 [B,6,C] both cast to the activation dtype."""
    rnd = _rounder(mode)
    s = sinusoidal_embedding_1d(freq_dim, t).float()
    e = F.linear(F.silu(F.linear(s, W["time_embedding.0.weight"].float(), W["time_embedding.0.bias"].float())),
                 W["time_embedding.2.weight"].float(), W["time_embedding.2.bias"].float())
    e0 = F.linear(F.silu(e), W["time_projection.1.weight"].float(),
                  W["time_projection.1.bias"].float()).unflatten(1, (6, dim))
    return rnd(e), rnd(e0)


def text_embed(W: Dict[str, Tensor], context: Sequence[Tensor], text_len: int = 512, mode: str = "fp32") -> Tensor:
    """VC.py:358-363.  zero-pad each prompt to text_len rows, Linear-GELU(tanh)-Linear."""
    ctx = torch.stack([torch.cat([u, u.new_zeros(text_len - u.size(0), u.size(1))]) for u in context])
    h = gelu_tanh(linear(ctx, W["text_embedding.0.weight"], W["text_embedding.0.bias"], mode), mode)
    return linear(h, W["text_embedding.2.weight"], W["text_embedding.2.bias"], mode)


def pad_cat(xs: Sequence[Tensor], seq_len: int) -> Tensor:
    """WT.py:198-201 / VC.py:264-267: zero-pad each [L_i, C] to seq_len rows and stack."""
    return torch.stack([torch.cat([u, u.new_zeros(seq_len - u.size(0), u.size(1))]) for u in xs])


class Config:
    """Hyper-parameters of VerseCrafterWanTransformer3DModel.__init__ (VC.py:153-170)."""

    def __init__(self, dim=2048, ffn_dim=8192, num_heads=16, num_layers=32, in_dim=16, out_dim=16,
                 geoada_in_dim=128, text_dim=4096, text_len=512, freq_dim=256, eps=1e-6,
                 geoada_layers=None):
        self.dim, self.ffn_dim, self.num_heads, self.num_layers = dim, ffn_dim, num_heads, num_layers
        self.in_dim, self.out_dim, self.geoada_in_dim = in_dim, out_dim, geoada_in_dim
        self.text_dim, self.text_len, self.freq_dim, self.eps = text_dim, text_len, freq_dim, eps
        self.geoada_layers = list(range(0, num_layers, 2)) if geoada_layers is None else list(geoada_layers)
        assert 0 in self.geoada_layers                                           # VC.py:178
        self.geoada_layers_mapping = {i: n for n, i in enumerate(self.geoada_layers)}
        self.head_dim = dim // num_heads

    @staticmethod
    def wan_14b():
        return Config(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40)

    @staticmethod
    def wan_1_3b():
        return Config(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30)


def forward_geoada(W, cfg: Config, x: Tensor, geoada_context: Sequence[Tensor], seq_len: int,
                   e0, seq_lens, grid_sizes, freqs, context, sp=(1, 0), mode="fp32", attn_fn=None):
    """VC.py:254-292 + VC.py:112-125.  Returns the list of NA hints [B, L(/P), C]."""
    rnd = _rounder(mode)
    P, rank = sp
    c = pad_cat([patch_embed(u, W["geoada_patch_embedding.weight"], W["geoada_patch_embedding.bias"], mode)
                 for u in geoada_context], seq_len)
    off = 0
    if P > 1:
        c = torch.chunk(c, P, dim=1)[rank]                                        # VC.py:269-270
        off = rank * c.shape[1]
    hints = []
    for n, _layer in enumerate(cfg.geoada_layers):
        p = f"geoada_blocks.{n}."
        if n == 0:                                                                 # VC.py:113-114
            c = rnd(linear(c, W[p + "before_proj.weight"], W[p + "before_proj.bias"], mode) + x)
        c = attention_block(W, p, c, e0, seq_lens, grid_sizes, freqs, context, cfg.num_heads,
                            cfg.eps, off, mode, attn_fn)
        hints.append(linear(c, W[p + "after_proj.weight"], W[p + "after_proj.bias"], mode))  # VC.py:121
    return hints


def forward(W: Dict[str, Tensor], cfg: Config, x: Sequence[Tensor], t: Tensor,
            geoada_context: Sequence[Tensor], context: Sequence[Tensor], seq_len: int,
            geoada_context_scale: float = 1.0, freqs: Optional[Tensor] = None, sp=(1, 0),
            mode: str = "fp32", attn_fn=None, all_gather=None, run_main_blocks: bool = True,
            residual: Optional[Tensor] = None, return_residual: bool = False):
    """VC.py:295-442.  x: list/batch of [16,T,h,w]; returns [B,16,T,h,w].

    sp=(P, rank) reproduces the contiguous sequence chunking of VC.py:366-367 / 269-270;
    `attn_fn` then has to implement the Ulysses exchange and `all_gather` VC.py:432-433.
    run_main_blocks=False / residual=...: the TeaCache skip branch (VC.py:390-396).
    """
    rnd = _rounder(mode)
    P, rank = sp
    if freqs is None:
        freqs = rope_table(cfg.head_dim)
    xs = [patch_embed(u, W["patch_embedding.weight"], W["patch_embedding.bias"], mode) for u in x]
    grid_sizes = [(u.shape[1], u.shape[2] // 2, u.shape[3] // 2) for u in x]
    seq_lens = [u.shape[0] for u in xs]
    if P > 1:
        seq_len = int(math.ceil(seq_len / P)) * P                                  # WT.py:195-196
    assert max(seq_lens) <= seq_len                                                # WT.py:197
    xx = pad_cat(xs, seq_len)
    e, e0 = time_embed(W, t, cfg.dim, cfg.freq_dim, mode)
    ctx = text_embed(W, context, cfg.text_len, mode)
    off = 0
    if P > 1:
        xx = torch.chunk(xx, P, dim=1)[rank]                                       # VC.py:366-367
        off = rank * xx.shape[1]
    hints = forward_geoada(W, cfg, xx, geoada_context, seq_len, e0, seq_lens, grid_sizes, freqs,
                           ctx, sp, mode, attn_fn)
    ori = xx
    if run_main_blocks:
        for i in range(cfg.num_layers):                                            # VC.py:66-84
            xx = attention_block(W, f"blocks.{i}.", xx, e0, seq_lens, grid_sizes, freqs, ctx,
                                 cfg.num_heads, cfg.eps, off, mode, attn_fn)
            if i in cfg.geoada_layers_mapping:                                     # VC.py:146-147
                xx = rnd(xx + rnd(hints[cfg.geoada_layers_mapping[i]] * geoada_context_scale))
    else:
        xx = rnd(xx + residual)                                                    # VC.py:396
    res = rnd(xx - ori)                                                            # VC.py:409
    y = head(W, xx, e, cfg.eps, mode)                                              # VC.py:430
    if P > 1:
        y = all_gather(y)                                                          # VC.py:432-433
    out = torch.stack(unpatchify(y, grid_sizes, cfg.out_dim))                      # VC.py:436-437
    return (out, res) if return_residual else out


# --------------------------------------------------------------------------------------
# TeaCache gate (WT.py:205-245; the TeaCache class itself is third-party)
# --------------------------------------------------------------------------------------
class TeaCacheState:
    def __init__(self, coefficients, num_steps, rel_l1_thresh, num_skip_start_steps=0):
        self.coefficients = [float(c) for c in coefficients]
        self.num_steps, self.rel_l1_thresh = num_steps, rel_l1_thresh
        self.num_skip_start_steps = num_skip_start_steps
        self.reset()

    def reset(self):
        self.cnt = 0
        self.accumulated = 0.0
        self.prev = None
        self.should_calc = True

    def rescale(self, x: float) -> float:                      # np.poly1d(coefficients)(x), Horner
        y = 0.0
        for c in self.coefficients:
            y = y * x + c
        return y


def teacache_gate(st: TeaCacheState, e0: Tensor) -> bool:
    """WT.py:219-243 with cond_flag=True and t.dim()==1 (modulated_inp = e0)."""
    if st.cnt < st.num_skip_start_steps:
        should, st.accumulated = True, 0.0
    else:
        rel = ((e0.float() - st.prev.float()).abs().mean() / st.prev.float().abs().mean()).item()
        st.accumulated += st.rescale(rel)
        if st.accumulated < st.rel_l1_thresh:
            should = False
        else:
            should, st.accumulated = True, 0.0
    st.prev = e0
    st.should_calc = should
    return should


def teacache_advance(st: TeaCacheState):
    """VC.py:438-441."""
    st.cnt += 1
    if st.cnt == st.num_steps:
        st.reset()


# --------------------------------------------------------------------------------------
# pipeline-side helpers on the path (PIPE.py)
# --------------------------------------------------------------------------------------
def geoada_encode_masks(mask: Tensor, vae_stride=(4, 8, 8)) -> Tensor:
    """PIPE.py:440-486 (ref_images=None).  mask [C,T,H,W] -> [64, (T+3)//4, H/8, W/8]:
    8x8 pixel-unshuffle of channel 0 then nearest-exact temporal resize."""
    c, depth, height, width = mask.shape
    new_depth = int((depth + 3) // vae_stride[0])
    height = 2 * (int(height) // (vae_stride[1] * 2))
    width = 2 * (int(width) // (vae_stride[2] * 2))
    m = mask[0].view(depth, height, vae_stride[1], width, vae_stride[1])
    m = m.permute(2, 4, 0, 1, 3).reshape(vae_stride[1] * vae_stride[2], depth, height, width)
    return F.interpolate(m.unsqueeze(0), size=(new_depth, height, width), mode="nearest-exact").squeeze(0)


def cfg_combine(noise_pred: Tensor, guidance_scale: float) -> Tensor:
    """PIPE.py:904-906.  batch order [uncond, cond] (PIPE.py:741)."""
    u, c = noise_pred.chunk(2)
    return u + guidance_scale * (c - u)


def seq_len_for(latent_shape) -> int:
    """PIPE.py:861-865 with patch (1,2,2): ceil(h*w/4 * T)."""
    _, T, h, w = latent_shape
    return math.ceil((h * w) / 4 * T)


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
def state_dict_shapes(cfg: Config) -> Dict[str, tuple]:
    """Key -> shape map of the reference state dict."""
    d, f = cfg.dim, cfg.ffn_dim
    s = {
        "patch_embedding.weight": (d, cfg.in_dim, 1, 2, 2), "patch_embedding.bias": (d,),
        "geoada_patch_embedding.weight": (d, cfg.geoada_in_dim, 1, 2, 2), "geoada_patch_embedding.bias": (d,),
        "text_embedding.0.weight": (d, cfg.text_dim), "text_embedding.0.bias": (d,),
        "text_embedding.2.weight": (d, d), "text_embedding.2.bias": (d,),
        "time_embedding.0.weight": (d, cfg.freq_dim), "time_embedding.0.bias": (d,),
        "time_embedding.2.weight": (d, d), "time_embedding.2.bias": (d,),
        "time_projection.1.weight": (6 * d, d), "time_projection.1.bias": (6 * d,),
        "head.modulation": (1, 2, d), "head.head.weight": (cfg.out_dim * 4, d), "head.head.bias": (cfg.out_dim * 4,),
    }

    def block(p):
        s[p + "modulation"] = (1, 6, d)
        for a in ("self_attn", "cross_attn"):
            for l in "qkvo":
                s[f"{p}{a}.{l}.weight"] = (d, d)
                s[f"{p}{a}.{l}.bias"] = (d,)
            s[f"{p}{a}.norm_q.weight"] = (d,)
            s[f"{p}{a}.norm_k.weight"] = (d,)
        s[p + "norm3.weight"] = (d,)
        s[p + "norm3.bias"] = (d,)
        s[p + "ffn.0.weight"] = (f, d)
        s[p + "ffn.0.bias"] = (f,)
        s[p + "ffn.2.weight"] = (d, f)
        s[p + "ffn.2.bias"] = (d,)

    for i in range(cfg.num_layers):
        block(f"blocks.{i}.")
    for n in range(len(cfg.geoada_layers)):
        p = f"geoada_blocks.{n}."
        block(p)
        if n == 0:
            s[p + "before_proj.weight"] = (d, d)
            s[p + "before_proj.bias"] = (d,)
        s[p + "after_proj.weight"] = (d, d)
        s[p + "after_proj.bias"] = (d,)
    return s


def random_weights(cfg: Config, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Synthetic weights: Xavier-uniform matrices (incl. the reference's zero-initialised
    head.head / before_proj / after_proj, otherwise the output is identically 0), small random
    biases, norm weights ~ 1, modulation randn/sqrt(d).  Drawn from numpy's frozen RandomState
    stream so that fixtures made from them stay reproducible across torch versions."""
    import numpy as np
    rs = np.random.RandomState(seed)
    W = {}
    for k, shp in state_dict_shapes(cfg).items():
        if k.endswith("modulation"):
            w = rs.standard_normal(shp) / cfg.dim ** 0.5
        elif "norm" in k and k.endswith("weight"):
            w = 1.0 + 0.1 * rs.standard_normal(shp)
        elif k.endswith("bias"):
            w = 0.02 * rs.standard_normal(shp)
        else:
            fan_out = shp[0]
            fan_in = math.prod(shp[1:])
            a = math.sqrt(6.0 / (fan_in + fan_out))
            w = rs.uniform(-a, a, shp)
        W[k] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(dtype)
    return W

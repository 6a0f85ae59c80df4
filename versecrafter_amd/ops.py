"""Per-kernel entry points of libvcengine on torch CUDA tensors (used by the parity tests and by tuning
scripts).  Each function enqueues one HIP kernel on torch's current stream; nothing here computes on the
host, and every call fails loudly if the tensors are not bf16 CUDA tensors."""
import ctypes as C
import math

import torch

from . import _lib

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_BIAS_GATE_RESID = 0, 1, 2, 3


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, dtype=torch.bfloat16):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (HIP) tensor: versecrafter_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def gemm(a, w, bias=None, epilogue=EPI_BIAS, resid=None, gate=None, rows_per_batch=0, hint=None, hint_scale=0.0,
         out=None, tile=0):
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T + bias).  gate: [B, N] rows selected by m // rows_per_batch."""
    lib = _lib.load()
    _chk(a, "a"); _chk(w, "w"); _chk(bias, "bias"); _chk(resid, "resid"); _chk(gate, "gate"); _chk(hint, "hint")
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    rc = lib.vc_op_gemm_bf16(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(out), out.stride(0), _ptr(bias), M, N, K,
                             epilogue, _ptr(resid), 0 if resid is None else resid.stride(0), _ptr(gate),
                             0 if gate is None else gate.stride(0), rows_per_batch, _ptr(hint),
                             0 if hint is None else hint.stride(0), float(hint_scale), tile, _stream())
    _lib.check(rc)
    return out


def quantize_rows_fp8(x):
    """bf16 [M, K] -> (OCP e4m3 bytes as uint8 [M, K], float32 scale [M]): x ~ q * scale[:, None], scale = amax / 448 per row."""
    lib = _lib.load()
    _chk(x, "x")
    M, K = x.shape
    assert x.stride(1) == 1
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    _lib.check(lib.vc_op_quantize_rows_fp8(_ptr(x), x.stride(0), _ptr(q), q.stride(0), _ptr(sc), M, K, _stream()))
    return q, sc


def gemm_fp8(a_q, a_scale, w_q, w_scale, bias=None, epilogue=EPI_BIAS, resid=None, gate=None, rows_per_batch=0, out=None, a_rows_padded=False):
    """out[M,N] bf16 = epilogue((a_q[M,K] @ w_q[N,K]^T) * a_scale[:, None] * w_scale[None, :] + bias) on e4m3 operands (uint8 storage):
    v_mfma_scale_f32_16x16x128_f8f6f4, fp32 accumulation.  BASELINE config 5's dtype; a capability of this build, off by default."""
    lib = _lib.load()
    M, K = a_q.shape
    N = w_q.shape[0]
    assert a_q.dtype == torch.uint8 and w_q.dtype == torch.uint8 and w_q.shape[1] == K and a_q.stride(1) == 1 and w_q.stride(1) == 1
    assert a_scale.dtype == torch.float32 and w_scale.dtype == torch.float32 and a_scale.numel() == M and w_scale.numel() == N
    _chk(bias, "bias"); _chk(resid, "resid"); _chk(gate, "gate")
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a_q.device)
    rc = lib.vc_op_gemm_fp8(_ptr(a_q), a_q.stride(0), _ptr(a_scale), _ptr(w_q), w_q.stride(0), _ptr(w_scale), _ptr(out), out.stride(0),
                            _ptr(bias), M, N, K, epilogue, _ptr(resid), 0 if resid is None else resid.stride(0), _ptr(gate),
                            0 if gate is None else gate.stride(0), rows_per_batch, 1 if a_rows_padded else 0, _stream())
    _lib.check(rc)
    return out


def attention(q, k, v, k_len=0, scale=None, out=None, variant=0):
    """q [B,Lq,H,128], k/v [B,Lk,H,128] (any strides with a contiguous last dim) -> [B,Lq,H,128].
    variant (tests / A-B tools): MFMA shape of the pipelined kernel, 32 or 16; 0 = the library's default."""
    lib = _lib.load()
    _chk(q, "q"); _chk(k, "k"); _chk(v, "v")
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    assert D == 128 and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    if out is None:
        out = torch.empty(B, Lq, H, D, dtype=torch.bfloat16, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    st = lambda t: _lib.i64x3(t.stride(0), t.stride(1), t.stride(2))
    if variant:
        rc = lib.vc_op_attention_variant(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, H, Lq, Lk, st(q), st(k), st(v), st(out),
                                         int(k_len), float(scale), int(variant), _stream())
    else:
        rc = lib.vc_op_attention(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, H, Lq, Lk, st(q), st(k), st(v), st(out),
                                 int(k_len), float(scale), _stream())
    _lib.check(rc)
    return out


def attention_fp8(q, k, v, k_len=0, scale=None, pmode=1, out=None, workspace=None, stage=0, return_workspace=False):
    """fp8 self-attention (this build; csrc/attention_fp8.hip): bf16 q [B,Lq,H,128], k / v [B,Lk,H,128] -> bf16 [B,Lq,H,128]; q, k, v and the
    softmax weights are e4m3 under MX-style block scales inside.  pmode 1: piecewise-linear 2^x for the weights' bytes; 0: v_exp_f32."""
    lib = _lib.load()
    _chk(q, "q"); _chk(k, "k"); _chk(v, "v")
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    assert D == 128 and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    if out is None:
        out = torch.empty(B, Lq, H, D, dtype=torch.bfloat16, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    need = int(lib.vc_op_attention_fp8_workspace_bytes(B, H, Lq, Lk))
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=q.device)
    assert workspace.dtype == torch.uint8 and workspace.numel() >= need and workspace.data_ptr() % 256 == 0
    st = lambda t: _lib.i64x3(t.stride(0), t.stride(1), t.stride(2))
    _lib.check(lib.vc_op_attention_fp8(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, H, Lq, Lk, st(q), st(k), st(v), st(out), int(k_len),
                                       float(scale), int(pmode), int(stage), _ptr(workspace), workspace.numel(), _stream()))
    return (out, workspace) if return_workspace else out


def attention_lse(q, k, v, k_len=0, scale=None):
    """attention() over one block of keys -> (out [B,Lq,H,128] bf16, lse [B,H,Lq] float32: log2 of the row's sum of exp2(logit * scale * log2 e))."""
    lib = _lib.load()
    _chk(q, "q"); _chk(k, "k"); _chk(v, "v")
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    assert D == 128 and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    out = torch.empty(B, Lq, H, D, dtype=torch.bfloat16, device=q.device)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    st = lambda t: _lib.i64x3(t.stride(0), t.stride(1), t.stride(2))
    _lib.check(lib.vc_op_attention_lse(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), B, H, Lq, Lk, st(q), st(k), st(v), st(out), int(k_len),
                                       float(scale), _stream()))
    return out, lse


def attention_merge(parts, lses):
    """Merge R partial attention outputs (each [B,Lq,H,128] bf16 contiguous, normalised over its own key block) with their log-sum-exps."""
    import ctypes as C
    lib = _lib.load()
    R = len(parts)
    B, Lq, H, D = parts[0].shape
    for p_, l_ in zip(parts, lses):
        _chk(p_, "part")
        assert p_.is_contiguous() and l_.is_contiguous() and l_.dtype == torch.float32 and tuple(l_.shape) == (B, H, Lq)
    out = torch.empty(B, Lq, H, D, dtype=torch.bfloat16, device=parts[0].device)
    pa = (C.c_void_p * R)(*[p_.data_ptr() for p_ in parts])
    la = (C.c_void_p * R)(*[l_.data_ptr() for l_ in lses])
    _lib.check(lib.vc_op_attention_merge(pa, la, R, _ptr(out), B, H, Lq, _lib.i64x3(out.stride(0), out.stride(1), out.stride(2)), _stream()))
    return out


def attention_padmerge(q, k, v, pad_from, scale=None):
    """attention() where, per batch b, the keys pad_from[b] .. Lk-1 are identical rows (zero-padded prompt positions):
    they are folded into one key with multiplicity Lk - pad_from[b].  Same result as attention(q, k, v) up to rounding."""
    lib = _lib.load()
    _chk(q, "q"); _chk(k, "k"); _chk(v, "v")
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    assert D == 128 and len(pad_from) == B
    out = torch.empty(B, Lq, H, D, dtype=torch.bfloat16, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    st = lambda t: _lib.i64x3(t.stride(0), t.stride(1), t.stride(2))
    pf = (C.c_int32 * B)(*[int(v_) for v_ in pad_from])
    rc = lib.vc_op_attention_padmerge(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, H, Lq, Lk, st(q), st(k), st(v), st(out), pf,
                                      float(scale), _stream())
    _lib.check(rc)
    return out


def attention_segmented(q, k, v, k_len=0, scale=None):
    """Attention on the Ulysses receive layout: q, k, v [S, B, Lseg, H, 128] (segment s holds tokens s*Lseg .. of the
    sequence) -> out in the same layout.  Equivalent to attention() on the [B, S*Lseg, H, 128] concatenation."""
    lib = _lib.load()
    _chk(q, "q"); _chk(k, "k"); _chk(v, "v")
    S, B, Ls, H, D = q.shape
    assert D == 128 and k.shape == q.shape and v.shape == q.shape
    assert q.stride(4) == 1 and k.stride(4) == 1 and v.stride(4) == 1
    out = torch.empty(S, B, Ls, H, D, dtype=torch.bfloat16, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    st = lambda t: (C.c_int64 * 4)(t.stride(1), t.stride(2), t.stride(3), t.stride(0))
    rc = lib.vc_op_attention_segmented(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, H, S * Ls, st(q), st(k), st(v), st(out),
                                       Ls, int(k_len), float(scale), _stream())
    _lib.check(rc)
    return out


def layernorm_modulate(x, scale, shift, rows_per_batch, eps=1e-6):
    """y = LN(x) * (1 + scale[b]) + shift[b];  x [rows, dim], scale/shift [B, dim] (same row stride)."""
    lib = _lib.load()
    _chk(x, "x"); _chk(scale, "scale"); _chk(shift, "shift")
    assert x.is_contiguous() and scale.stride(0) == shift.stride(0)
    y = torch.empty_like(x)
    rc = lib.vc_op_layernorm(_ptr(x), _ptr(y), x.shape[0], x.shape[1], rows_per_batch, eps, 0, _ptr(scale),
                             _ptr(shift), scale.stride(0), _stream())
    _lib.check(rc)
    return y


def layernorm_affine(x, weight, bias, eps=1e-6):
    lib = _lib.load()
    _chk(x, "x"); _chk(weight, "weight"); _chk(bias, "bias")
    assert x.is_contiguous()
    y = torch.empty_like(x)
    rc = lib.vc_op_layernorm(_ptr(x), _ptr(y), x.shape[0], x.shape[1], 0, eps, 1, _ptr(weight), _ptr(bias), 0,
                             _stream())
    _lib.check(rc)
    return y


def rope_table_device(cis, device):
    """complex128 [1024, 64] cis table -> float2 (cos, sin) device table the kernels read."""
    t = torch.view_as_real(cis.to(torch.complex128)).to(torch.float32).contiguous()
    return t.to(device)


def rmsnorm_rope_(x, weight, eps=1e-6, table=None, grid=None, token_offset=0, rows_per_batch=0):
    """In place on x [rows, dim] (row stride free): WanRMSNorm, then rope_apply when `table` is given.
    grid = (F, H, W) of the token lattice."""
    lib = _lib.load()
    _chk(x, "x"); _chk(weight, "weight")
    assert x.stride(1) == 1
    g = None
    if table is not None:
        _chk(table, "table", torch.float32)
        g = (C.c_int32 * 5)(int(grid[0]), int(grid[1]), int(grid[2]), int(token_offset), int(rows_per_batch))
    rc = lib.vc_op_rmsnorm_rope(_ptr(x), x.stride(0), x.shape[0], x.shape[1], _ptr(weight), eps, _ptr(table), g,
                                _stream())
    _lib.check(rc)
    return x


def qkv_front(qkv, wq, wk, table, grid, token_offset=0, rows_per_batch=0, eps=1e-6, P=1, pack=False):
    """Self-attention front on qkv [rows, 3*dim] (contiguous): WanRMSNorm + rope_apply of the q and k thirds in ONE pass.
    pack=False: in place, returns qkv.  pack=True: returns send [3, B, P, Lloc, dim/P] (B * Lloc = rows, Lloc = rows_per_batch or
    rows) -- q, k (normed, rotated) and v in the Ulysses exchange layout (dist.pack_qkv); qkv is left untouched."""
    lib = _lib.load()
    _chk(qkv, "qkv"); _chk(wq, "wq"); _chk(wk, "wk"); _chk(table, "table", torch.float32)
    assert qkv.is_contiguous() and qkv.shape[1] % 3 == 0
    rows, dim = qkv.shape[0], qkv.shape[1] // 3
    g = (C.c_int32 * 5)(int(grid[0]), int(grid[1]), int(grid[2]), int(token_offset), int(rows_per_batch))
    rpb = int(rows_per_batch) if rows_per_batch else rows
    assert rows % rpb == 0
    send = torch.empty(3, rows // rpb, P, rpb, dim // P, dtype=torch.bfloat16, device=qkv.device) if pack else None
    rc = lib.vc_op_qkv_front(_ptr(qkv), rows, dim, _ptr(wq), _ptr(wk), eps, _ptr(table), g, _ptr(send), P, _stream())
    _lib.check(rc)
    return send if pack else qkv


def geoada_context(z, mask):
    """PIPE.py:440-488 for one sample: z [64,T,h,w] bf16 control latents, mask [C,F,H,W] (bf16 or fp32; channel 0 is
    used, as in the reference) -> [128,T,h,w] bf16 = cat(z, nearest-exact frame resize of the 8x8 pixel-unshuffled mask)."""
    lib = _lib.load()
    _chk(z, "z")
    if not mask.is_cuda:
        raise RuntimeError("mask must be a CUDA (HIP) tensor: versecrafter_amd has no CPU path")
    if mask.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError(f"mask must be bfloat16 or float32, got {mask.dtype}")
    if z.dim() != 4 or z.shape[0] != 64 or mask.dim() != 4:
        raise ValueError(f"z must be [64,T,h,w] and mask [C,F,H,W]; got {tuple(z.shape)} and {tuple(mask.shape)}")
    _, T, h, w = z.shape
    _, F, H, W = mask.shape
    z = z.contiguous()
    m0 = mask[0].contiguous()
    out = torch.empty(128, T, h, w, dtype=torch.bfloat16, device=z.device)
    rc = lib.vc_op_geoada_context(_ptr(z), _ptr(m0), int(mask.dtype == torch.float32), _ptr(out), T, h, w, F, H, W,
                                  _stream())
    if rc != 0:
        raise ValueError(f"geoada_context: mask {tuple(mask.shape)} does not map onto latents {tuple(z.shape)} "
                         "(needs T = (F+3)//4, H = 8h, W = 8w, h and w even; PIPE.py:459-466)")
    return out


def unipc_update(noise_pred, sample, scalars, flags, last=None, m0=None, m1=None, want_sample=True):
    """Fused CFG combine + flow x0 + UniPC corrector + predictor (include/vcengine.h: vc_op_unipc_update).
    noise_pred: [2, ...] (uncond, cond) when flags & 1 else [1, ...] / [...]; returns (x0, corrected sample or None, next)."""
    lib = _lib.load()
    _chk(noise_pred, "noise_pred"); _chk(sample, "sample"); _chk(last, "last"); _chk(m0, "m0"); _chk(m1, "m1")
    n = sample.numel()
    noise_pred = noise_pred.contiguous()
    if flags & 1:
        assert noise_pred.numel() == 2 * n
        nu, nc = noise_pred.view(2, -1)[0], noise_pred.view(2, -1)[1]
    else:
        assert noise_pred.numel() == n
        nu, nc = None, noise_pred.view(-1)
    sample = sample.contiguous()
    x0 = torch.empty_like(sample)
    nxt = torch.empty_like(sample)
    samp = torch.empty_like(sample) if (want_sample and flags & 2) else None
    sc = (C.c_float * 13)(*[float(v) for v in scalars])
    cont = lambda t: None if t is None else t.contiguous()
    last, m0, m1 = cont(last), cont(m0), cont(m1)
    rc = lib.vc_op_unipc_update(_ptr(nu), _ptr(nc), _ptr(sample), _ptr(last), _ptr(m0), _ptr(m1), _ptr(x0), _ptr(samp),
                                _ptr(nxt), n, sc, int(flags), _stream())
    _lib.check(rc)
    return x0, samp, nxt

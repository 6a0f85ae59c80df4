"""CPU restatement of the fp8 self-attention mode (versecrafter_amd/csrc/attention_fp8.hip).  TEST INFRASTRUCTURE: only tests/ may import
this file; nothing under versecrafter_amd/ does.

The reference (ztitomir/VerseCrafter) has no fp8 arithmetic: its self-attention is `attention(q, k, v, k_lens=...)` in bf16 through
flash-attn (versecrafter/models/wan_transformer3d.py:394-399).  The fp8 mode is a capability of this build (BASELINE config 5 names "fp8
MFMA"), so there is nothing in the reference to pin it against -- PARITY UNPINNED BY NATURE.  What IS pinned here is the definition, so
that the HIP kernels can be checked bit for bit (the quantiser and its layouts) and to rounding (the attention):

  quantise(q, k, v, scale)    q * (scale * log2(e) * 8 / 65535), k, v  ->  OCP e4m3 bytes + one E8M0 scale byte per block of 32 elements
                              along the contraction: blocks of a q / k row are d in {64 s + 16 b .. +15} u {64 s + 32 + 16 b .. +15}
                              (s, b in {0, 1}: what one k-block of v_mfma_scale_f32_32x32x64_f8f6f4 covers when a lane reads 32 contiguous
                              bytes of the row), a block of v is one column d over the keys 32 j .. 32 j + 31.  Scale = the smallest power of
                              two 2^e with amax <= 448 * 2^e; elements rounded to nearest even (torch.float8_e4m3fn's own cast).
  pack_workspace(...)         those arrays in the byte layout the kernels exchange (tile images with 144 / 80-byte row pitches).
  attention(...)              the kernel's arithmetic tile by tile: raw logits S from the dequantised operands, the deferred-rescale
                              reference m_run, the per-(row, tile) exponent et = ceil((m_tile - m_run) K1), the weights' e4m3 bytes
                              (pmode 1: byte = round(65535 (S - m_run) + 120 - 8 et), the piecewise-linear 2^x; pmode 0: e4m3(exp2(.))),
                              the block scale 2^(et - 8), O = sum P~ V~, l = sum P~ over the SAME P~, out = O / l in bf16.
Sums run in float64 (the kernel accumulates in fp32 on the matrix pipe: differences are at the 1e-6 level, far below one e4m3 step).
"""
import math

import numpy as np
import torch

K1 = np.float32(65535.0 / 8.0)
DEFER_T = np.float32(8.0)
KT = 64


def qfold(scale: float) -> np.float32:
    return np.float32(float(scale) * 1.4426950408889634 * 8.0 / 65535.0)


def _scale_byte(amax: torch.Tensor) -> torch.Tensor:
    """smallest e with amax <= 448 * 2^e as the E8M0 byte e + 127, clamped to [0, 254]; amax float32 >= 0"""
    u = amax.contiguous().view(torch.int32)
    E = (u >> 23) & 0xFF
    sb = E - 8 + ((u & 0x7FFFFF) > 0x600000).to(torch.int32)
    return sb.clamp(0, 254)


def _inv_scale(sb: torch.Tensor) -> torch.Tensor:
    return ((254 - sb).to(torch.int32) << 23).view(torch.float32)


def _to_e4m3(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float8_e4m3fn).view(torch.uint8)


def _from_e4m3(b: torch.Tensor) -> torch.Tensor:
    return b.view(torch.float8_e4m3fn).to(torch.float32)


def _qk_block_index():
    """d -> block index 2 s + b of a q / k row"""
    d = torch.arange(128)
    s = d // 64
    b = (d % 32) // 16
    return (2 * s + b).long()


def quantise_rows(x: torch.Tensor, fold=None):
    """x [..., 128] (bf16 or float) -> (bytes uint8 [..., 128], scale bytes int32 [..., 4])"""
    xf = x.to(torch.float32)
    if fold is not None:
        xf = xf * torch.tensor(float(fold), dtype=torch.float32)              # one fp32 rounding, as the kernel
    blk = _qk_block_index()
    out = torch.empty(xf.shape, dtype=torch.uint8)
    sbs = torch.empty(xf.shape[:-1] + (4,), dtype=torch.int32)
    for i in range(4):
        sel = (blk == i).nonzero().flatten()
        part = xf[..., sel]
        sb = _scale_byte(part.abs().amax(dim=-1))
        out[..., sel] = _to_e4m3(part * _inv_scale(sb)[..., None])
        sbs[..., i] = sb
    return out, sbs


def quantise_v(v: torch.Tensor):
    """v [B, Lk, H, 128] -> bytes uint8 [B, H, nT*64, 128] (rows past Lk are zero), scale bytes int32 [B, H, nT*2, 128] (per 32 keys, per d)"""
    B, Lk, H, D = v.shape
    nT = (Lk + KT - 1) // KT
    vf = torch.zeros(B, H, nT * KT, D, dtype=torch.float32)
    vf[:, :, :Lk] = v.to(torch.float32).permute(0, 2, 1, 3)
    blocks = vf.view(B, H, nT * 2, 32, D)
    sb = _scale_byte(blocks.abs().amax(dim=3))                                 # [B, H, nT*2, D]
    q8 = _to_e4m3(blocks * _inv_scale(sb)[:, :, :, None, :]).view(B, H, nT * KT, D)
    return q8, sb


def quantise(q, k, v, scale=None, k_len=0):
    """q [B, Lq, H, 128], k / v [B, Lk, H, 128] -> dict of logical arrays (heads second).  Keys past k_len are quantised as zeros."""
    if scale is None:
        scale = 1.0 / math.sqrt(128)
    if 0 < k_len < k.shape[1]:
        k, v = k.clone(), v.clone()
        k[:, k_len:] = 0
        v[:, k_len:] = 0
    q8, qs = quantise_rows(q.permute(0, 2, 1, 3), qfold(scale))
    k8, ks = quantise_rows(k.permute(0, 2, 1, 3))
    v8, vs = quantise_v(v)
    return dict(q8=q8, qs=qs, k8=k8, ks=ks, v8=v8, vs=vs, Lq=q.shape[1], Lk=k.shape[1])


def _up256(n):
    return (n + 255) // 256 * 256


REC = 9216 + 10240      # one key-tile RECORD of the workspace: the K image of tile j followed by the V^T image of tile j - 1


def workspace_bytes(B, H, Lq, Lk):
    nTq, nTk, bh = (Lq + KT - 1) // KT, (Lk + KT - 1) // KT, B * H
    return _up256(bh * nTq * KT * 128) + _up256(bh * nTq * KT * 4) + _up256(bh * (nTk + 3) * REC)


def pack_workspace(Q):
    """The byte image the kernels exchange (attention_fp8.hip's header): returns (uint8 array, mask of the bytes that are defined)."""
    q8, qs, k8, ks, v8, vs, Lq, Lk = (Q[n] for n in ("q8", "qs", "k8", "ks", "v8", "vs", "Lq", "Lk"))
    B, H = q8.shape[:2]
    nTq, nTk, bh = (Lq + KT - 1) // KT, (Lk + KT - 1) // KT, B * H
    ws = np.zeros(workspace_bytes(B, H, Lq, Lk), dtype=np.uint8)
    known = np.zeros_like(ws, dtype=bool)
    off = 0
    # Q8 [bh][tile][64][128] natural order; rows past Lq undefined
    a = np.zeros((bh, nTq * KT, 128), np.uint8); m = np.zeros_like(a, bool)
    a[:, :Lq] = q8.reshape(bh, Lq, 128).numpy(); m[:, :Lq] = True
    ws[off:off + a.size] = a.ravel(); known[off:off + a.size] = m.ravel(); off += _up256(a.size)
    a = np.zeros((bh, nTq * KT, 4), np.uint8); m = np.zeros_like(a, bool)
    a[:, :Lq] = qs.reshape(bh, Lq, 4).numpy().astype(np.uint8); m[:, :Lq] = True
    ws[off:off + a.size] = a.ravel(); known[off:off + a.size] = m.ravel(); off += _up256(a.size)
    # records [bh][nTk + 3][19456]: record j = K image of tile j (j < nTk) | V^T image of tile j - 1 (1 <= j <= nTk); the rest is never
    # read by arithmetic (the attention kernel copies whole records, three past the last tile) and stays undefined
    rec = np.zeros((bh, nTk + 3, REC), np.uint8); rm = np.zeros_like(rec, bool)
    # K image: 64 rows (keys) at a pitch of 144: 128 data bytes, then 16 pad bytes of which the first four of row 32 b + r hold the scale
    # bytes the MFMA lane 32 b + r supplies: byte kb * 2 + s  <-  key kb * 32 + r, block 2 s + b; rows past Lk are zero with scale byte 0
    kk = np.zeros((bh, nTk * KT, 128), np.uint8)
    kk[:, :Lk] = k8.reshape(bh, Lk, 128).numpy()
    kss = np.zeros((bh, nTk * KT, 4), np.uint8)
    kss[:, :Lk] = ks.reshape(bh, Lk, 4).numpy().astype(np.uint8)
    kss = kss.reshape(bh, nTk, 2, 32, 2, 2)                       # [bh][tile][kb][r][s][b]
    a = np.zeros((bh, nTk, KT, 144), np.uint8)
    a[..., :128] = kk.reshape(bh, nTk, KT, 128)
    a[..., 128:132] = kss.transpose(0, 1, 5, 3, 2, 4).reshape(bh, nTk, KT, 4)     # [bh][tile][b][r][kb][s] -> row 32 b + r
    rec[:, :nTk, :9216] = a.reshape(bh, nTk, 9216); rm[:, :nTk, :9216] = True
    # V^T image: 128 rows (d) at a pitch of 80: byte 32 h + 16 kb + j  <-  key kb * 32 + (j & 3) + 8 (j >> 2) + 4 h; pad bytes 64..67 of row
    # 32 kb + r (rows 0..63) hold the scale bytes of lane 32 kb + r: byte db  <-  d = db * 32 + r, key block kb
    vv = v8.reshape(bh, nTk, KT, 128).numpy()
    a = np.zeros((bh, nTk, 128, 80), np.uint8)
    for hh in range(2):
        for kb in range(2):
            for j in range(16):
                key = kb * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh
                a[:, :, :, 32 * hh + 16 * kb + j] = vv[:, :, key, :]
    vss = vs.reshape(bh, nTk, 2, 4, 32).numpy().astype(np.uint8)    # [bh][tile][kb][db][r]
    a[:, :, :64, 64:68] = vss.transpose(0, 1, 2, 4, 3).reshape(bh, nTk, 64, 4)     # [bh][tile][kb][r][db] -> row 32 kb + r
    rec[:, 1:nTk + 1, 9216:] = a.reshape(bh, nTk, 10240); rm[:, 1:nTk + 1, 9216:] = True
    ws[off:off + rec.size] = rec.ravel(); known[off:off + rec.size] = rm.ravel()
    return ws, known


def dequantise(Q):
    """float64 q~ [B,H,Lq,128] (still folded), k~ [B,H,Lk,128], v~ [B,H,nT*64,128]"""
    blk = _qk_block_index()
    def rows(b8, sb):
        e = (sb.to(torch.float64) - 127.0)[..., blk]
        return _from_e4m3(b8).to(torch.float64) * torch.pow(torch.tensor(2.0, dtype=torch.float64), e)
    qd, kd = rows(Q["q8"], Q["qs"]), rows(Q["k8"], Q["ks"])
    B, H, n, D = Q["v8"].shape
    ev = (Q["vs"].to(torch.float64) - 127.0)                                   # [B,H,nT*2,D]
    vd = _from_e4m3(Q["v8"]).to(torch.float64).view(B, H, n // 32, 32, D) * torch.pow(torch.tensor(2.0, dtype=torch.float64), ev)[:, :, :, None, :]
    return qd, kd, vd.reshape(B, H, n, D)


def attention(q, k, v, k_len=0, scale=None, pmode=1, Q=None):
    """[B, Lq, H, 128] float32 result of the fp8 mode (before the final bf16 rounding) and the same rounded to bf16."""
    if Q is None:
        Q = quantise(q, k, v, scale, k_len)
    qd, kd, vd = dequantise(Q)
    B, H, Lq, D = qd.shape
    Lk = Q["Lk"]
    k_len = Lk if (k_len <= 0 or k_len > Lk) else k_len
    nt = (k_len + KT - 1) // KT
    out = torch.empty(B, H, Lq, D, dtype=torch.float64)
    f32 = lambda t: t.to(torch.float32)
    K1t, Tt = torch.tensor(float(K1), dtype=torch.float32), torch.tensor(float(DEFER_T), dtype=torch.float32)
    inv65535 = torch.tensor(np.float32(1.0) / np.float32(65535.0), dtype=torch.float32)
    for b in range(B):
        for hd in range(H):
            S = f32(qd[b, hd] @ kd[b, hd, :nt * KT].T if nt * KT <= kd.shape[2] else
                    qd[b, hd] @ torch.cat([kd[b, hd], torch.zeros(nt * KT - kd.shape[2], D, dtype=torch.float64)]).T)    # [Lq, nt*64] raw units
            key = torch.arange(nt * KT)
            S = torch.where(key[None, :] < k_len, S, torch.tensor(-1e30, dtype=torch.float32))
            St = S.view(Lq, nt, KT)
            m_tile_all = St.amax(dim=2)                                        # [Lq, nt]
            m_run = torch.full((Lq,), -1e30, dtype=torch.float32)
            m_new = torch.full((Lq,), -1e30, dtype=torch.float32)
            num = torch.zeros(Lq, D, dtype=torch.float64)
            den = torch.zeros(Lq, dtype=torch.float64)
            ref = torch.zeros(Lq, dtype=torch.float64)                         # log2 reference the running sums are expressed against
            first = True
            for t in range(nt):
                m_tile = m_tile_all[:, t]
                m_new = torch.maximum(m_new, m_tile)
                moved = (m_new - m_run) * K1t > Tt
                m_ref = torch.where(moved, m_new, m_run)
                # exact rescale of the running sums (the kernel multiplies by exp2 in fp32: same to ~1e-7)
                new_ref = m_ref.to(torch.float64) * float(K1)
                if first:
                    first = False
                else:
                    a = torch.pow(torch.tensor(2.0, dtype=torch.float64), ref - new_ref)
                    num *= a[:, None]; den *= a
                ref = new_ref
                m_run = m_ref
                et = torch.clamp(torch.ceil((m_tile - m_run) * K1t), min=-100.0)           # fp32
                Sc = St[:, t]                                                              # [Lq, 64]
                if pmode == 1:
                    kc = (120.0 - 8.0 * et) * inv65535 - m_run                             # fp32, the kernel's operation order
                    y = Sc + kc[:, None]
                    byte = torch.round(y.to(torch.float64).clamp(0.0, 1.0) * 65535.0).to(torch.int64) & 0xFF
                    pq = _from_e4m3(byte.to(torch.uint8)).to(torch.float64)
                else:
                    ka = 8.0 - et - m_run * K1t
                    pe = torch.exp2(f32(Sc.to(torch.float64) * float(K1) + ka.to(torch.float64)[:, None]))
                    pq = _from_e4m3(_to_e4m3(pe)).to(torch.float64)
                P = pq * torch.pow(torch.tensor(2.0, dtype=torch.float64), (et.to(torch.float64) - 8.0))[:, None]
                num += P @ vd[b, hd, t * KT:(t + 1) * KT]
                den += P.sum(dim=1)
            out[b, hd] = num / den[:, None]
    res = out.permute(0, 2, 1, 3).to(torch.float32).contiguous()
    return res, res.to(torch.bfloat16)


def exact_attention(q, k, v, k_len=0, scale=None):
    """softmax(q k^T scale) v in float64 on the bf16 inputs (what the bf16 path approximates): [B, Lq, H, 128] float32"""
    if scale is None:
        scale = 1.0 / math.sqrt(128)
    qd, kd, vd = (t.to(torch.float64).permute(0, 2, 1, 3) for t in (q, k, v))
    Lk = k.shape[1]
    k_len = Lk if (k_len <= 0 or k_len > Lk) else k_len
    s = qd @ kd.transpose(-1, -2) * scale
    s[..., k_len:] = -float("inf")
    return (torch.softmax(s, dim=-1) @ vd).permute(0, 2, 1, 3).to(torch.float32).contiguous()

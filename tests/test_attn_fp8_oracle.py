"""CPU checks of oracle/attn_fp8_oracle.py, the restatement the fp8 self-attention kernels are tested against (tests/test_gpu_attention_fp8.py).
The reference has no fp8 arithmetic, so these pin the DEFINITION: the block-scale rule, the byte mapping of the weights, the workspace layout,
and how far the mode sits from exact attention."""
import math

import numpy as np
import torch

from oracle import attn_fp8_oracle as A


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def test_block_scale_is_the_smallest_power_of_two_that_fits_448():
    amax = torch.tensor([0.0, 1e-30, 447.9, 448.0, 448.1, 1.0, 1.75, 1.7500001, 3.5, 896.0, 1e-3, 6e4], dtype=torch.float32)
    sb = A._scale_byte(amax)
    e = sb.double() - 127
    big = amax.double() > 2.0 ** -120
    assert (amax.double()[big] <= 448.0 * 2.0 ** e[big]).all()                 # fits ...
    assert (amax.double()[big] > 448.0 * 2.0 ** (e[big] - 1)).all()            # ... and the next smaller power of two would not
    assert sb[0] == 0 and sb[1] >= 0
    inv = A._inv_scale(sb)
    assert torch.equal(inv[big], torch.pow(torch.tensor(2.0, dtype=torch.float64), -e[big]).float())
    assert not torch.isnan(A._from_e4m3(A._to_e4m3(amax * inv))).any()         # nothing saturates to NaN


def test_qk_blocks_follow_the_mfma_k_blocks():
    blk = A._qk_block_index()
    # a lane reads 32 contiguous bytes (2s+h)*32.. of its row for k-step s; bytes 0-15 go to k-block 0, bytes 16-31 to k-block 1
    for s in range(2):
        for h in range(2):
            base = (2 * s + h) * 32
            assert (blk[base:base + 16] == 2 * s + 0).all() and (blk[base + 16:base + 32] == 2 * s + 1).all()
    assert [int((blk == i).sum()) for i in range(4)] == [32, 32, 32, 32]


def test_quantised_rows_dequantise_within_one_e4m3_step():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(2, 3, 50, 128, generator=g) * torch.logspace(-3, 2, 50)[None, None, :, None]).bfloat16()
    b8, sb = A.quantise_rows(x)
    blk = A._qk_block_index()
    deq = A._from_e4m3(b8).double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), (sb.double() - 127)[..., blk])
    amax = torch.stack([x.float().abs()[..., blk == i].amax(-1) for i in range(4)], -1)[..., blk].double()
    assert ((deq - x.double()).abs() <= amax * 2.0 ** -4 + 1e-30).all()          # half a step of the block's top binade at worst


def test_piecewise_linear_byte_is_within_6_2_percent_above_the_exponential():
    y = torch.linspace(-6.0, 8.0, 20001, dtype=torch.float64)                    # log2 of the scaled weight
    byte = torch.round(8.0 * y + 56.0).clamp(0, 126).to(torch.uint8)
    val = A._from_e4m3(byte).double()
    ratio = val / torch.exp2(y)
    assert ratio.min() > 2.0 ** (-1 / 16) - 1e-9 and ratio.max() < 1.0615 * 2.0 ** (1 / 16)
    # without the byte rounding: 2^floor(y) (1 + frac) / 2^y in [1, 1.0615]
    f = torch.linspace(0, 1, 1001, dtype=torch.float64)
    assert ((1 + f) / torch.exp2(f)).max() < 1.0615


def test_workspace_layout_sizes_and_coverage():
    g = torch.Generator().manual_seed(1)
    B, H, Lq, Lk = 2, 3, 100, 130
    q, k, v = (torch.randn(B, L, H, 128, generator=g).bfloat16() for L in (Lq, Lk, Lk))
    Q = A.quantise(q, k, v)
    ws, known = A.pack_workspace(Q)
    assert ws.size == A.workspace_bytes(B, H, Lq, Lk) and ws.size % 256 == 0
    nTk = 3
    assert Q["v8"].shape == (B, H, nTk * 64, 128) and Q["vs"].shape == (B, H, nTk * 2, 128)
    assert (Q["v8"][:, :, Lk:] == 0).all()
    # every key of a tile appears exactly once in a V^T image row
    keys = sorted(kb * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh for hh in range(2) for kb in range(2) for j in range(16))
    assert keys == list(range(64))


def test_mode_distance_from_exact_attention_and_masked_keys():
    g = torch.Generator().manual_seed(2)
    B, H, Lq, Lk = 1, 2, 96, 400
    q, k, v = (torch.randn(B, L, H, 128, generator=g).bfloat16() for L in (Lq, Lk, Lk))
    exact = A.exact_attention(q, k, v, k_len=390)
    for pmode in (1, 0):
        o32, o16 = A.attention(q, k, v, k_len=390, pmode=pmode)
        e = rel(o32, exact)
        assert 1e-2 < e < 8e-2, e                                             # three e4m3 operands: ~5.5e-2 on gaussian data
        k2, v2 = k.clone(), v.clone()
        k2[:, 390:] = 77.0
        v2[:, 390:] = -1e4
        assert torch.equal(A.attention(q, k2, v2, k_len=390, pmode=pmode)[0], o32)    # keys past k_len never reach a sum
        assert o16.dtype == torch.bfloat16


def test_one_hot_rows_and_tiles_far_below_the_reference():
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(1, L, 1, 128, generator=g).bfloat16() for L in (40, 256, 256))
    k[0, 200, 0] = q[0, 7, 0] * 4.0                                           # row 7: one key hundreds of bits above everything else
    for pmode in (1, 0):
        o32, _ = A.attention(q, k, v, pmode=pmode)
        assert rel(o32[0, 7, 0], v[0, 200, 0].float()) < 5e-2                  # e4m3 of v itself: 2^-4 per element at worst
        assert rel(o32, A.exact_attention(q, k, v)) < 8e-2

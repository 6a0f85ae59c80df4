#!/usr/bin/env python3
"""Generate tests/golden/*.safetensors by running the REFERENCE's own model code.

Runs only in the build container (needs /root/reference).  The reference's
versecrafter/models/wan_transformer3d.py and wan_transformer3d_versecrafter.py are imported
as they lie, with in-process stand-ins (tests/golden/_ref_stubs.py) for the third-party
modules that are absent here.  What is recorded is data only: seeded inputs and the
reference's outputs (fp32, CPU).  Weights are NOT stored: they are re-derived from
oracle.wan_oracle.random_weights(cfg, seed) (numpy RandomState stream) by both this script
and the tests.

    python tests/golden/make_golden.py

Fixtures
    ops_small.safetensors      per-function I/O of the reference (WT.py)
    forward_tiny.safetensors   VerseCrafterWanTransformer3DModel.forward, tiny config
    teacache_trace.safetensors _process_teacache_skip_logic decisions over a t-sweep
"""
import os
import sys
import warnings

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np
import torch
from safetensors.torch import save_file

import _ref_stubs

_ref_stubs.install()

from versecrafter.models import wan_transformer3d as WT                      # noqa: E402
from versecrafter.models.wan_transformer3d_versecrafter import \
    VerseCrafterWanTransformer3DModel                                        # noqa: E402

from oracle import wan_oracle as O                                           # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)

TINY = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=4, text_dim=64, text_len=48,
            geoada_in_dim=128, in_dim=16, out_dim=16, freq_dim=256)


def rs_randn(rs, *shape):
    return torch.from_numpy(rs.standard_normal(shape).astype(np.float32))


def build_ref_model(cfg_kwargs, seed):
    m = VerseCrafterWanTransformer3DModel(**cfg_kwargs).eval().float()
    ocfg = O.Config(**{k: v for k, v in cfg_kwargs.items()})
    W = O.random_weights(ocfg, seed)
    missing, unexpected = m.load_state_dict(W, strict=True)
    return m, ocfg, W


def ops_small():
    rs = np.random.RandomState(11)
    out = {}
    # --- sinusoidal_embedding_1d (WT.py:39-49)
    t = torch.tensor([999.0, 500.0, 37.0, 0.0])
    out["sinus.t"] = t
    out["sinus.out"] = WT.sinusoidal_embedding_1d(256, t)
    # --- rope table (WT.py:52-60, 788-795) : selected rows as real/imag
    d = 128
    freqs = torch.cat([WT.rope_params(1024, d - 4 * (d // 6)), WT.rope_params(1024, 2 * (d // 6)),
                       WT.rope_params(1024, 2 * (d // 6))], dim=1)
    rows = torch.tensor(list(range(0, 48)) + [100, 511, 1000, 1023])
    out["rope.rows"] = rows
    out["rope.table_rows"] = torch.view_as_real(freqs[rows]).contiguous()
    # --- riflex table (WT.py:63-121)
    rf = WT.get_1d_rotary_pos_embed_riflex(1024, d - 4 * (d // 6), use_real=False, k=6, L_test=66,
                                           L_test_scale=4.886)
    out["riflex.table_rows"] = torch.view_as_real(rf[rows]).contiguous()
    # --- rope_apply (WT.py:145-172): B=2, ragged grids, padded tail passes through
    x = rs_randn(rs, 2, 80, 2, 128)
    grids = torch.tensor([[3, 4, 6], [2, 5, 8]])
    out["rope_apply.x"] = x
    out["rope_apply.grids"] = grids
    out["rope_apply.out"] = WT.rope_apply(x, grids, freqs)
    # --- WanRMSNorm (WT.py:307-323)
    n = WT.WanRMSNorm(256, eps=1e-6)
    n.weight.copy_(1 + 0.1 * rs_randn(rs, 256))
    xr = rs_randn(rs, 2, 9, 256) * 3
    out["rms.x"], out["rms.w"], out["rms.out"] = xr, n.weight.clone(), n(xr)
    # --- WanLayerNorm (WT.py:326-346) plain and affine
    ln = WT.WanLayerNorm(256, 1e-6)
    out["ln.x"] = xr
    out["ln.out"] = ln(xr)
    lna = WT.WanLayerNorm(256, 1e-6, elementwise_affine=True)
    lna.weight.copy_(1 + 0.1 * rs_randn(rs, 256))
    lna.bias.copy_(0.1 * rs_randn(rs, 256))
    out["lna.w"], out["lna.b"], out["lna.out"] = lna.weight.clone(), lna.bias.clone(), lna(xr)
    # --- unpatchify (WT.py:1127-1150)
    m, _, _ = build_ref_model(TINY, 3)
    u = rs_randn(rs, 2, 80, 64)
    up = m.unpatchify(u, grids)
    out["unpatchify.x"] = u
    out["unpatchify.out0"], out["unpatchify.out1"] = up[0].contiguous(), up[1].contiguous()
    # --- one WanAttentionBlock (WT.py:564-611) with key mask (seq_lens < L)
    blk = m.blocks[1]
    xb = rs_randn(rs, 2, 80, 256)
    e0 = rs_randn(rs, 2, 6, 256) * 0.5
    ctx = rs_randn(rs, 2, 48, 256)
    seq_lens = torch.tensor([72, 80])
    yb = WT.WanAttentionBlock.forward(blk, xb, e0, seq_lens, grids, freqs, ctx, None, dtype=torch.float32)
    out["block.x"], out["block.e0"], out["block.ctx"] = xb, e0, ctx
    out["block.seq_lens"], out["block.out"] = seq_lens, yb
    # --- Head (WT.py:631-644)
    e = rs_randn(rs, 2, 256)
    out["head.e"], out["head.out"] = e, m.head(xb, e)
    save_file({k: v.contiguous().clone() for k, v in out.items()}, os.path.join(HERE, "ops_small.safetensors"))
    print("ops_small:", {k: tuple(v.shape) for k, v in out.items()})


def forward_tiny():
    rs = np.random.RandomState(2025)
    m, ocfg, W = build_ref_model(TINY, 7)
    out = {}
    # case A: B=2 (CFG pair), same latent shape, seq_len == L  (the pipeline's case)
    T, h, w = 3, 8, 12
    x = rs_randn(rs, 2, 16, T, h, w)
    g = torch.cat([rs_randn(rs, 2, 64, T, h, w),
                   torch.from_numpy((rs.uniform(size=(2, 64, T, h, w)) < 0.5).astype(np.float32))], dim=1)
    ctx = [rs_randn(rs, 20, 64), rs_randn(rs, 33, 64)]
    t = torch.tensor([875.0, 875.0])
    seq_len = T * (h // 2) * (w // 2)
    for scale, tag in ((1.0, "A"), (0.6, "A06")):
        y = m(x, t, g, ctx, seq_len, geoada_context_scale=scale)
        out[f"{tag}.out"] = y
    out["A.x"], out["A.geoada"], out["A.ctx0"], out["A.ctx1"], out["A.t"] = x, g, ctx[0], ctx[1], t
    out["A.seq_len"] = torch.tensor([seq_len])
    # case B: seq_len > L (zero-padded tail tokens, masked keys: WT.py:198-201, 398)
    seq_len_b = seq_len + 8
    out["B.out"] = m(x, t, g, ctx, seq_len_b)
    out["B.seq_len"] = torch.tensor([seq_len_b])
    # case C: TeaCache residual path (VC.py:384-411): calc step stores residual, skip step re-adds it
    m.enable_teacache([1.0, 0.0], num_steps=3, rel_l1_thresh=1e9, num_skip_start_steps=1, offload=False)
    t1, t2 = torch.tensor([900.0, 900.0]), torch.tensor([880.0, 880.0])
    y1 = m(x, t1, g, ctx, seq_len)
    assert m.should_calc
    out["C.residual"] = m.teacache.previous_residual_cond.clone()
    x2 = x + 0.1 * rs_randn(rs, *x.shape)
    y2 = m(x2, t2, g, ctx, seq_len)
    assert not m.should_calc
    out["C.t1"], out["C.t2"], out["C.x2"], out["C.out1"], out["C.out2"] = t1, t2, x2, y1, y2
    m.disable_teacache()
    save_file({k: v.contiguous().clone() for k, v in out.items()}, os.path.join(HERE, "forward_tiny.safetensors"))
    print("forward_tiny:", {k: tuple(v.shape) for k, v in out.items()},
          "absmax out", out["A.out"].abs().max().item())


def teacache_trace():
    """_process_teacache_skip_logic (WT.py:205-245) with the CLI's coefficients / threshold /
    skip-start (CLI.py:104-116, 305-313) over a synthetic, slowly drifting e0 sequence that
    produces a mix of calc / skip decisions (random-weight time MLPs change e0 too fast to skip)."""
    rs = np.random.RandomState(5)
    coeffs = [8.10705460e+03, 2.13393892e+03, -3.72934672e+02, 1.66203073e+01, -4.17769401e-02]
    n = 30
    base = rs_randn(rs, 2, 6, 256)
    drift = rs_randn(rs, 2, 6, 256)
    tc = _ref_stubs.TeaCache(coeffs, n, rel_l1_thresh=0.10, num_skip_start_steps=5, offload=False)
    e0s, decisions, accs = [], [], []
    t = torch.tensor([500.0, 500.0])
    for i in range(n):
        e0 = base + (0.004 * i + 0.00004 * i * i) * drift
        d = WT._process_teacache_skip_logic(tc, e0, t, True)
        e0s.append(e0)
        decisions.append(int(d))
        accs.append(float(tc.accumulated_rel_l1_distance))
        tc.cnt += 1
    out = {"e0": torch.stack(e0s), "decisions": torch.tensor(decisions),
           "acc": torch.tensor(accs, dtype=torch.float64),
           "coeffs": torch.tensor(coeffs, dtype=torch.float64)}
    save_file(out, os.path.join(HERE, "teacache_trace.safetensors"))
    print("teacache decisions:", decisions)


if __name__ == "__main__":
    ops_small()
    forward_tiny()
    teacache_trace()

"""Pins oracle/wan_oracle.py (the CPU restatement) against golden vectors recorded from the
reference's own WT.py / VC.py (tests/golden/make_golden.py).  CPU only."""
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import wan_oracle as O

TINY = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=4, text_dim=64, text_len=48,
            geoada_in_dim=128, in_dim=16, out_dim=16, freq_dim=256)


@pytest.fixture(scope="module")
def ops(golden_dir):
    return load_file(os.path.join(golden_dir, "ops_small.safetensors"))


@pytest.fixture(scope="module")
def fwd(golden_dir):
    return load_file(os.path.join(golden_dir, "forward_tiny.safetensors"))


def close(a, b, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(a.double(), b.double(), rtol=rtol, atol=atol)


def test_sinusoidal(ops):
    close(O.sinusoidal_embedding_1d(256, ops["sinus.t"]), ops["sinus.out"], 1e-12, 1e-12)


def test_rope_table(ops):
    tab = O.rope_table(128)
    assert tab.shape == (1024, 64) and tab.dtype == torch.complex128
    close(torch.view_as_real(tab[ops["rope.rows"]]), ops["rope.table_rows"], 1e-13, 1e-13)


def test_rope_table_riflex(ops):
    tab = O.rope_table_riflex(128, k=6, L_test=66, L_test_scale=4.886)
    close(torch.view_as_real(tab[ops["rope.rows"], :22]), ops["riflex.table_rows"], 1e-13, 1e-13)


def test_rope_apply(ops):
    y = O.rope_apply(ops["rope_apply.x"], ops["rope_apply.grids"].tolist(), O.rope_table(128))
    close(y, ops["rope_apply.out"], 1e-6, 1e-6)


def test_rope_apply_sp_offset(ops):
    """Chunked application with the rank offset equals the unchunked result (VC.py:366-367)."""
    x, grids = ops["rope_apply.x"], ops["rope_apply.grids"].tolist()
    tab = O.rope_table(128)
    for P in (2, 4, 5):
        Lp = x.shape[1] // P
        parts = [O.rope_apply(x[:, r * Lp:(r + 1) * Lp], grids, tab, token_offset=r * Lp) for r in range(P)]
        close(torch.cat(parts, 1), ops["rope_apply.out"], 1e-6, 1e-6)


def test_rms_norm(ops):
    close(O.rms_norm(ops["rms.x"], ops["rms.w"], 1e-6), ops["rms.out"])


def test_layer_norm(ops):
    close(O.layer_norm(ops["ln.x"]), ops["ln.out"])
    close(O.layer_norm(ops["ln.x"], ops["lna.w"], ops["lna.b"]), ops["lna.out"])


def test_unpatchify(ops):
    up = O.unpatchify(ops["unpatchify.x"], ops["rope_apply.grids"].tolist())
    assert torch.equal(up[0], ops["unpatchify.out0"]) and torch.equal(up[1], ops["unpatchify.out1"])


def test_block_and_head(ops):
    cfg = O.Config(**TINY)
    W = O.random_weights(cfg, 3)
    y = O.attention_block(W, "blocks.1.", ops["block.x"], ops["block.e0"], ops["block.seq_lens"].tolist(),
                          ops["rope_apply.grids"].tolist(), O.rope_table(128), ops["block.ctx"], cfg.num_heads)
    close(y, ops["block.out"], 1e-4, 1e-4)
    close(O.head(W, ops["block.x"], ops["head.e"]), ops["head.out"], 1e-4, 1e-4)


def _inputs(fwd):
    return fwd["A.x"], fwd["A.geoada"], [fwd["A.ctx0"], fwd["A.ctx1"]], fwd["A.t"]


def test_forward_tiny(fwd):
    cfg = O.Config(**TINY)
    W = O.random_weights(cfg, 7)
    x, g, ctx, t = _inputs(fwd)
    close(O.forward(W, cfg, x, t, g, ctx, int(fwd["A.seq_len"])), fwd["A.out"], 2e-4, 2e-4)
    close(O.forward(W, cfg, x, t, g, ctx, int(fwd["A.seq_len"]), geoada_context_scale=0.6),
          fwd["A06.out"], 2e-4, 2e-4)
    close(O.forward(W, cfg, x, t, g, ctx, int(fwd["B.seq_len"])), fwd["B.out"], 2e-4, 2e-4)


def test_forward_teacache_residual(fwd):
    cfg = O.Config(**TINY)
    W = O.random_weights(cfg, 7)
    x, g, ctx, _ = _inputs(fwd)
    L = int(fwd["A.seq_len"])
    y1, res = O.forward(W, cfg, x, fwd["C.t1"], g, ctx, L, return_residual=True)
    close(y1, fwd["C.out1"], 2e-4, 2e-4)
    close(res, fwd["C.residual"], 2e-4, 2e-4)
    y2 = O.forward(W, cfg, fwd["C.x2"], fwd["C.t2"], g, ctx, L, run_main_blocks=False, residual=res)
    close(y2, fwd["C.out2"], 2e-4, 2e-4)


def test_teacache_gate(golden_dir):
    tr = load_file(os.path.join(golden_dir, "teacache_trace.safetensors"))
    st = O.TeaCacheState(tr["coeffs"].tolist(), 30, 0.10, num_skip_start_steps=5)
    dec, acc = [], []
    for e0 in tr["e0"]:
        dec.append(int(O.teacache_gate(st, e0)))
        acc.append(st.accumulated)
        st.cnt += 1
    assert dec == tr["decisions"].tolist()
    close(torch.tensor(acc, dtype=torch.float64), tr["acc"], 1e-6, 1e-9)
    assert 0 in dec and 1 in dec


def test_bf16_mode_is_close_to_fp32(fwd):
    """The bf16-rounding mode (used to size GPU tolerances) stays near the fp32 result."""
    cfg = O.Config(**TINY)
    W = {k: v.bfloat16().float() for k, v in O.random_weights(cfg, 7).items()}
    x, g, ctx, t = _inputs(fwd)
    a = O.forward(W, cfg, x, t, g, ctx, int(fwd["A.seq_len"]))
    b = O.forward(W, cfg, x, t, g, ctx, int(fwd["A.seq_len"]), mode="bf16")
    rel = (a - b).norm() / a.norm()
    assert rel < 3e-2, rel

"""Host-side sampler logic (CPU): UniPC scheduler and TeaCache gate of the product package against the
oracle restatements.  Both sides restate third-party code that is absent from the reference tree
(videox_fun @ unknown commit): parity unpinned; these tests pin the two restatements to each other and to
the reference's _process_teacache_skip_logic trace (tests/golden/teacache_trace.safetensors)."""
import os

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

from oracle import unipc_oracle as U
from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
from versecrafter_amd.utils.teacache import TeaCache


@pytest.mark.parametrize("n,shift", [(4, 16.0), (10, 5.0), (30, 16.0)])
def test_unipc_matches_oracle(n, shift):
    rs = np.random.RandomState(n)
    sch = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    sch.set_timesteps(n, device="cpu", shift=shift)
    orc = U.UniPCOracle(n, shift)
    assert sch.timesteps.tolist() == orc.timesteps.tolist()
    assert sch.timesteps.dtype == torch.int64 and sch.order == 1
    x = rs.standard_normal((1, 16, 2, 4, 4))
    xt = torch.from_numpy(x).float()
    for i, t in enumerate(sch.timesteps):
        v = rs.standard_normal(x.shape) * 0.5 + 0.1 * x
        xt = sch.step(torch.from_numpy(v).float(), t, xt, return_dict=False)[0]
        x = orc.step(v, x)
        np.testing.assert_allclose(xt.double().numpy(), x, rtol=2e-4, atol=2e-4, err_msg=f"step {i}")
    assert np.isfinite(x).all()


def test_unipc_first_and_last_step_closed_form():
    """Order-1 predictor = one Euler step in sigma for flow matching: x + (s_next - s) v; last step lands on x0."""
    sch = FlowUniPCMultistepScheduler(shift=1)
    sch.set_timesteps(1, device="cpu", shift=16.0)
    x, v = torch.randn(1, 4, 2, 2), torch.randn(1, 4, 2, 2)
    s0 = float(sch.sigmas[0])
    out = sch.step(v, sch.timesteps[0], x, return_dict=False)[0]
    torch.testing.assert_close(out, x - s0 * v, rtol=1e-5, atol=1e-5)


def test_teacache_gate_matches_reference_trace(golden_dir):
    tr = load_file(os.path.join(golden_dir, "teacache_trace.safetensors"))
    tc = TeaCache(tr["coeffs"].tolist(), 30, rel_l1_thresh=0.10, num_skip_start_steps=5, offload=False)
    dec, acc = [], []
    for e0 in tr["e0"]:
        dec.append(int(tc.gate(e0)))
        acc.append(float(tc.accumulated_rel_l1_distance))
        tc.cnt += 1
    assert dec == tr["decisions"].tolist()
    np.testing.assert_allclose(acc, tr["acc"].numpy(), rtol=1e-6, atol=1e-9)


def test_teacache_validates_arguments():
    with pytest.raises(ValueError):
        TeaCache([1.0], 0)
    with pytest.raises(ValueError):
        TeaCache([1.0], 10, rel_l1_thresh=-1)

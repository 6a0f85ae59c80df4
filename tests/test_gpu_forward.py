"""End-to-end parity of the HIP engine (through the drop-in model class and the C ABI) against the golden
vectors recorded from the reference (tests/golden/forward_tiny.safetensors) and the CPU oracle.

Tolerance: the engine computes in bf16 with fp32 accumulation, like the reference under bf16 autocast.  Its
distance to the fp32 golden output must be (a) below 3e-2 relative L2 and (b) no more than 3x the distance of
the oracle's own bf16-rounding mode (the reference's rounding points) from the same golden."""
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import wan_oracle as O

pytestmark = pytest.mark.gpu

TINY = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=4, text_dim=64, text_len=48,
            geoada_in_dim=128, in_dim=16, out_dim=16, freq_dim=256)


@pytest.fixture(scope="module")
def fwd(golden_dir):
    return load_file(os.path.join(golden_dir, "forward_tiny.safetensors"))


@pytest.fixture(scope="module")
def model():
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    cfg = O.Config(**TINY)
    m = VerseCrafterWanTransformer3DModel(**TINY)
    m.load_state_dict(O.random_weights(cfg, 7))
    return m.to(torch.bfloat16).to("cuda")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def bf16_reference_error(fwd, **kw):
    cfg = O.Config(**TINY)
    W = {k: v.bfloat16().float() for k, v in O.random_weights(cfg, 7).items()}
    x, g = fwd["A.x"].bfloat16().float(), fwd["A.geoada"].bfloat16().float()
    ctx = [fwd["A.ctx0"].bfloat16().float(), fwd["A.ctx1"].bfloat16().float()]
    return O.forward(W, cfg, x, fwd["A.t"], g, ctx, mode="bf16", **kw)


def run(model, fwd, seq_len, x=None, t=None, scale=1.0):
    x = fwd["A.x"] if x is None else x
    t = fwd["A.t"] if t is None else t
    ctx = [fwd["A.ctx0"].bfloat16().cuda(), fwd["A.ctx1"].bfloat16().cuda()]
    out = model(x.bfloat16().cuda(), t.cuda(), fwd["A.geoada"].bfloat16().cuda(), ctx, seq_len,
                geoada_context_scale=scale)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("case,scale", [("A", 1.0), ("A06", 0.6), ("B", 1.0)])
def test_forward_matches_reference_golden(model, fwd, case, scale):
    seq_len = int(fwd["B.seq_len" if case == "B" else "A.seq_len"])
    got = run(model, fwd, seq_len, scale=scale)
    want = fwd[f"{case}.out"]
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    e_hip = rel(got, want)
    e_ref = rel(bf16_reference_error(fwd, seq_len=seq_len, geoada_context_scale=scale), want)
    print(f"case {case}: engine rel L2 {e_hip:.4g}; bf16-reference rel L2 {e_ref:.4g}")
    assert e_hip < 3e-2
    assert e_hip < 3 * e_ref + 2e-3


def test_forward_is_deterministic(model, fwd):
    a = run(model, fwd, int(fwd["A.seq_len"]))
    b = run(model, fwd, int(fwd["A.seq_len"]))
    assert torch.equal(a, b)


def test_teacache_residual_path(model, fwd):
    """VC.py:384-411: a calc step stores x_out - x_in; a skipped step re-adds it instead of running the blocks."""
    model.enable_teacache([1.0, 0.0], num_steps=3, rel_l1_thresh=1e9, num_skip_start_steps=1, offload=False)
    try:
        L = int(fwd["A.seq_len"])
        y1 = run(model, fwd, L, t=fwd["C.t1"])
        assert model.should_calc
        y2 = run(model, fwd, L, x=fwd["C.x2"], t=fwd["C.t2"])
        assert not model.should_calc
        assert rel(y1, fwd["C.out1"]) < 3e-2
        assert rel(y2, fwd["C.out2"]) < 3e-2
    finally:
        model.disable_teacache()


def _fresh_model():
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    m = VerseCrafterWanTransformer3DModel(**TINY)
    m.load_state_dict(O.random_weights(O.Config(**TINY), 7))
    return m.to(torch.bfloat16).to("cuda")


def test_teacache_with_cfg_skip_crosses_the_batch_switch(fwd):
    """TeaCache is on by default in the CLI and cfg_skip is a CLI knob (CLI.py:104, 122): when cfg_skip drops the
    unconditional half mid-sampling the batch shrinks 2 -> 1, and a skipped step must re-add the LAST sample of the stored
    residual (previous_residual[-x.size(0):], VC.py:396).  Expected value: the conditional half of the same sampler run
    without cfg_skip (same gate decisions: the gate only sees the time embedding)."""
    L = int(fwd["A.seq_len"])
    xs = [fwd["A.x"], fwd["C.x2"], fwd["A.x"] * 0.5, fwd["C.x2"] * 0.75]
    ts = [fwd["C.t1"], fwd["C.t2"], fwd["C.t2"] - 40.0, fwd["C.t2"] - 80.0]
    a, b = _fresh_model(), _fresh_model()
    for m in (a, b):                                    # step 0 computes, every later step is skipped (huge threshold)
        m.enable_teacache([1.0, 0.0], num_steps=4, rel_l1_thresh=1e9, num_skip_start_steps=1, offload=False)
    a.enable_cfg_skip(0.5, 4)                           # steps 2, 3: conditional half only
    for i in range(4):
        a.current_steps = b.current_steps = i
        ya = run(a, fwd, L, x=xs[i], t=ts[i])
        yb = run(b, fwd, L, x=xs[i], t=ts[i])
        assert a.should_calc == b.should_calc == (i == 0)
        if i < 2:
            assert torch.equal(ya, yb)
        else:
            assert torch.equal(ya[0], ya[1]) and torch.equal(ya[1], yb[1]), i


def test_graph_replay_of_a_skipped_step_follows_the_residual_slot_across_the_batch_switch(fwd, monkeypatch):
    """Advisor finding (round 2): a captured USE_RESIDUAL graph bakes in the slot's read offset (resid_B - B) rows
    (previous_residual[-B:], VC.py:396).  Store at B = 2, cfg_skip drops to B = 1, two skipped steps capture a graph that
    reads the LAST sample of the pair; a calc step then stores at B = 1 (no reallocation, so nothing dropped the graph) and
    the next skipped step must read sample 0 of the new residual, not the stale half of the old one.  Eager recomputes the
    pointer; the graph path must equal it bit for bit on every step."""
    L = int(fwd["A.seq_len"])
    xs = [fwd["A.x"], fwd["C.x2"], fwd["A.x"] * 0.5, fwd["C.x2"] * 0.75, fwd["A.x"] * -0.3, fwd["C.x2"] * 0.4]
    ts = [fwd["C.t1"], fwd["C.t2"], fwd["C.t2"] - 40.0, fwd["C.t2"] - 80.0, fwd["C.t2"] - 120.0, fwd["C.t2"] - 160.0]
    calc = [True, False, False, True, False, False]          # B = 2 store | B = 1: skip, skip (capture), calc + store, skip, skip
    monkeypatch.setenv("VC_GRAPH", "0")
    eager = _fresh_model()
    monkeypatch.setenv("VC_GRAPH", "1")
    graph = _fresh_model()
    outs = {}
    for name, m in (("eager", eager), ("graph", graph)):
        m.enable_teacache([1.0, 0.0], num_steps=len(calc), rel_l1_thresh=1e9, num_skip_start_steps=1, offload=False)
        m.enable_cfg_skip(5.0 / 6.0, len(calc))              # from step 1 on: conditional half only
        it = iter(calc)
        tc = m.teacache

        def gate(e0, tc=tc, it=it):
            tc.previous_modulated_input = e0
            tc.should_calc = next(it)
            return tc.should_calc
        tc.gate = gate
        ys = []
        for i in range(len(calc)):
            m.current_steps = i
            ys.append(run(m, fwd, L, x=xs[i], t=ts[i]))
            assert m.should_calc == calc[i]
        outs[name] = ys
    for i, (a, b) in enumerate(zip(outs["graph"], outs["eager"])):
        assert torch.equal(a, b), f"step {i}: graph replay differs from the eager engine"
    assert not torch.equal(outs["eager"][4], outs["eager"][2])


def test_teacache_keeps_separate_cond_and_uncond_residuals(fwd):
    """cond_flag=False forwards (VC.py:391-394, 408-411) store and re-use previous_residual_uncond, never the conditional
    residual; their gate decision is the conditional call's (WT.py:244-245)."""
    L = int(fwd["A.seq_len"])
    geo = fwd["A.geoada"].bfloat16().cuda()
    ctx = [fwd["A.ctx0"].bfloat16().cuda(), fwd["A.ctx1"].bfloat16().cuda()]
    xs = [fwd["A.x"].bfloat16().cuda(), fwd["C.x2"].bfloat16().cuda()]
    ts = [fwd["C.t1"].cuda(), fwd["C.t2"].cuda()]
    both, only_c, only_u = _fresh_model(), _fresh_model(), _fresh_model()
    for m in (both, only_c, only_u):
        m.enable_teacache([1.0, 0.0], num_steps=3, rel_l1_thresh=1e9, num_skip_start_steps=1, offload=False)
    for i in range(2):
        # the reference's un-batched CFG: a conditional forward, then an unconditional one with cond_flag=False
        yc = both(xs[i][1:], ts[i][1:], geo[1:], ctx[1:], L, cond_flag=True)
        yu = both(xs[i][:1], ts[i][:1], geo[:1], ctx[:1], L, cond_flag=False)
        assert both.should_calc == (i == 0)
        wc = only_c(xs[i][1:], ts[i][1:], geo[1:], ctx[1:], L)
        wu = only_u(xs[i][:1], ts[i][:1], geo[:1], ctx[:1], L)
        torch.cuda.synchronize()
        assert torch.equal(yc, wc), i
        assert torch.equal(yu, wu), i
    assert not torch.equal(yc, yu)


def test_enable_riflex_reaches_the_engine_and_is_reversible(fwd):
    """enable_riflex (WT.py:873-888, CLI.py:315-317) replaces one temporal frequency of the RoPE table (checked against the
    reference's table in test_oracle_golden.py).  On a 3-frame clip its effect on the output (2e-3 relative) is below the
    bf16 noise of a forward, so the engine check is structural: the table reaches the kernels (output changes, equals a
    model that had RIFLEx enabled from the start bit for bit, stays within the oracle bound) and disable_riflex restores it."""
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    cfg = O.Config(**TINY)
    W = O.random_weights(cfg, 7)

    def make():
        m = VerseCrafterWanTransformer3DModel(**TINY)
        m.load_state_dict(W)
        return m.to(torch.bfloat16).to("cuda")

    m = make()
    seq_len = int(fwd["A.seq_len"])
    base = run(m, fwd, seq_len)
    m.enable_riflex(k=1, L_test=3)
    got = run(m, fwd, seq_len)
    assert not torch.equal(got, base)
    fresh = make()
    fresh.enable_riflex(k=1, L_test=3)
    assert torch.equal(run(fresh, fwd, seq_len), got)
    Wb = {k: v.bfloat16().float() for k, v in W.items()}
    x, g = fwd["A.x"].bfloat16().float(), fwd["A.geoada"].bfloat16().float()
    ctx = [fwd["A.ctx0"].bfloat16().float(), fwd["A.ctx1"].bfloat16().float()]
    want = O.forward(Wb, cfg, x, fwd["A.t"], g, ctx, seq_len,
                     freqs=O.rope_table_riflex(TINY["dim"] // TINY["num_heads"], 1, 3, 4.886))
    assert rel(got, want) < 3e-2
    m.disable_riflex()
    assert torch.equal(run(m, fwd, seq_len), base)


def test_shared_cfg_prefix_is_bit_identical(model, fwd, monkeypatch):
    """The sampler's CFG pair (PIPE.py:878-887) duplicates latents, timestep and control maps; the engine then computes the
    prompt-independent prefix of block 0 of both chains once (VC_FWD_SHARED_CFG_INPUT).  Detected from the tensors, and the
    result must equal the full computation bit for bit; different samples must not take the shortcut."""
    from versecrafter_amd import _lib
    seq_len = int(fwd["A.seq_len"])
    x = fwd["A.x"][:1].repeat(2, 1, 1, 1, 1)
    geo = fwd["A.geoada"][:1].repeat(2, 1, 1, 1, 1).bfloat16().cuda()
    t = fwd["A.t"][:1].repeat(2)
    ctx = [fwd["A.ctx0"].bfloat16().cuda(), fwd["A.ctx1"].bfloat16().cuda()]
    fast = model(x.bfloat16().cuda(), t.cuda(), geo, ctx, seq_len)
    assert model._last_flags & _lib.VC_FWD_SHARED_CFG_INPUT
    monkeypatch.setenv("VC_NO_SHARED_CFG", "1")
    full = model(x.bfloat16().cuda(), t.cuda(), geo, ctx, seq_len)
    assert not (model._last_flags & _lib.VC_FWD_SHARED_CFG_INPUT)
    monkeypatch.delenv("VC_NO_SHARED_CFG")
    torch.cuda.synchronize()
    assert torch.equal(fast, full) and not torch.equal(fast[0], fast[1])          # the prompts differ
    run(model, fwd, seq_len)                                                       # golden inputs: the samples differ
    assert not (model._last_flags & _lib.VC_FWD_SHARED_CFG_INPUT)
    x2 = x.clone()
    x2[1, 0, 0, 0, 0] += 1.0
    model(x2.bfloat16().cuda(), t.cuda(), geo, ctx, seq_len)
    assert not (model._last_flags & _lib.VC_FWD_SHARED_CFG_INPUT)


def test_cfg_skip_runs_conditional_half_only(model, fwd):
    """cfg_skip (third-party decorator bound at WT.py:850-871; CLI.py:313): for the last `ratio` of the steps only the
    conditional half is computed and returned twice."""
    seq_len = int(fwd["A.seq_len"])
    full = run(model, fwd, seq_len)
    model.enable_cfg_skip(0.5, 4)
    try:
        model.current_steps = 0
        assert torch.equal(run(model, fwd, seq_len), full)            # early step: both halves
        model.current_steps = 3
        late = run(model, fwd, seq_len)                               # late step: conditional half, duplicated
        assert late.shape == full.shape and torch.equal(late[0], late[1])
        x1 = fwd["A.x"][1:].bfloat16().cuda()
        ctx1 = [fwd["A.ctx1"].bfloat16().cuda()]
        model.disable_cfg_skip()
        single = model(x1, fwd["A.t"][1:].cuda(), fwd["A.geoada"][1:].bfloat16().cuda(), ctx1, seq_len)
        assert torch.equal(late[1:], single)
    finally:
        model.disable_cfg_skip()
    assert torch.equal(run(model, fwd, seq_len), full)


def test_errors_mirror_reference(model, fwd):
    ctx = [fwd["A.ctx0"].bfloat16().cuda(), fwd["A.ctx1"].bfloat16().cuda()]
    x, g, t = fwd["A.x"].bfloat16().cuda(), fwd["A.geoada"].bfloat16().cuda(), fwd["A.t"].cuda()
    with pytest.raises(ValueError):          # assert seq_lens.max() <= seq_len  (WT.py:197)
        model(x, t, g, ctx, 10)
    with pytest.raises(ValueError):          # 112 != geoada_in_dim 128 (the demo_data state)
        model(x, t, g[:, :112], ctx, 72)
    with pytest.raises(TypeError):
        model(x.float(), t, g, ctx, 72)
    with pytest.raises(RuntimeError):        # no CPU path
        model(x.cpu(), t.cpu(), g.cpu(), [c.cpu() for c in ctx], 72)


def test_pipeline_three_step_trace_vs_oracle(model, fwd):
    """Sampler loop (PIPE.py:871-925) on the engine vs the oracle loop: oracle forward (fp32), oracle CFG combine,
    oracle UniPC.  bf16 latents drift a little more each step: rel-L2 bound 6e-2 after 3 steps."""
    import numpy as np
    from oracle import unipc_oracle as U
    from versecrafter_amd.pipeline import WanVerseCrafterPipeline
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    cfg = O.Config(**TINY)
    W = {k: v.bfloat16().float() for k, v in O.random_weights(cfg, 7).items()}
    T, h, w = 3, 8, 12
    g = torch.Generator().manual_seed(5)
    geo_lat = torch.randn(64, T, h, w, generator=g).bfloat16()
    mask_lat = (torch.rand(64, T, h, w, generator=g) < 0.5).to(torch.bfloat16)
    pe, ne = torch.randn(33, 64, generator=g).bfloat16(), torch.randn(20, 64, generator=g).bfloat16()
    lat0 = torch.randn(1, 16, T, h, w, generator=g).bfloat16()
    n, guidance = 3, 5.0
    pipe = WanVerseCrafterPipeline(transformer=model, scheduler=FlowUniPCMultistepScheduler(shift=1))
    got = pipe(prompt_embeds=[pe.cuda()], negative_prompt_embeds=[ne.cuda()], height=64, width=96,
               geoada_latents=[geo_lat.cuda()], mask_latents=[mask_lat.cuda()], num_inference_steps=n,
               guidance_scale=guidance, shift=16, latents=lat0.clone().cuda(), output_type="latent").videos
    torch.cuda.synchronize()
    orc = U.UniPCOracle(n, 16.0)
    x = lat0.double().numpy()
    geo = torch.cat([geo_lat, mask_lat], 0).float()
    for i in range(n):
        t = torch.tensor([float(orc.timesteps[i])] * 2)
        xin = torch.from_numpy(x).float().repeat(2, 1, 1, 1, 1)
        v = O.forward(W, cfg, xin, t, torch.stack([geo, geo]), [ne.float(), pe.float()], T * (h // 2) * (w // 2))
        x = orc.step(U.cfg_combine(v[0:1].double().numpy(), v[1:2].double().numpy(), guidance), x)
    r = rel(got[0], torch.from_numpy(x[0]))
    print("3-step sampler trace rel L2:", r)
    assert got.shape == (1, 16, T, h, w) and r < 6e-2


def test_pipeline_mask_video_front_end_equals_mask_latents(model, fwd):
    """__call__(mask_video=...) builds geoada_context with the HIP front-end kernel (PIPE.py:440-488); the result must be
    bit-identical to feeding mask_latents produced by the host restatement of geoada_encode_masks."""
    from versecrafter_amd.pipeline import WanVerseCrafterPipeline
    from versecrafter_amd.pipeline.pipeline_wan_versecrafter import geoada_encode_masks
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    F_, H_, W_ = 9, 64, 96
    T, h, w = 3, 8, 12
    g = torch.Generator().manual_seed(6)
    geo_lat = torch.randn(64, T, h, w, generator=g).bfloat16()
    mask_video = (torch.rand(1, 1, F_, H_, W_, generator=g) < 0.5).float()
    mask_video[:, :, 0] = 0                                                        # CLI.py:395
    pe, ne = torch.randn(33, 64, generator=g).bfloat16(), torch.randn(20, 64, generator=g).bfloat16()
    lat0 = torch.randn(1, 16, T, h, w, generator=g).bfloat16()
    outs = []
    for use_video in (True, False):
        pipe = WanVerseCrafterPipeline(transformer=model, scheduler=FlowUniPCMultistepScheduler(shift=1))
        kw = dict(mask_video=mask_video.cuda()) if use_video else dict(
            mask_latents=[m.to(torch.bfloat16).cuda() for m in geoada_encode_masks(torch.tile(mask_video, [1, 3, 1, 1, 1]))])
        outs.append(pipe(prompt_embeds=[pe.cuda()], negative_prompt_embeds=[ne.cuda()], height=H_, width=W_,
                         geoada_latents=[geo_lat.cuda()], num_inference_steps=2, guidance_scale=5.0, shift=16,
                         latents=lat0.clone().cuda(), output_type="latent", **kw).videos)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0], outs[1])


def test_cli_runs_end_to_end(tmp_path):
    """inference/versecrafter_inference.py with the reference's flags, synthetic inputs, tiny random model."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "inference", "versecrafter_inference.py"), "--rendering_maps_path", "x",
           "--prompt", "a car drives", "--input_image_path", "x.png", "--ulysses_degree", "1", "--ring_degree", "1",
           "--num_inference_steps", "8", "--sample_size", "64,96", "--video_length", "9", "--save_path", str(tmp_path),
           "--synthetic_inputs", "--synthetic_model", "tiny", "--num_skip_start_steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = load_file(os.path.join(str(tmp_path), "generated_latents_0.safetensors"))["latents"]
    assert out.shape == (1, 16, 3, 8, 12) and torch.isfinite(out).all()
    # the Wan2.2-style expert pair through the CLI (this build's flags): a second random model runs the high-noise steps
    r = subprocess.run(cmd + ["--synthetic_high_noise_expert", "--boundary", "0.9", "--shift", "12", "--save_path", str(tmp_path / "moe")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    two = load_file(os.path.join(str(tmp_path / "moe"), "generated_latents_0.safetensors"))["latents"]
    assert two.shape == out.shape and torch.isfinite(two).all() and not torch.equal(two, out)
    # fp8 linear layers through the CLI: eight sampler steps later the latents are still the bf16 run's, up to the quantisation drift
    r = subprocess.run(cmd + ["--fp8_linear", "1", "--save_path", str(tmp_path / "fp8")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    q = load_file(os.path.join(str(tmp_path / "fp8"), "generated_latents_0.safetensors"))["latents"]
    drift = ((q.float() - out.float()).norm() / out.float().norm()).item()
    assert torch.isfinite(q).all() and 1e-4 < drift < 0.25, drift


@pytest.mark.parametrize("name,dims,grid,e_ref_recorded", [
    # BASELINE.json config 1 geometry (1.3B width: d=1536, 12 heads, ffn 8960; 9 frames 320x512 -> latent [16,3,40,64],
    # 1920 tokens) with the depth cut to 4+2 blocks so that the CPU oracle finishes in seconds
    ("wan1.3b-width", dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=4), (3, 40, 64), 0.007611),
    # 14B width (d=5120, 40 heads, ffn 13824, text 512 x 4096) on a short clip: every kernel at its production shape in
    # the channel dimension (10-chunk row kernels, N=13824 GEMM, 40-head attention, 512-key cross attention)
    ("wan14b-width", dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=2), (2, 16, 24), 0.006594),
])
def test_forward_production_widths_vs_oracle(name, dims, grid, e_ref_recorded):
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    cfgk = dict(dims, geoada_in_dim=128, in_dim=16, out_dim=16, text_dim=4096, text_len=512, freq_dim=256)
    cfg = O.Config(**cfgk)
    g = torch.Generator().manual_seed(3)
    W = {}
    for k, shp in O.state_dict_shapes(cfg).items():       # cheap torch RNG (numpy RandomState is slow at this size)
        if k.endswith("modulation"):
            w = torch.randn(shp, generator=g) / cfg.dim ** 0.5
        elif "norm" in k and k.endswith("weight"):
            w = 1 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            w = 0.02 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for s_ in shp[1:]:
                fan_in *= s_
            a = (6.0 / (fan_in + shp[0])) ** 0.5
            w = (torch.rand(shp, generator=g) * 2 - 1) * a
        W[k] = w.bfloat16()
    T, h, w_ = grid
    x = torch.randn(2, 16, T, h, w_, generator=g).bfloat16()
    geo = torch.randn(2, 128, T, h, w_, generator=g).bfloat16()
    ctx = [torch.randn(60, 4096, generator=g).bfloat16(), torch.randn(77, 4096, generator=g).bfloat16()]
    t = torch.tensor([700.0, 700.0])
    L = T * (h // 2) * (w_ // 2)
    m = VerseCrafterWanTransformer3DModel(**cfgk)
    m.load_state_dict(W)
    m = m.to(torch.bfloat16).to("cuda")
    got = m(x.cuda(), t.cuda(), geo.cuda(), [c.cuda() for c in ctx], L)
    torch.cuda.synchronize()
    Wf = {k: v.float() for k, v in W.items()}
    args = (Wf, cfg, x.float(), t, geo.float(), [c.float() for c in ctx], L)
    want = O.forward(*args)
    e_hip = rel(got, want)
    # the oracle's own bf16-rounding mode on THESE seeded inputs, one constant per case (recorded with the oracle in this repo: 8.5 s
    # and 1.6 s of CPU each; VC_TEST_RECOMPUTE_BF16_REF=1 recomputes them).  Round 3 had put the 45-block cfg-1 figure (0.01444) here,
    # which made the 3x clause looser than the absolute bound; with the cases' own figures it binds again: 0.0248 / 0.0218 < 3e-2.
    e_ref = rel(O.forward(*args, mode="bf16"), want) if os.environ.get("VC_TEST_RECOMPUTE_BF16_REF") == "1" else e_ref_recorded
    print(f"{name}: engine rel L2 {e_hip:.4g}; bf16-reference rel L2 {e_ref:.4g}")
    assert torch.isfinite(got.float()).all()
    assert 3 * e_ref + 2e-3 < 3e-2                    # the clause below is the binding one
    assert e_hip < 3 * e_ref + 2e-3
    del m
    torch.cuda.empty_cache()


def _prod_weights(cfg, seed=3):
    """Random weights of a production-size config from the cheap torch RNG (numpy RandomState is slow at this size)."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, shp in O.state_dict_shapes(cfg).items():
        if k.endswith("modulation"):
            w = torch.randn(shp, generator=g) / cfg.dim ** 0.5
        elif "norm" in k and k.endswith("weight"):
            w = 1 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            w = 0.02 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for s_ in shp[1:]:
                fan_in *= s_
            a = (6.0 / (fan_in + shp[0])) ** 0.5
            w = (torch.rand(shp, generator=g) * 2 - 1) * a
        W[k] = w.bfloat16()
    return W, g


def test_cfg1_full_depth_forward_vs_oracle_and_four_step_sampler():
    """BASELINE.json config 1 at FULL depth: Wan2.1-1.3B + GeoAdapter (d=1536, 12 heads, ffn 8960, 30 main + 15 adapter
    blocks, 2.15 B parameters), 9 frames 320x512 -> latent [16,3,40,64], 1920 tokens, CFG pair.  One forward against the CPU
    oracle (0.018 PFLOP: the error growth over 45 blocks, same bound as the goldens: < 3e-2 and <= 3x the oracle's own
    bf16-rounding mode), then the config's 4-step sampler: finite, deterministic."""
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    from versecrafter_amd.pipeline import WanVerseCrafterPipeline
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    cfgk = dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, geoada_in_dim=128, in_dim=16, out_dim=16, text_dim=4096,
                text_len=512, freq_dim=256)
    cfg = O.Config(**cfgk)
    W, g = _prod_weights(cfg)
    T, h, w_ = 3, 40, 64
    x = torch.randn(2, 16, T, h, w_, generator=g).bfloat16()
    geo = torch.randn(2, 128, T, h, w_, generator=g).bfloat16()
    ctx = [torch.randn(60, 4096, generator=g).bfloat16(), torch.randn(77, 4096, generator=g).bfloat16()]
    t = torch.tensor([700.0, 700.0])
    L = O.seq_len_for((16, T, h, w_))
    assert L == 1920
    m = VerseCrafterWanTransformer3DModel(**cfgk, skip_init=True)
    m.load_state_dict(W)
    m = m.to(torch.bfloat16).to("cuda")
    got = m(x.cuda(), t.cuda(), geo.cuda(), [c.cuda() for c in ctx], L)
    torch.cuda.synchronize()
    Wf = {k: v.float() for k, v in W.items()}
    args = (Wf, cfg, x.float(), t, geo.float(), [c.float() for c in ctx], L)
    # The oracle's fp32 output for exactly these seeded weights and inputs is a recorded fixture (tests/golden/make_golden_cfg1_oracle.py:
    # 60-100 s of CPU time per run otherwise, a sixth of the GPU suite); the check sums prove the test regenerated the same numbers.
    # VC_TEST_RECOMPUTE_ORACLE=1 runs the oracle instead.
    gold = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_full_depth_oracle.safetensors"))
    bits = lambda v: int(v.contiguous().view(torch.int16).to(torch.int64).sum())      # exact, independent of the summation order
    same_inputs = (int(gold["x_bits"]) == bits(x) and int(gold["geo_bits"]) == bits(geo) and
                   int(gold["w_bits"]) == sum(bits(v) for v in W.values()))
    if os.environ.get("VC_TEST_RECOMPUTE_ORACLE") == "1" or not same_inputs:
        want = O.forward(*args)
        if same_inputs:
            assert rel(want, gold["want"]) < 1e-5, "the recorded oracle output no longer matches the oracle"    # thread count changes fp32 sum order
    else:
        want = gold["want"]
    e_hip = rel(got, want)
    # the oracle's own bf16-rounding mode on these seeded inputs: 0.01444 (recorded; a second full-depth CPU forward costs 85 s of the
    # suite -- VC_TEST_RECOMPUTE_BF16_REF=1 recomputes it)
    e_ref = rel(O.forward(*args, mode="bf16"), want) if os.environ.get("VC_TEST_RECOMPUTE_BF16_REF") == "1" else 0.01444
    print(f"cfg-1 full depth (30+15 blocks): engine rel L2 {e_hip:.4g}; bf16-reference rel L2 {e_ref:.4g}"
          f"{'' if same_inputs else ' (fixture inputs differ: oracle recomputed)'}")
    assert torch.isfinite(got.float()).all()
    assert e_hip < 3e-2 and e_hip < 3 * e_ref + 2e-3

    def sample():
        sch = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        pipe = WanVerseCrafterPipeline(transformer=m, scheduler=sch)
        out = pipe(prompt_embeds=[ctx[1].cuda()], negative_prompt_embeds=[ctx[0].cuda()], height=320, width=512,
                   num_frames=9, num_inference_steps=4, guidance_scale=5.0, latents=x[:1].cuda(), shift=16,
                   geoada_latents=[geo[0, :64].cuda()], mask_latents=[geo[0, 64:].cuda()], output_type="latent")
        torch.cuda.synchronize()
        return out.videos
    a, b = sample(), sample()
    assert a.shape == (1, 16, T, h, w_) and torch.isfinite(a.float()).all() and a.float().abs().max() > 0
    assert torch.equal(a, b)
    del m
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def wan14b():
    """Wan2.1-14B + GeoAdapter (21.9 B parameters, random, bf16, 43.7 GB) built once for the full-size property tests."""
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = VerseCrafterWanTransformer3DModel(geoada_in_dim=128, param_device=dev, param_dtype=torch.bfloat16,
                                          dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, skip_init=True)
    m.init_weights(zero_init_outputs=False)
    yield m
    del m
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def wan14b_high():
    """The HIGH-noise expert of a Wan2.2-style pair: a second Wan-14B + GeoAdapter (another 43.7 GB), built once."""
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    m = VerseCrafterWanTransformer3DModel(geoada_in_dim=128, param_device=dev, param_dtype=torch.bfloat16,
                                          dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, skip_init=True)
    m.init_weights(zero_init_outputs=False)
    yield m
    del m
    torch.cuda.empty_cache()


def test_config5_two_experts_fp8_at_81_frames_720p(wan14b, wan14b_high):
    """BASELINE config 5 at ITS workload on one GPU: the Wan2.2-style pair (two resident 14B + GeoAdapter experts switched at `boundary`),
    fp8 linear layers AND fp8 self-attention, latent [16, 21, 90, 160] = 81 frames 1280 x 720, L = 75 600, the CFG pair batched.  The
    reference has no code for this configuration (config/wan2.2/wan_civitai_t2v.yaml:4-7 is read by nothing) and sizes like this are out
    of the CPU oracle's reach: size-independent properties -- both experts run (one sampler step on each side of the boundary), finite,
    the two CFG halves are computed independently (swapping the prompts swaps the outputs bit for bit, which is also determinism), and the
    fp8 forward stays within the drift bound test_fp8_linear_mode / test_fp8_attention_mode state against the bf16 forward."""
    from versecrafter_amd.pipeline import WanVerseCrafterPipeline
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    low, high, dev = wan14b, wan14b_high, torch.device("cuda", 0)
    T, h, w = 21, 90, 160
    g = torch.Generator().manual_seed(55)
    x = torch.randn(1, 16, T, h, w, generator=g).to(dev, torch.bfloat16)
    geo = torch.randn(1, 128, T, h, w, generator=g).to(dev, torch.bfloat16)
    pe, ne = torch.randn(77, 4096, generator=g).to(dev, torch.bfloat16), torch.randn(60, 4096, generator=g).to(dev, torch.bfloat16)
    L = T * (h // 2) * (w // 2)
    assert L == 75600
    x2, geo2, t = torch.cat([x, x]), torch.cat([geo, geo]), torch.tensor([900.0, 900.0], device=dev)
    ref = low(x2, t, geo2, [ne, pe], L).clone()                                   # bf16
    try:
        for m in (low, high):
            m.enable_fp8_linear()
            m.enable_fp8_attention(True, 1)
        a = low(x2, t, geo2, [ne, pe], L).clone()
        b = low(x2, t, geo2, [pe, ne], L)
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a[0], b[1]) and torch.equal(a[1], b[0])
        e = rel(a, ref)
        print(f"config 5 clip, fp8 linear + fp8 self-attention vs bf16 on the low-noise expert: rel L2 {e:.4g}")
        assert 1e-3 < e < 0.2, e
        # the pair through the sampler: two steps, one on each side of the boundary (shift 1: t = 1000, 500)
        pipe = WanVerseCrafterPipeline(transformer=low, transformer_2=high, scheduler=FlowUniPCMultistepScheduler(shift=1))
        out = pipe(prompt_embeds=[pe], negative_prompt_embeds=[ne], height=h * 8, width=w * 8, geoada_latents=[geo[0, :64]],
                   mask_latents=[geo[0, 64:]], num_inference_steps=2, guidance_scale=5.0, shift=1, latents=x.clone(), output_type="latent",
                   boundary=0.875).videos
        torch.cuda.synchronize()
        assert pipe._high_noise_steps == [True, False]
        assert out.shape == (1, 16, T, h, w) and torch.isfinite(out.float()).all()
        free, total = torch.cuda.mem_get_info()
        print(f"config 5 at 81 x 720 x 1280, two resident experts, fp8: {(total - free) / 2**30:.1f} GiB of {total / 2**30:.0f} GiB in use")
    finally:
        for m in (low, high):
            m.enable_fp8_linear(False)
            m.enable_fp8_attention(False)
    assert torch.equal(low(x2, t, geo2, [ne, pe], L), ref)                         # the modes switch off cleanly


def test_two_resident_14b_experts_switch_at_the_boundary(wan14b, wan14b_high):
    """BASELINE config 5's model pair at full size: two Wan-14B + GeoAdapter experts (2 x 43.7 GB of weights) resident on ONE MI355X,
    a four-step sampler that crosses the boundary; the result equals the single-expert pipelines swapped by hand at the switch."""
    from versecrafter_amd.pipeline import WanVerseCrafterPipeline
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    dev = torch.device("cuda", 0)
    high = wan14b_high
    g = torch.Generator().manual_seed(9)
    T, h, w = 2, 16, 24
    lat0 = torch.randn(1, 16, T, h, w, generator=g).to(dev, torch.bfloat16)
    geo = torch.randn(64, T, h, w, generator=g).to(dev, torch.bfloat16)
    msk = (torch.rand(64, T, h, w, generator=g) < 0.5).to(dev, torch.bfloat16)
    pe, ne = torch.randn(77, 4096, generator=g).to(dev, torch.bfloat16), torch.randn(60, 4096, generator=g).to(dev, torch.bfloat16)

    def run(t1, t2=None, callback=None):
        pipe = WanVerseCrafterPipeline(transformer=t1, transformer_2=t2, scheduler=FlowUniPCMultistepScheduler(shift=1))
        out = pipe(prompt_embeds=[pe], negative_prompt_embeds=[ne], height=h * 8, width=w * 8, geoada_latents=[geo], mask_latents=[msk],
                   num_inference_steps=4, guidance_scale=5.0, shift=12, latents=lat0.clone(), output_type="latent", boundary=0.875,
                   callback_on_step_end=callback).videos
        torch.cuda.synchronize()
        return out, pipe
    mixed, pipe = run(wan14b, high)
    hi = pipe._high_noise_steps
    assert 0 < sum(hi) < 4 and torch.isfinite(mixed.float()).all()

    def swap(p, i, t, kw):
        p.transformer = high if (i + 1 < 4 and hi[i + 1]) else wan14b
        return kw
    want, _ = run(high if hi[0] else wan14b, callback=swap)
    assert torch.equal(mixed, want)
    free, total = torch.cuda.mem_get_info()
    print(f"two resident 14B experts: {(total - free) / 2**30:.1f} GiB of {total / 2**30:.0f} GiB in use")
    assert total - free > 80 * 2**30                          # both sets of weights are on the device


@pytest.mark.parametrize("name,T,h,w", [
    ("cfg2: 49 frames 480x832, L=20280", 13, 60, 104),
    ("cfg3: 81 frames 480x832, L=32760 (the bench workload)", 21, 60, 104),
    ("cfg4: 81 frames 720x1280, L=75600, the whole sequence on one GPU", 21, 90, 160),
])
def test_full_14b_model_properties(wan14b, name, T, h, w):
    """BASELINE.json configs 2, 3 and 4 on the full model (B=2): finite, deterministic, and the two CFG halves are computed
    independently (swapping them swaps the outputs bit for bit).  Sizes the CPU oracle cannot reach: size-independent
    properties instead."""
    m, dev = wan14b, torch.device("cuda", 0)
    g = torch.Generator().manual_seed(2025)
    x = torch.randn(2, 16, T, h, w, generator=g).to(dev, torch.bfloat16)
    geo = torch.randn(2, 128, T, h, w, generator=g).to(dev, torch.bfloat16)
    ctx = [torch.randn(60, 4096, generator=g).to(dev, torch.bfloat16), torch.randn(77, 4096, generator=g).to(dev, torch.bfloat16)]
    t = torch.tensor([900.0, 900.0], device=dev)
    L = T * (h // 2) * (w // 2)
    a = m(x, t, geo, ctx, L)
    b = m(x, t, geo, ctx, L)
    c = m(x.flip(0), t, geo.flip(0), ctx[::-1], L)
    torch.cuda.synchronize()
    assert a.shape == (2, 16, T, h, w) and torch.isfinite(a.float()).all() and a.float().abs().max() > 0
    assert torch.equal(a, b)
    assert torch.equal(c, a.flip(0))


def test_from_pretrained_to_cuda_forward_equals_load_state_dict_bitwise(fwd, tmp_path):
    """a18: checkpoint directory (config.json + sharded safetensors, WT.py:1176-1322) -> from_pretrained -> .to(cuda) ->
    forward must equal constructing the module and load_state_dict-ing the same tensors."""
    import json
    from safetensors.torch import save_file
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    W = O.random_weights(O.Config(**TINY), 7)
    json.dump(dict(TINY), open(tmp_path / "config.json", "w"))
    keys = sorted(W)
    save_file({k: W[k].bfloat16() for k in keys[::2]}, str(tmp_path / "a-00001-of-00002.safetensors"))
    save_file({k: W[k].bfloat16() for k in keys[1::2]}, str(tmp_path / "a-00002-of-00002.safetensors"))
    m = VerseCrafterWanTransformer3DModel.from_pretrained(str(tmp_path), low_cpu_mem_usage=True, torch_dtype=torch.bfloat16)
    m = m.to("cuda")
    L = int(fwd["A.seq_len"])
    assert torch.equal(run(m, fwd, L), run(_fresh_model(), fwd, L))


@pytest.mark.parametrize("n_steps,shift,do_cfg", [(8, 16.0, True), (5, 5.0, True), (6, 16.0, False), (2, 16.0, True)])
def test_fused_sampler_update_equals_torch_formulation_bitwise(n_steps, shift, do_cfg):
    """PIPE.py:903-909 -- CFG combine + scheduler.step -- as one HIP kernel (FlowUniPCMultistepScheduler.step_cfg ->
    vc_op_unipc_update) against the op-by-op torch formulation on the same device tensors: every step of a run must agree
    bit for bit (first step without corrector, order-1 and order-2 corrector / predictor, lower-order final step)."""
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    g = torch.Generator().manual_seed(n_steps * 100 + int(shift))
    shape = (1, 16, 3, 8, 12)
    x = torch.randn(shape, generator=g).bfloat16().cuda()
    a = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    b = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    a.set_timesteps(n_steps, device="cuda", shift=shift)
    b.set_timesteps(n_steps, device="cuda", shift=shift)
    xa, xb = x.clone(), x.clone()
    guidance = 5.0
    for i, t in enumerate(a.timesteps):
        npred = torch.randn((2 if do_cfg else 1,) + shape[1:], generator=g).bfloat16().cuda()
        if do_cfg:
            u, c = npred.chunk(2)
            noise = u + guidance * (c - u)
        else:
            noise = npred
        xa = a.step(noise, t, xa, return_dict=False)[0]
        xb = b.step_cfg(npred, t, xb, guidance if do_cfg else None)
        torch.cuda.synchronize()
        assert xb.dtype == torch.bfloat16 and torch.equal(xa, xb), f"step {i}: max diff {(xa.float() - xb.float()).abs().max()}"
        assert torch.equal(a.model_outputs[-1], b.model_outputs[-1]) and torch.equal(a.last_sample, b.last_sample)
    assert torch.isfinite(xb.float()).all()


@pytest.mark.parametrize("n_steps,shift,do_cfg", [(8, 16.0, True), (5, 5.0, True), (6, 16.0, False), (2, 16.0, True), (50, 16.0, True)])
def test_fused_sampler_update_vs_unipc_oracle_per_step(n_steps, shift, do_cfg):
    """FlowUniPCMultistepScheduler.step_cfg (ONE HIP kernel: CFG combine, x0, UniPC corrector, predictor) against
    oracle/unipc_oracle.py (float64, closed forms; third-party algorithm: parity unpinned) -- every step type of a run: first
    step, order-1 and order-2 corrector / predictor, lower-order final step.

    Per step the oracle is loaded with the kernel's own bf16 state (sample, last sample, the two previous x0) so that one
    update is compared, not accumulated drift.  Tolerance: the kernel rounds to bf16 after each of the <= 14 elementwise ops of
    the torch formulation it reproduces; with |coefficients| <= ~2 the worst case is a handful of half-ulps of the largest
    term: |hip - oracle| <= 8 * 2^-8 * max(|sample|, |x0|, |noise|) elementwise-max norm.  Measured: <= 2.5 of those ulps.
    The free-running trajectories (no re-synchronisation) must stay within 3e-2 relative L2 after the whole run."""
    import numpy as np
    from oracle.unipc_oracle import UniPCOracle, cfg_combine
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    g = torch.Generator().manual_seed(n_steps * 100 + int(shift))
    shape = (1, 16, 3, 8, 12)
    xb = torch.randn(shape, generator=g).bfloat16().cuda()
    b = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    b.set_timesteps(n_steps, device="cuda", shift=shift)
    orc, free = UniPCOracle(n_steps, shift), UniPCOracle(n_steps, shift)
    assert np.array_equal(orc.timesteps, b.timesteps.cpu().numpy())
    f64 = lambda t: t.detach().double().cpu().numpy()
    x_free = f64(xb)
    guidance, worst = 5.0, 0.0
    for i, t in enumerate(b.timesteps):
        npred = torch.randn((2 if do_cfg else 1,) + shape[1:], generator=g).bfloat16().cuda()
        v = cfg_combine(f64(npred[:1]), f64(npred[1:]), guidance) if do_cfg else f64(npred)
        # oracle state := the kernel's state before this step
        orc.i = i
        orc.m = [f64(m) for m in b.model_outputs if m is not None][-2:]
        orc.last_sample = None if b.last_sample is None else f64(b.last_sample)
        orc.order_used, orc.lower = b.this_order if i > 0 else 1, b.lower_order_nums
        want = orc.step(v, f64(xb))
        x_free = free.step(v, x_free)
        scale = max(float(xb.float().abs().max()), float(np.abs(v).max()), float(np.abs(orc.m[-1]).max()))
        xb = b.step_cfg(npred, t, xb, guidance if do_cfg else None)
        torch.cuda.synchronize()
        err = float(np.abs(f64(xb) - want).max()) / (2.0 ** -8 * scale)
        worst = max(worst, err)
        assert err <= 8.0, f"step {i}: {err:.2f} bf16 ulps of {scale:.3g}"
    rel_free = float(np.linalg.norm(f64(xb) - x_free) / np.linalg.norm(x_free))
    print(f"unipc n={n_steps} shift={shift}: worst per-step error {worst:.2f} ulp_bf16, free-running rel L2 {rel_free:.3g}")
    assert rel_free < 3e-2


def test_time_embedding_vs_oracle(model):
    """a11 in isolation: sinusoidal_embedding_1d (WT.py:39-49) -> time_embedding -> time_projection (VC.py:347-350), fp32 on
    the engine (vc_time_embedding) against the oracle on the same bf16-rounded weights; timesteps across the schedule."""
    cfg = O.Config(**TINY)
    W = {k: v.bfloat16().float() for k, v in O.random_weights(cfg, 7).items()}
    t = torch.tensor([999.0, 875.0, 500.0, 31.0, 1.0, 0.0])
    got = model.time_embedding_e0(t.cuda()).cpu()
    e, e0 = O.time_embed(W, t, cfg.dim, cfg.freq_dim)
    assert got.shape == e0.shape == (6, 6, cfg.dim)
    err = (got - e0).abs().max().item()
    print(f"time embedding: max abs err {err:.3g} (|e0| max {e0.abs().max():.3g})")
    assert torch.allclose(got, e0, rtol=1e-4, atol=1e-4 * float(e0.abs().max()))       # fp32 dot products, different summation order


def test_graph_replay_equals_eager_bitwise(fwd, monkeypatch):
    """Launch-bound sizes replay a captured hipGraph from the second forward with a (flags, scale) key on: inputs staged in,
    graph launched on the caller's stream, output staged out.  Every replay must equal the eager engine (VC_GRAPH=0) bit for
    bit -- changing latents and timesteps, a second scale (its own graph), the TeaCache flag pairs, and a non-default stream."""
    L = int(fwd["A.seq_len"])
    xs = [fwd["A.x"], fwd["C.x2"], fwd["A.x"] * 0.5, fwd["C.x2"] * 0.75, fwd["A.x"] * -0.3]
    ts = [fwd["C.t1"], fwd["C.t2"], fwd["C.t2"] - 40.0, fwd["C.t2"] - 80.0, fwd["C.t2"] - 120.0]
    monkeypatch.setenv("VC_GRAPH", "0")
    eager = _fresh_model()
    monkeypatch.setenv("VC_GRAPH", "1")
    graph = _fresh_model()
    for i in range(5):
        for scale in (1.0, 0.6):
            assert torch.equal(run(graph, fwd, L, x=xs[i], t=ts[i], scale=scale), run(eager, fwd, L, x=xs[i], t=ts[i], scale=scale)), (i, scale)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        a = run(graph, fwd, L, x=xs[1], t=ts[1])
    side.synchronize()
    assert torch.equal(a, run(eager, fwd, L, x=xs[1], t=ts[1]))
    for m in (eager, graph):
        m.enable_teacache([1.0, 0.0], num_steps=6, rel_l1_thresh=1e9, num_skip_start_steps=2, offload=False)
    for i in range(5):                               # calc, calc, skip, skip, skip: STORE graphs, then USE graphs
        assert torch.equal(run(graph, fwd, L, x=xs[i], t=ts[i]), run(eager, fwd, L, x=xs[i], t=ts[i])), i
        assert graph.should_calc == eager.should_calc == (i < 2)


def test_two_expert_sampler_switches_at_the_boundary(model):
    """BASELINE config 5's "MoE" as upstream Wan2.2 defines it (two full DiTs switched by the timestep; the reference ships only the
    configs): steps with t >= boundary * 1000 run transformer_2.  Checked against single-model runs: boundary above every timestep =
    the low-noise model alone, boundary 0 = the high-noise model alone, a boundary in between = a single-model pipeline whose model is
    swapped by a step callback at the switch -- all bit-equal; both experts carry their own TeaCache state."""
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    from versecrafter_amd.pipeline import WanVerseCrafterPipeline
    from versecrafter_amd.pipeline.pipeline_wan_versecrafter import expert_schedule
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    cfg = O.Config(**TINY)
    low = model
    high = VerseCrafterWanTransformer3DModel(**TINY)
    high.load_state_dict(O.random_weights(cfg, 23))
    high = high.to(torch.bfloat16).to("cuda")
    g = torch.Generator().manual_seed(5)
    T, h, w = 3, 8, 12
    lat0 = torch.randn(1, 16, T, h, w, generator=g).bfloat16().cuda()
    geo = torch.randn(64, T, h, w, generator=g).bfloat16().cuda()
    msk = (torch.rand(64, T, h, w, generator=g) < 0.5).to(torch.bfloat16).cuda()
    pe, ne = torch.randn(33, cfg.text_dim, generator=g).bfloat16().cuda(), torch.randn(20, cfg.text_dim, generator=g).bfloat16().cuda()
    steps = 6

    def run(t1, t2=None, boundary=0.875, callback=None):
        pipe = WanVerseCrafterPipeline(transformer=t1, transformer_2=t2, scheduler=FlowUniPCMultistepScheduler(shift=1))
        out = pipe(prompt_embeds=[pe], negative_prompt_embeds=[ne], height=h * 8, width=w * 8, geoada_latents=[geo], mask_latents=[msk],
                   num_inference_steps=steps, guidance_scale=5.0, shift=12, latents=lat0.clone(), output_type="latent", boundary=boundary,
                   callback_on_step_end=callback).videos
        torch.cuda.synchronize()
        return out, pipe
    only_low, _ = run(low)
    only_high, _ = run(high)
    assert not torch.equal(only_low, only_high)
    a, _ = run(low, high, boundary=1.1)
    assert torch.equal(a, only_low)
    b, _ = run(low, high, boundary=0.0)
    assert torch.equal(b, only_high)
    mixed, pipe = run(low, high, boundary=0.875)
    hi = pipe._high_noise_steps
    assert hi == expert_schedule(pipe.scheduler.timesteps, 0.875) and 0 < sum(hi) < steps          # shift 12: the first steps are above 875

    def swap(p, i, t, kw):
        p.transformer = high if (i + 1 < steps and hi[i + 1]) else low
        return kw
    want, _ = run(high if hi[0] else low, callback=swap)
    assert torch.equal(mixed, want)
    assert torch.isfinite(mixed.float()).all() and not torch.equal(mixed, only_low) and not torch.equal(mixed, only_high)


def test_two_expert_pipeline_with_teacache_is_reusable(model):
    """Round-3 advisor finding: with two experts each TeaCache only counts the steps ITS expert ran, so the reference's own reset
    (cnt == num_steps at the end of forward, VC.py:438-441) never fires; a second video through the same pipeline then started past
    num_skip_start_steps, could skip at step 0 and re-add the PREVIOUS video's residual.  The sampler now resets every expert's gate and
    the engines' residual slots at the start of a call: call 2 must equal what a fresh set of gates gives."""
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    from versecrafter_amd.pipeline import WanVerseCrafterPipeline
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    cfg = O.Config(**TINY)
    low = model
    high = VerseCrafterWanTransformer3DModel(**TINY)
    high.load_state_dict(O.random_weights(cfg, 23))
    high = high.to(torch.bfloat16).to("cuda")
    g = torch.Generator().manual_seed(8)
    T, h, w = 3, 8, 12
    lats = [torch.randn(1, 16, T, h, w, generator=g).bfloat16().cuda() for _ in range(2)]
    geo = torch.randn(64, T, h, w, generator=g).bfloat16().cuda()
    msk = (torch.rand(64, T, h, w, generator=g) < 0.5).to(torch.bfloat16).cuda()
    pe, ne = torch.randn(33, cfg.text_dim, generator=g).bfloat16().cuda(), torch.randn(20, cfg.text_dim, generator=g).bfloat16().cuda()
    steps = 6
    coeff = [0.0, 0.0, 0.0, 1.0, 0.0]                      # accumulated distance = sum of relative L1 changes: far below the threshold

    def gates():
        for m in (low, high):
            m.enable_teacache(coeff, steps, 1e6, num_skip_start_steps=1, offload=False)     # skips every step it may skip

    def call(pipe, lat):
        out = pipe(prompt_embeds=[pe], negative_prompt_embeds=[ne], height=h * 8, width=w * 8, geoada_latents=[geo], mask_latents=[msk],
                   num_inference_steps=steps, guidance_scale=5.0, shift=12, latents=lat.clone(), output_type="latent", boundary=0.875).videos
        torch.cuda.synchronize()
        return out
    try:
        gates()
        pipe = WanVerseCrafterPipeline(transformer=low, transformer_2=high, scheduler=FlowUniPCMultistepScheduler(shift=1))
        first = call(pipe, lats[0])
        second = call(pipe, lats[1])                       # same pipeline, same gates, another video
        assert 0 < sum(pipe._high_noise_steps) < steps
        gates()                                            # fresh gates (and the pipeline drops the engines' residuals itself)
        fresh = call(WanVerseCrafterPipeline(transformer=low, transformer_2=high, scheduler=FlowUniPCMultistepScheduler(shift=1)), lats[1])
        assert torch.isfinite(second.float()).all() and torch.equal(second, fresh)
        assert not torch.equal(first, second)
        # the gates really skipped: the low-noise expert's counter shows more forwards than computed steps would leave it at
        assert low.teacache.cnt > 0 and high.teacache.cnt > 0
    finally:
        low.disable_teacache()
        high.disable_teacache()


def test_fp8_attention_mode(model):
    """enable_fp8_attention (this build; BASELINE config 5's dtype): the blocks' SELF-attention through csrc/attention_fp8.hip.  The output
    moves away from the bf16 forward by the quantisation error (tiny model: < 8e-2 rel L2, both ways of making the weights' bytes), is
    deterministic, a graph replay equals the eager call, and switching the mode off restores the bf16 forward bit for bit."""
    g = torch.Generator().manual_seed(22)
    T, h, w = 3, 8, 12
    x = torch.randn(2, 16, T, h, w, generator=g).bfloat16().cuda()
    geo = torch.randn(2, 128, T, h, w, generator=g).bfloat16().cuda()
    ctx = [torch.randn(20, 64, generator=g).bfloat16().cuda(), torch.randn(33, 64, generator=g).bfloat16().cuda()]
    t = torch.tensor([640.0, 640.0]).cuda()
    ref = model(x, t, geo, ctx, 72).clone()
    try:
        outs = {}
        for pmode in (1, 0):
            model.enable_fp8_attention(True, pmode)
            a = model(x, t, geo, ctx, 72).clone()
            b = model(x, t, geo, ctx, 72).clone()
            c = model(x, t, geo, ctx, 72).clone()
            assert torch.equal(a, b) and torch.equal(a, c) and torch.isfinite(a.float()).all()
            e = rel(a, ref)
            print(f"tiny model, fp8 self-attention pmode {pmode} vs bf16: rel L2 {e:.4g}")
            assert 1e-4 < e < 8e-2, e
            outs[pmode] = a
        assert not torch.equal(outs[0], outs[1])
        model.enable_fp8_linear()                          # both fp8 modes together
        d = model(x, t, geo, ctx, 72)
        assert torch.isfinite(d.float()).all() and rel(d, ref) < 0.12
    finally:
        model.enable_fp8_linear(False)
        model.enable_fp8_attention(False)
    assert torch.equal(model(x, t, geo, ctx, 72), ref)


def test_fp8_linear_mode(model):
    """enable_fp8_linear (this build; BASELINE config 5's dtype): the blocks' Linear layers on e4m3 operands.  The output moves away
    from the bf16 forward by the quantisation error and no further (tiny model: < 8e-2 rel L2), is deterministic, survives re-loading
    the weights, and switching the mode off restores the bf16 forward bit for bit."""
    g = torch.Generator().manual_seed(21)
    T, h, w = 3, 8, 12
    x = torch.randn(2, 16, T, h, w, generator=g).bfloat16().cuda()
    geo = torch.randn(2, 128, T, h, w, generator=g).bfloat16().cuda()
    ctx = [torch.randn(20, 64, generator=g).bfloat16().cuda(), torch.randn(33, 64, generator=g).bfloat16().cuda()]
    t = torch.tensor([640.0, 640.0]).cuda()
    ref = model(x, t, geo, ctx, 72).clone()
    model.enable_fp8_linear()
    try:
        a = model(x, t, geo, ctx, 72).clone()
        b = model(x, t, geo, ctx, 72).clone()
        assert torch.equal(a, b) and torch.isfinite(a.float()).all()
        e = rel(a, ref)
        assert 1e-4 < e < 8e-2, e
        model.mark_weights_changed()                               # parameters re-registered: the e4m3 copies are rebuilt
        for p_ in model.parameters():
            p_._version                                            # (no change of values)
        c = model(x, t, geo, ctx, 72)
        assert torch.equal(c, a)
    finally:
        model.enable_fp8_linear(False)
    assert torch.equal(model(x, t, geo, ctx, 72), ref)


@pytest.mark.parametrize("width", ["tiny", "14b"])
def test_fp8_layernorm_fusion_is_bit_equal_to_the_separate_quantiser(width, monkeypatch):
    """In fp8 mode the three LayerNorms of a block that feed only GEMMs (norm1 -> q/k/v, norm3 -> cross q, norm2 -> ffn.0) write the
    GEMM's e4m3 operand themselves (no bf16 row, no quantiser pass).  Same bytes, same scales: the forward is bit-equal to the
    unfused form (VC_FP8_FUSE_LN=0)."""
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    dev = torch.device("cuda", 0)
    dims = dict(TINY) if width == "tiny" else dict(geoada_in_dim=128, dim=5120, ffn_dim=13824, num_heads=40, num_layers=2)
    T, h, w = (3, 8, 12) if width == "tiny" else (2, 16, 48)
    td = dims.get("text_dim", 4096)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 16, T, h, w, generator=g).to(dev, torch.bfloat16)
    geo = torch.randn(2, 128, T, h, w, generator=g).to(dev, torch.bfloat16)
    ctx = [torch.randn(20, td, generator=g).to(dev, torch.bfloat16), torch.randn(33, td, generator=g).to(dev, torch.bfloat16)]
    t = torch.tensor([640.0, 640.0], device=dev)
    L = T * (h // 2) * (w // 2)
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("VC_FP8_FUSE_LN", fuse)                   # read at vc_create
        torch.manual_seed(0)
        m = VerseCrafterWanTransformer3DModel(param_device=dev, param_dtype=torch.bfloat16, skip_init=True, **dims)
        m.init_weights(zero_init_outputs=False)
        m.enable_fp8_linear()
        outs.append(m(x, t, geo, ctx, L).clone())
        outs.append(m(x, t, geo, ctx, L).clone())                    # second call: captured and replayed (M <= 16384 rows)
        # the capture really happened: in fp8 mode the capture stream's scratch exists before the capture begins (round-3 advisor finding:
        # it used to be allocated INSIDE the capture, the capture was dropped silently and every forward stayed eager)
        assert m.graph_replays() >= 1
        del m
        torch.cuda.empty_cache()
    assert torch.isfinite(outs[0].float()).all() and float(outs[0].float().std()) > 1e-3
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[2], outs[3])


def test_fp8_linear_at_the_14b_width_against_bf16():
    """The same at the production width (d = 5120, 40 heads, ffn 13824; 2 + 1 blocks, 768 tokens): every fp8 GEMM shape of the real
    model, against the bf16 engine."""
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = VerseCrafterWanTransformer3DModel(param_device=dev, param_dtype=torch.bfloat16, skip_init=True, geoada_in_dim=128, dim=5120,
                                          ffn_dim=13824, num_heads=40, num_layers=2)
    m.init_weights(zero_init_outputs=False)
    g = torch.Generator().manual_seed(2)
    T, h, w = 2, 16, 48
    x = torch.randn(2, 16, T, h, w, generator=g).to(dev, torch.bfloat16)
    geo = torch.randn(2, 128, T, h, w, generator=g).to(dev, torch.bfloat16)
    ctx = [torch.randn(60, 4096, generator=g).to(dev, torch.bfloat16), torch.randn(77, 4096, generator=g).to(dev, torch.bfloat16)]
    t = torch.tensor([700.0, 700.0], device=dev)
    L = T * (h // 2) * (w // 2)
    ref = m(x, t, geo, ctx, L).clone()
    m.enable_fp8_linear()
    got = m(x, t, geo, ctx, L).clone()
    e = rel(got, ref)
    print(f"14B width, fp8 linear layers vs bf16: rel L2 {e:.4g}")
    assert torch.isfinite(got.float()).all() and 1e-4 < e < 8e-2, e
    del m
    torch.cuda.empty_cache()


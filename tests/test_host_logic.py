"""CPU tests of the host side: C-ABI symbol table, state-dict contract, from_pretrained rules, the sampler loop
semantics (with a test-double transformer; the arithmetic checker is the oracle), pipeline front-end helpers."""
import ctypes
import json
import math
import os

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

from oracle import unipc_oracle as U
from oracle import wan_oracle as O
from versecrafter_amd import _lib
from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
from versecrafter_amd.pipeline import WanVerseCrafterPipeline
from versecrafter_amd.pipeline.pipeline_wan_versecrafter import geoada_encode_masks
from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler

TINY = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=4, text_dim=64, text_len=48,
            geoada_in_dim=128, in_dim=16, out_dim=16, freq_dim=256)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """include/vcengine.h <-> _lib.SYMBOLS <-> libvcengine.so (load only: no compute without a GPU)."""
    assert os.path.isfile(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "vcengine.h")).read()
    for name in _lib.SYMBOLS:
        assert hasattr(lib, name), name
        assert name + "(" in header, f"{name} not declared in include/vcengine.h"
    import re
    declared = set(re.findall(r"\b(vc_[a-z0-9_]+)\s*\(", header)) - {"vc_all_to_all_fn", "vc_all_gather_fn"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib.vc_abi_version.restype = ctypes.c_int
    assert lib.vc_abi_version() == _lib.VC_ABI_VERSION


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("only meaningful without a GPU")
    lib = _lib.load()
    cfg = _lib.vc_config(dim=256, ffn_dim=512, num_heads=2, num_layers=4, in_dim=16, out_dim=16, geoada_in_dim=128,
                         text_dim=64, text_len=48, freq_dim=256, eps=1e-6, num_geoada_layers=0)
    h = ctypes.c_void_p()
    rc = lib.vc_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == _lib.VC_E_HIP and b"no HIP device" in lib.vc_last_error(None)
    m = VerseCrafterWanTransformer3DModel(**TINY)
    x = torch.zeros(2, 16, 3, 8, 12, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        m(x, torch.zeros(2), torch.zeros(2, 128, 3, 8, 12, dtype=torch.bfloat16), [torch.zeros(4, 64)] * 2, 72)


def test_state_dict_contract_matches_reference_keys():
    m = VerseCrafterWanTransformer3DModel(**TINY)
    sd = m.state_dict()
    want = O.state_dict_shapes(O.Config(**TINY))
    assert set(sd) == set(want)
    assert all(tuple(sd[k].shape) == want[k] for k in want)
    assert len(sd) == 185                                   # same count as the reference class (tiny config)
    assert m.blocks[1].self_attn.q.weight.shape == (256, 256)
    assert m.geoada_blocks[0].before_proj.weight.abs().max() == 0      # zero-init (VC.py:106-110)
    assert m.head.head.weight.abs().max() == 0                         # WT.py:1174
    assert m.config.patch_size == (1, 2, 2) and m.freqs.shape == (1024, 64) and m.freqs.dtype == torch.complex128
    with pytest.raises(AssertionError):
        VerseCrafterWanTransformer3DModel(geoada_layers=[2], **TINY)   # assert 0 in geoada_layers (VC.py:178)


def test_from_pretrained_rules(tmp_path):
    cfg = O.Config(**TINY)
    W = O.random_weights(cfg, 5)
    ck = {k: v.clone() for k, v in W.items()}
    ck["patch_embedding.weight"] = W["patch_embedding.weight"][:, :12].contiguous()        # narrower: zero-padded
    ck["geoada_patch_embedding.weight"] = torch.randn(256, 16, 1, 2, 2)                    # other geoada_in_dim: skipped
    ck["blocks.0.ffn.0.bias"] = torch.zeros(7)                                             # mismatched: skipped
    ck["not_a_key"] = torch.zeros(1)
    save_file(ck, str(tmp_path / "diffusion_pytorch_model.safetensors"))
    conf = dict(TINY, geoada_in_dim=16, hidden_size=256, in_channels=16)
    conf.pop("in_dim"), conf.pop("dim")
    conf.update(in_dim=16, dim=256)
    json.dump(conf, open(tmp_path / "config.json", "w"))
    m = VerseCrafterWanTransformer3DModel.from_pretrained(
        str(tmp_path), transformer_additional_kwargs={"geoada_in_dim": 128,
                                                      "dict_mapping": {"in_dim": "in_channels", "dim": "hidden_size"}},
        low_cpu_mem_usage=True, torch_dtype=torch.bfloat16)
    sd = m.state_dict()
    assert sd["blocks.1.self_attn.q.weight"].dtype == torch.bfloat16
    assert torch.equal(sd["blocks.1.self_attn.q.weight"], W["blocks.1.self_attn.q.weight"].bfloat16())
    pw = sd["patch_embedding.weight"]
    assert torch.equal(pw[:, :12], W["patch_embedding.weight"][:, :12].bfloat16()) and pw[:, 12:].abs().max() == 0
    g = sd["geoada_patch_embedding.weight"]
    assert g.shape == (256, 128, 1, 2, 2) and g.abs().max() > 0 and sd["geoada_patch_embedding.bias"].abs().max() == 0
    with pytest.raises(RuntimeError):
        VerseCrafterWanTransformer3DModel.from_pretrained(str(tmp_path / "nope"))


def test_from_pretrained_bin_shards_and_missing_keys(tmp_path):
    """WT.py:1220, 1281: a pickle checkpoint (read with torch.load(weights_only=True): nothing executed) and the sharded
    *.safetensors glob load the same tensors; keys the checkpoint lacks are initialised (never left as raw memory)."""
    cfg = O.Config(**TINY)
    W = O.random_weights(cfg, 5)
    json.dump(dict(TINY), open(tmp_path / "config.json", "w"))
    drop = {"blocks.3.ffn.2.bias", "geoada_blocks.1.after_proj.weight", "head.head.weight"}
    torch.save({k: v for k, v in W.items() if k not in drop}, str(tmp_path / "diffusion_pytorch_model.bin"))
    a = VerseCrafterWanTransformer3DModel.from_pretrained(str(tmp_path), torch_dtype=torch.bfloat16)
    sa = a.state_dict()
    for k, v in W.items():
        if k not in drop:
            assert torch.equal(sa[k], v.bfloat16()), k
    assert sa["blocks.3.ffn.2.bias"].abs().max() == 0                       # bias family: zeros
    assert sa["head.head.weight"].abs().max() == 0                          # zero-initialised output (WT.py:1174)
    assert sa["geoada_blocks.1.after_proj.weight"].abs().max() == 0         # VC.py:106-110
    assert all(torch.isfinite(v.float()).all() for v in sa.values())

    shards = tmp_path / "sharded"
    shards.mkdir()
    json.dump(dict(TINY), open(shards / "config.json", "w"))
    keys = sorted(W)
    save_file({k: W[k] for k in keys[::2]}, str(shards / "model-00001-of-00002.safetensors"))
    save_file({k: W[k] for k in keys[1::2]}, str(shards / "model-00002-of-00002.safetensors"))
    b = VerseCrafterWanTransformer3DModel.from_pretrained(str(shards), torch_dtype=torch.bfloat16)
    for k, v in b.state_dict().items():
        assert torch.equal(v, W[k].bfloat16()), k


def test_geoada_encode_masks_matches_oracle():
    g = torch.Generator().manual_seed(0)
    mask = (torch.rand(3, 9, 64, 96, generator=g) < 0.5).float()
    got = geoada_encode_masks([mask])[0]
    want = O.geoada_encode_masks(mask)
    assert got.shape == (64, 3, 8, 12) and torch.equal(got, want)


class FakeTransformer(torch.nn.Module):
    """Test double: records what the sampler passes and returns a known function of its inputs."""

    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.config = type("C", (), {"patch_size": (1, 2, 2)})()
        self.calls = []

    def forward(self, x, context, t, geoada_context, seq_len, geoada_context_scale):
        self.calls.append(dict(x=x.clone(), t=t.clone(), ctx_lens=[len(c) for c in context], geo=geoada_context.shape,
                               seq_len=seq_len, step=self.current_steps))
        out = 0.1 * x + 0.01 * t.view(-1, 1, 1, 1, 1) / 1000
        out[x.shape[0] // 2:] += 0.05                      # cond half differs from uncond half
        return out


def test_sampler_loop_semantics_against_oracle():
    """PIPE.py:871-925: CFG order [uncond, cond], t.expand(B), u + g (c - u), scheduler.step; seq_len PIPE.py:861-865."""
    tr = FakeTransformer()
    sch = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    pipe = WanVerseCrafterPipeline(transformer=tr, scheduler=sch)
    g = torch.Generator().manual_seed(2025)
    T, h, w = 3, 8, 12
    geo = [torch.randn(64, T, h, w)]
    mask_video = (torch.rand(1, 1, 9, 64, 96, generator=g) < 0.5).float()
    lat0 = torch.randn(1, 16, T, h, w, generator=g)
    n, guidance = 5, 5.0
    out = pipe(prompt_embeds=[torch.randn(7, 64)], negative_prompt_embeds=[torch.randn(5, 64)], height=64, width=96,
               geoada_latents=geo, mask_video=mask_video, num_inference_steps=n, guidance_scale=guidance, shift=16,
               latents=lat0.clone(), output_type="latent").videos
    assert len(tr.calls) == n
    c0 = tr.calls[0]
    assert c0["ctx_lens"] == [5, 7]                          # negative first (PIPE.py:741)
    assert c0["geo"] == (2, 128, T, h, w) and c0["seq_len"] == math.ceil(h * w / 4 * T)
    assert [c["step"] for c in tr.calls] == list(range(n))
    assert torch.equal(c0["x"][0], c0["x"][1]) and c0["t"].shape == (2,) and c0["t"][0] == c0["t"][1]
    # replay with the oracle scheduler + oracle CFG combine
    orc = U.UniPCOracle(n, 16.0)
    x = lat0.double().numpy()
    for i in range(n):
        t = float(orc.timesteps[i])
        v_u = 0.1 * x + 0.01 * t / 1000
        v_c = v_u + 0.05
        x = orc.step(U.cfg_combine(v_u, v_c, guidance), x)
    np.testing.assert_allclose(out.double().numpy(), x, rtol=5e-4, atol=5e-4)


def test_pipeline_input_checks():
    pipe = WanVerseCrafterPipeline(transformer=FakeTransformer(), scheduler=FlowUniPCMultistepScheduler(shift=1))
    with pytest.raises(ValueError):          # PIPE.py:588-589
        pipe(prompt_embeds=[torch.zeros(2, 64)], height=60, width=96, geoada_latents=[torch.zeros(64, 1, 8, 12)],
             mask_latents=[torch.zeros(64, 1, 8, 12)], output_type="latent")
    with pytest.raises(ValueError):
        pipe(height=64, width=96, geoada_latents=[torch.zeros(64, 1, 8, 12)], mask_latents=[torch.zeros(64, 1, 8, 12)])
    with pytest.raises(RuntimeError):        # no T5 here: prompts need embeddings
        pipe(prompt="a cat", height=64, width=96, geoada_latents=[torch.zeros(64, 1, 8, 12)],
             mask_latents=[torch.zeros(64, 1, 8, 12)], output_type="latent")


def test_bench_flop_accounting_matches_survey_8d():
    """bench.py's algorithmic FLOPs per denoise step (the numerator of the roofline fractions) against SURVEY 8d / Appendix B:
    cfg-2 2.5423, cfg-3 5.1075, cfg-4 19.737 PFLOP; the output-neutral work the engine skips is a small, positive part."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vc_bench", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for L, want in ((20280, 2.5423), (32760, 5.1075), (75600, 19.737)):
        got = b.step_flops(5120, 13824, 40, 20, L) / 1e15
        assert abs(got - want) < 5e-4 * want, (L, got, want)
    f = b.step_flops(5120, 13824, 40, 20, 32760)
    sk = b.skipped_flops(5120, 40, 20, 32760, 2, (60, 77))
    assert 0.015 * f < sk < 0.025 * f
    assert b.skipped_flops(5120, 40, 20, 32760, 2, (512, 512)) < sk          # nothing to fold when the prompts are full
    assert set(b.WORKLOADS) >= {"wan14b-81f-480x832", "wan14b-49f-480x832", "wan14b-81f-720x1280", "wan1.3b-9f-320x512", "tiny"}


def test_video_io_frame_dumps_roundtrip(tmp_path):
    """utils/video_io.py: control maps from frame dumps next to the .mp4 name (the image has no codec), resize to sample_size,
    cut to video_length; grayscale dumps become 3 channels; the writer falls back to a uint8 frame dump."""
    import numpy as np
    from PIL import Image
    from versecrafter_amd.utils.video_io import read_image, read_video, save_video
    rs = np.random.RandomState(0)
    frames = rs.randint(0, 255, (7, 16, 24, 3), dtype=np.uint8)
    np.save(tmp_path / "background_RGB.npy", frames)
    v = read_video(str(tmp_path / "background_RGB.mp4"), 5, (16, 24))
    assert v.shape == (1, 3, 5, 16, 24) and v.dtype == torch.float32
    assert torch.equal((v[0, :, 2] * 255).round().to(torch.uint8), torch.from_numpy(frames[2]).permute(2, 0, 1))
    assert read_video(str(tmp_path / "background_RGB.mp4"), 5, (32, 48)).shape == (1, 3, 5, 32, 48)
    np.savez(tmp_path / "merged_mask.npz", frames=frames[..., 0])
    m = read_video(str(tmp_path / "merged_mask.mp4"), 81, (16, 24))
    assert m.shape == (1, 3, 7, 16, 24) and torch.equal(m[:, 0], m[:, 1])
    with pytest.raises(FileNotFoundError):
        read_video(str(tmp_path / "nope.mp4"), 5, (16, 24))
    Image.fromarray(frames[0]).save(tmp_path / "0001.png")
    im = read_image(str(tmp_path / "0001.png"), (16, 24))
    assert im.shape == (1, 3, 1, 16, 24) and torch.equal((im[0, :, 0] * 255).round().to(torch.uint8), torch.from_numpy(frames[0]).permute(2, 0, 1))
    out = save_video(v, str(tmp_path / "o" / "generated_video_0.mp4"), fps=16)
    assert os.path.isfile(out)
    if out.endswith(".npy"):
        assert np.array_equal(np.load(out), frames[:5])


def test_expert_schedule_of_a_two_expert_pair():
    """BASELINE config 5 (config/wan2.2: transformer_combination_type "moe", boundary 0.875 / 0.900): which sampler steps run the
    high-noise expert -- t >= boundary * num_train_timesteps, decided once per call from the scheduler's own timesteps."""
    from versecrafter_amd.pipeline.pipeline_wan_versecrafter import expert_schedule
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    assert expert_schedule(torch.tensor([999.0, 875.0, 874.99, 10.0]), 0.875) == [True, True, False, False]
    assert expert_schedule([500, 400], 0.45, 1000) == [True, False]
    sch = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    sch.set_timesteps(50, device="cpu", shift=12.0)                     # the yaml's shift
    hi = expert_schedule(sch.timesteps, 0.875, sch.config.num_train_timesteps)
    assert hi[0] and not hi[-1] and hi == sorted(hi, reverse=True)      # one switch, high noise first
    t = sch.timesteps.float()
    assert sum(hi) == int((t >= 875).sum()) and 0 < sum(hi) < 50
    assert not any(expert_schedule(sch.timesteps, 1.1)) and all(expert_schedule(sch.timesteps, 0.0))


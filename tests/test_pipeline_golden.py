"""Sampler-pipeline parity against the REFERENCE's own pipeline file, executed in the build container
(tests/golden/make_golden_pipe.py -> tests/golden/pipe_trace.safetensors).

Pinned here (all recorded from versecrafter/pipeline/pipeline_wan_versecrafter.py itself):
  * geoada_encode_masks + geoada_latent (PIPE.py:440-488) at frame counts 81 -> 21, 49 -> 13, 9 -> 3, 5 -> 2: the oracle
    restatement, the host function and (GPU) the HIP kernel vc_op_geoada_context, all bit-exact;
  * the seq_len formula (PIPE.py:861-865);
  * __call__ (PIPE.py:652-948) around a closed-form fake DiT / VAE / T5 (tests/golden/_fake_parts.py): the geoada_context the
    DiT receives, CFG batch order [uncond, cond], per-step latent batch and broadcast timestep, guidance combine, scheduler
    hand-over, transformer.current_steps / num_inference_steps, final latents -- with and without CFG.
Unpinned, as before: the UniPC arithmetic itself (third-party; this repo's scheduler class is used on both sides)."""
import os
import sys

import pytest
import torch
from safetensors.torch import load_file

from oracle import wan_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from _fake_parts import FakeTextEncoder, FakeTokenizer, FakeTransformer, FakeVAE   # noqa: E402

CASES = ("f81", "f49", "f9", "f5")


@pytest.fixture(scope="module")
def trace(golden_dir):
    return load_file(os.path.join(golden_dir, "pipe_trace.safetensors"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_and_host_mask_planes_match_reference_bitwise(trace, name):
    from versecrafter_amd.pipeline.pipeline_wan_versecrafter import geoada_encode_masks, geoada_latent
    mask, z = trace[f"masks.{name}.mask"], trace[f"masks.{name}.z"]
    want_m, want_g = trace[f"masks.{name}.mask_latents"], trace[f"masks.{name}.geoada_context"]
    assert torch.equal(O.geoada_encode_masks(mask[0]), want_m)
    got = geoada_encode_masks(mask)
    assert len(got) == 1 and torch.equal(got[0], want_m)
    assert torch.equal(geoada_latent([z], got)[0], want_g)
    assert want_m.shape[1] == (mask.shape[2] + 3) // 4


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_geoada_context_matches_reference_bitwise(trace, name):
    """vc_op_geoada_context against what the reference's geoada_encode_masks + geoada_latent produced (bf16 inputs: the
    kernel moves values, it does no arithmetic)."""
    from versecrafter_amd import ops
    mask = trace[f"masks.{name}.mask"].bfloat16()
    z = trace[f"masks.{name}.z"].bfloat16()
    want = torch.cat([z, O.geoada_encode_masks(mask[0].float()).bfloat16()], 0)
    assert torch.equal(want[64:].float(), trace[f"masks.{name}.mask_latents"].bfloat16().float())   # same selection of pixels
    got = ops.geoada_context(z.cuda(), mask[0].cuda())
    assert torch.equal(got.cpu(), want)
    got32 = ops.geoada_context(z.cuda(), trace[f"masks.{name}.mask"][0].cuda())                      # fp32 mask input
    assert torch.equal(got32.cpu(), want)


def test_seq_len_formula(trace):
    assert int(trace["call.cfg.seq_len"]) == O.seq_len_for((16, 3, 4, 6)) == 18


@pytest.mark.parametrize("case,gs", [("cfg", 5.0), ("nocfg", 1.0)])
def test_pipeline_call_reproduces_reference_trace(trace, case, gs):
    from versecrafter_amd.pipeline import WanVerseCrafterPipeline
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    tr = FakeTransformer()
    pipe = WanVerseCrafterPipeline(tokenizer=FakeTokenizer(), text_encoder=FakeTextEncoder(64), vae=FakeVAE(), transformer=tr,
                                   scheduler=FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1,
                                                                         use_dynamic_shifting=False))
    controls = [trace[f"call.control{i}"] for i in range(4)]
    res = pipe(prompt="a red car drives past a lake", negative_prompt="blurry", height=32, width=48, video=None,
               mask_video=trace["call.mask_video"], control_video=controls, subject_ref_images=None, num_frames=9,
               num_inference_steps=5, guidance_scale=gs, generator=None, latents=trace["call.latents0"].clone(),
               output_type="latent", return_dict=True, shift=16, geoada_context_scale=0.8, max_sequence_length=48)
    assert len(tr.calls) == 5 and tr.num_inference_steps == 5
    c0 = tr.calls[0]
    assert torch.equal(c0["geoada"], trace[f"call.{case}.geoada_context"])          # control latents | mask planes, stacked
    assert c0["seq_len"] == int(trace[f"call.{case}.seq_len"])
    assert c0["ctx_lens"] == trace[f"call.{case}.ctx_lens"].tolist()                # [uncond, cond] order (PIPE.py:741)
    assert torch.allclose(torch.tensor(c0["ctx_sums"]), trace[f"call.{case}.ctx_sums"], rtol=1e-6)
    assert [c["step"] for c in tr.calls] == trace[f"call.{case}.current_steps"].tolist()
    assert torch.equal(torch.stack([c["t"] for c in tr.calls]), trace[f"call.{case}.t"])
    assert torch.allclose(torch.tensor([c["scale"] for c in tr.calls]), trace[f"call.{case}.scale"])
    xs = torch.stack([c["x"] for c in tr.calls])
    assert xs.shape == trace[f"call.{case}.x"].shape
    assert torch.allclose(xs, trace[f"call.{case}.x"], rtol=1e-6, atol=1e-6)
    assert torch.allclose(res.videos.float(), trace[f"call.{case}.final_latents"], rtol=1e-6, atol=1e-6)

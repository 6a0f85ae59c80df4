#!/usr/bin/env python3
"""Record tests/golden/pipe_trace.safetensors by running the REFERENCE's own sampler pipeline
(versecrafter/pipeline/pipeline_wan_versecrafter.py, imported as it lies under /root/reference) on CPU.

Runs only in the build container.  Third-party modules the file imports but that are absent here (diffusers, torchvision,
videox_fun ...) get in-process stand-ins (_ref_stubs.install_pipeline_stubs); the per-video models are the closed-form fakes of
_fake_parts.py; the scheduler is this repo's FlowUniPCMultistepScheduler (third-party in the reference, un-vendored: the
UniPC arithmetic itself stays unpinned, what is pinned is everything the reference's PIPELINE does around it).

Recorded (data only): the seeded inputs, and -- from the reference's __call__ (PIPE.py:652-948) -- the geoada_context the
DiT receives (PIPE.py:440-488, 766-835: mask pre-processing, 8x8 pixel-unshuffle, nearest-exact frame resize, concat with the
control latents), seq_len (PIPE.py:861-865), every step's latent batch / timestep / prompt order (PIPE.py:871-901), and the
final latents (PIPE.py:903-909); plus geoada_encode_masks on frame counts 81 -> 21, 49 -> 13, 9 -> 3 and retrieve_timesteps'
hand-over to the scheduler (PIPE.py:48-104).

    python tests/golden/make_golden_pipe.py
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
_ref_stubs.install_pipeline_stubs()

from versecrafter.pipeline import pipeline_wan_versecrafter as PIPE            # noqa: E402  (the reference's file)

from _fake_parts import FakeTextEncoder, FakeTokenizer, FakeTransformer, FakeVAE   # noqa: E402
from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler     # noqa: E402

torch.set_grad_enabled(False)
# PIPE.py:893 enters torch.cuda.device(device) around the DiT call; this container has no GPU and the run is on CPU
import contextlib                                                              # noqa: E402
torch.cuda.device = lambda device=None: contextlib.nullcontext()
out = {}


def rs_rand(rs, *shape):
    return torch.from_numpy(rs.random_sample(shape).astype(np.float32))


# ---- 1. geoada_encode_masks / geoada_latent in isolation (PIPE.py:440-488) ------------------------------------------
pipe0 = PIPE.WanVerseCrafterPipeline.__new__(PIPE.WanVerseCrafterPipeline)     # the methods below never touch self
rs = np.random.RandomState(11)
for name, (F_, H, W) in {"f81": (81, 16, 32), "f49": (49, 32, 16), "f9": (9, 32, 48), "f5": (5, 16, 16)}.items():
    mask = (rs_rand(rs, 1, 2, F_, H, W) < 0.5).float() * rs_rand(rs, 1, 2, F_, H, W)     # [B, C, F, H, W], arbitrary values
    m = pipe0.geoada_encode_masks(mask)
    z = [torch.from_numpy(rs.standard_normal((64,) + tuple(m[0].shape[1:])).astype(np.float32))]
    g = pipe0.geoada_latent(z, m)
    out[f"masks.{name}.mask"] = mask.contiguous()
    out[f"masks.{name}.z"] = z[0].contiguous()
    out[f"masks.{name}.mask_latents"] = m[0].contiguous()
    out[f"masks.{name}.geoada_context"] = g[0].contiguous()


# ---- 2. retrieve_timesteps (PIPE.py:48-104): what reaches scheduler.set_timesteps -----------------------------------
class RecSched:
    def __init__(self):
        self.seen = []

    def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kw):
        self.seen.append((num_inference_steps, timesteps, sigmas, tuple(sorted(kw.items()))))
        n = num_inference_steps if num_inference_steps is not None else len(timesteps if timesteps is not None else sigmas)
        self.timesteps = torch.arange(n, 0, -1)


rsch = RecSched()
ts, n = PIPE.retrieve_timesteps(rsch, 7, "cpu", None, mu=1)
assert rsch.seen[-1] == (7, None, None, (("mu", 1),)) and n == 7
ts, n = PIPE.retrieve_timesteps(rsch, None, "cpu", [9, 5, 1])
assert rsch.seen[-1][1] == [9, 5, 1] and n == 3
out["retrieve.n_default"] = torch.tensor([7, 3])

# ---- 3. the whole __call__ (PIPE.py:652-948) with CFG, 5 steps, 9 frames 32x48 ----------------------------------------
rs = np.random.RandomState(2025)
F_, H, W = 9, 32, 48
controls = [rs_rand(rs, 1, 3, F_, H, W) for _ in range(4)]
mask_video = (rs_rand(rs, 1, 1, F_, H, W) < 0.5).float()
mask_video[:, :, 0] = 0                                                       # CLI.py:395
latents0 = torch.from_numpy(rs.standard_normal((1, 16, 3, H // 8, W // 8)).astype(np.float32))
for case, gs in (("cfg", 5.0), ("nocfg", 1.0)):
    tr = FakeTransformer()
    sched = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    pipe = PIPE.WanVerseCrafterPipeline(tokenizer=FakeTokenizer(), text_encoder=FakeTextEncoder(64), vae=FakeVAE(),
                                        transformer=tr, scheduler=sched)
    res = pipe(prompt="a red car drives past a lake", negative_prompt="blurry", height=H, width=W, video=None,
               mask_video=mask_video, control_video=controls, subject_ref_images=None, num_frames=F_,
               num_inference_steps=5, guidance_scale=gs, generator=None, latents=latents0.clone(), output_type="latent",
               return_dict=True, shift=16, geoada_context_scale=0.8, max_sequence_length=48)
    assert len(tr.calls) == 5
    out[f"call.{case}.final_latents"] = res.videos.float().contiguous()
    out[f"call.{case}.geoada_context"] = tr.calls[0]["geoada"].contiguous()
    out[f"call.{case}.seq_len"] = torch.tensor([tr.calls[0]["seq_len"]])
    out[f"call.{case}.ctx_lens"] = torch.tensor(tr.calls[0]["ctx_lens"])
    out[f"call.{case}.ctx_sums"] = torch.tensor(tr.calls[0]["ctx_sums"])
    out[f"call.{case}.x"] = torch.stack([c["x"] for c in tr.calls]).contiguous()
    out[f"call.{case}.t"] = torch.stack([c["t"] for c in tr.calls]).contiguous()
    out[f"call.{case}.scale"] = torch.tensor([c["scale"] for c in tr.calls])
    out[f"call.{case}.current_steps"] = torch.tensor([c["step"] for c in tr.calls])
    assert tr.num_inference_steps == 5
out["call.mask_video"] = mask_video.contiguous()
out["call.latents0"] = latents0.contiguous()
for i, c in enumerate(controls):
    out[f"call.control{i}"] = c.contiguous()

path = os.path.join(HERE, "pipe_trace.safetensors")
save_file(out, path)
print("wrote", path, {k: tuple(v.shape) for k, v in out.items() if k.startswith("call.cfg")})

"""Sampler: host-side mirror of the reference's WanVerseCrafterPipeline
(versecrafter/pipeline/pipeline_wan_versecrafter.py:170-948) around the HIP denoising engine.

Same constructor (tokenizer, text_encoder, vae, transformer, scheduler) and __call__ signature.  The
denoise loop (PIPE.py:871-925) is restated in `denoise_step`: CFG batch order [uncond, cond], t.expand(B),
transformer call, `uncond + g (cond - uncond)`, scheduler.step.  The per-video stages on either side of the
loop (T5 prompt encoding, VAE encode of the control maps, VAE decode) are used through the reference's attribute contract
(`text_encoder(ids, attention_mask=mask)[0]`, `vae.encode(f)[0].mode()`, `vae.decode(z).sample`): this package's HIP mirrors
(models.WanT5EncoderModel, models.AutoencoderKLWan) or any object with that surface.  They can be bypassed with
`prompt_embeds=` / `negative_prompt_embeds=`, `geoada_latents=` and `output_type="latent"`.
"""
import math
from dataclasses import dataclass
from typing import Callable, List, Optional

import torch
import torch.nn.functional as F

from ..utils.fm_solvers_unipc import FlowUniPCMultistepScheduler


@dataclass
class WanPipelineOutput:
    videos: torch.Tensor


def geoada_encode_masks(masks, vae_stride=(4, 8, 8)) -> List[torch.Tensor]:
    """Mask planes of the GeoAdapter context (what PIPE.py:440-486 produces for ref_images=None): channel 0 of each
    per-sample [C,F,H,W] mask becomes [64, (F+3)//4, H/8, W/8] -- plane 8*i+j holds pixel (8y+i, 8x+j) of latent cell
    (y, x), and latent frame n shows source frame floor((n + 0.5) * F / T) (nearest-exact).  Host path for CPU tensors and
    pre-computed inputs; on the GPU the pipeline uses the HIP kernel (ops.geoada_context).  Pinned bit-exact against the
    reference's own function by tests/golden/pipe_trace.safetensors."""
    st, sh, sw = vae_stride
    assert sh == sw, "square spatial stride (the reference reshapes both axes by vae_stride[1])"
    planes = []
    for mask in masks:
        frames, rows, cols = mask.shape[1], 2 * (mask.shape[2] // (2 * sh)), 2 * (mask.shape[3] // (2 * sw))
        if mask.shape[2] != rows * sh or mask.shape[3] != cols * sw:
            raise RuntimeError(f"mask {tuple(mask.shape)} does not tile into {sh}x{sw} cells on an even latent grid")
        t_out = (frames + 3) // st
        src = torch.clamp(((torch.arange(t_out, dtype=torch.float64) + 0.5) * (frames / t_out)).floor().long(), max=frames - 1)
        cells = F.pixel_unshuffle(mask[0].index_select(0, src.to(mask.device)).unsqueeze(1), sh)   # [T, 64, rows, cols]
        planes.append(cells.transpose(0, 1).contiguous())
    return planes


def geoada_latent(z, m):
    """PIPE.py:488."""
    return [torch.cat([zz, mm], dim=0) for zz, mm in zip(z, m)]


def expert_schedule(timesteps, boundary: float, num_train_timesteps: int = 1000):
    """Which expert runs each sampler step of a Wan2.2-style pair: True -> the high-noise expert (t >= boundary *
    num_train_timesteps).  One host read of the schedule per call; the loop itself stays free of device read-backs."""
    thr = float(boundary) * float(num_train_timesteps)
    return [bool(v >= thr) for v in torch.as_tensor(timesteps).detach().float().cpu().tolist()]


class WanVerseCrafterPipeline:
    def __init__(self, tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None, transformer_2=None):
        """transformer_2 (this build; BASELINE config 5): the HIGH-noise expert of a Wan2.2-style pair -- the reference ships the
        configs (config/wan2.2/*.yaml: transformer_combination_type "moe", low / high noise sub-paths, `boundary`) but no code that
        reads them.  Upstream's "MoE" is two full DiTs switched by the timestep, not token routing: steps with t >= boundary *
        num_train_timesteps run transformer_2, the rest `transformer` (the low-noise expert).  Both stay resident in HBM (2 x 43.7 GB
        for 14B + GeoAdapter), each with its own per-video state; there is no exchange between them."""
        self.tokenizer, self.text_encoder, self.vae = tokenizer, text_encoder, vae
        self.transformer, self.scheduler = transformer, scheduler
        self.transformer_2 = transformer_2
        self._high_noise_steps = None   # per sampler step: True -> transformer_2 (decided on the host once per call)
        self._guidance_scale = 1.0
        self._interrupt = False
        self._device = None
        self._cfg_pair_maps = None      # the [maps, maps] stack __call__ built for classifier-free guidance

    # -- small parts of the DiffusionPipeline surface the CLI touches ---------------------------------
    def to(self, device):
        self._device = torch.device(device)
        for m in (self.transformer, self.transformer_2, self.vae, self.text_encoder):
            if m is not None and hasattr(m, "to"):
                m.to(device)
        return self

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def interrupt(self):
        return self._interrupt

    @property
    def _execution_device(self):
        if self._device is not None:
            return self._device
        return next(self.transformer.parameters()).device

    def check_inputs(self, prompt, height, width, negative_prompt, prompt_embeds=None, negative_prompt_embeds=None):
        """PIPE.py:579-632."""
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")

    def encode_prompt(self, prompt, negative_prompt, do_cfg, prompt_embeds=None, negative_prompt_embeds=None,
                      max_sequence_length=512, device=None):
        """PIPE.py:284-363.  Returns two lists of [len_i, text_dim] tensors."""
        def enc(p):
            if self.tokenizer is None or self.text_encoder is None:
                raise RuntimeError("no tokenizer / text_encoder: construct the pipeline with them "
                                   "(models.WanT5EncoderModel) or pass prompt_embeds and negative_prompt_embeds")
            p = [p] if isinstance(p, str) else p
            ti = self.tokenizer(p, padding="max_length", max_length=max_sequence_length, truncation=True,
                                add_special_tokens=True, return_tensors="pt")
            ids, mask = ti.input_ids.to(device), ti.attention_mask.to(device)
            lens = mask.gt(0).sum(dim=1).long()
            emb = self.text_encoder(ids, attention_mask=mask)[0]
            return [u[:v] for u, v in zip(emb, lens)]
        if prompt_embeds is None:
            prompt_embeds = enc(prompt)
        if do_cfg and negative_prompt_embeds is None:
            negative_prompt_embeds = enc(negative_prompt if negative_prompt is not None else "")
        as_list = lambda e: list(e) if isinstance(e, (list, tuple)) else [u for u in e]
        return as_list(prompt_embeds), (as_list(negative_prompt_embeds) if do_cfg else None)

    def geoada_encode_multi_frames(self, multi_frames):
        """PIPE.py:397-438 (ref_images=None): VAE-encode each control video, concat per sample on channels."""
        if self.vae is None:
            raise RuntimeError("no VAE: construct the pipeline with vae=models.AutoencoderKLWan(...) or pass geoada_latents")
        enc = [self.vae.encode(f)[0].mode() for f in multi_frames]
        if hasattr(self.vae, "release_workspace"):
            self.vae.release_workspace()          # tens of GB of activation buffers: not needed again until the final decode
        return [torch.cat(items, dim=0) for items in zip(*enc)]

    def prepare_latents(self, batch_size, channels, shape_thw, dtype, device, generator, latents=None):
        """PIPE.py:365-395."""
        shape = (batch_size, channels, *shape_thw)
        if latents is None:
            gdev = generator.device if generator is not None else device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device)
        if hasattr(self.scheduler, "init_noise_sigma"):
            latents = latents * self.scheduler.init_noise_sigma
        return latents

    # -- the hot loop body ----------------------------------------------------------------------------
    def denoise_step(self, i, t, latents, in_prompt_embeds, geoada_context_input, seq_len, do_cfg,
                     geoada_context_scale=1.0):
        """One iteration of PIPE.py:871-925."""
        model = self.transformer
        if self.transformer_2 is not None and self._high_noise_steps is not None and self._high_noise_steps[i]:
            model = self.transformer_2
        model.current_steps = i
        latent_model_input = torch.cat([latents] * 2) if do_cfg else latents
        if hasattr(self.scheduler, "scale_model_input"):
            latent_model_input = self.scheduler.scale_model_input(latent_model_input, t)
        timestep = t.expand(latent_model_input.shape[0])
        if do_cfg and latents.shape[0] == 1 and self._cfg_pair_maps is geoada_context_input and \
                hasattr(model, "assert_cfg_pair"):
            model.assert_cfg_pair(latent_model_input)               # [u, u] built right here and in __call__: no device check
        noise_pred = model(x=latent_model_input, context=in_prompt_embeds, t=timestep,
                           geoada_context=geoada_context_input, seq_len=seq_len,
                           geoada_context_scale=geoada_context_scale)
        if (hasattr(self.scheduler, "step_cfg") and latents.is_cuda and latents.dtype == torch.bfloat16 and
                noise_pred.dtype == torch.bfloat16 and (not do_cfg or noise_pred.shape[0] == 2 * latents.shape[0])):
            # CFG combine + x0 + UniPC corrector / predictor in one HIP kernel; bit-identical to the two steps below
            return self.scheduler.step_cfg(noise_pred, t, latents, self.guidance_scale if do_cfg else None)
        if do_cfg:
            noise_pred_uncond, noise_pred_text = noise_pred.chunk(2)
            noise_pred = noise_pred_uncond + self.guidance_scale * (noise_pred_text - noise_pred_uncond)
        return self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]

    @torch.no_grad()
    def __call__(self, prompt=None, negative_prompt=None, height: int = 480, width: int = 720, video=None,
                 mask_video=None, control_video=None, subject_ref_images=None, num_frames: int = 49,
                 num_inference_steps: int = 50, timesteps=None, guidance_scale: float = 6,
                 num_videos_per_prompt: int = 1, eta: float = 0.0, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, output_type: str = "numpy", return_dict: bool = False,
                 callback_on_step_end: Optional[Callable] = None, attention_kwargs=None,
                 callback_on_step_end_tensor_inputs=("latents",), max_sequence_length: int = 512,
                 comfyui_progressbar: bool = False, shift: int = 5, geoada_context_scale: float = 1.0,
                 geoada_latents=None, mask_latents=None, boundary: float = 0.875):
        """PIPE.py:652-948.  Extensions (keyword-only in practice): `geoada_latents` (list of [64,T,h,w] control
        latents, replacing the VAE encode), `mask_latents` (list of [64,T,h,w]) and `boundary` (with a transformer_2: the fraction
        of num_train_timesteps at and above which the high-noise expert runs; config/wan2.2/wan_civitai_t2v.yaml: 0.875)."""
        if subject_ref_images is not None:
            raise NotImplementedError("subject_ref_images is not used by the VerseCrafter CLI (CLI.py:431)")
        num_videos_per_prompt = 1                                                   # PIPE.py:696
        self.check_inputs(prompt, height, width, negative_prompt, prompt_embeds, negative_prompt_embeds)
        self._guidance_scale = guidance_scale
        self._interrupt = False
        # A new video: every expert's TeaCache starts from step 0 and the engines forget their stored residuals.  With ONE expert the
        # reference's own reset (cnt == num_steps at the end of forward, VC.py:438-441) already does this; with two experts each gate only
        # counts the steps its expert ran, never reaches num_steps, and call 2 would otherwise start past num_skip_start_steps with the
        # previous video's modulated input and residual (round-3 advisor finding).  A TeaCache shared by the pair is reset once.
        seen = set()
        for m in (self.transformer, self.transformer_2):
            tc = getattr(m, "teacache", None)
            if tc is not None and id(tc) not in seen:
                seen.add(id(tc))
                tc.reset()
            if m is not None and hasattr(m, "reset_residuals"):
                m.reset_residuals()
        batch_size = 1 if isinstance(prompt, str) else (len(prompt) if prompt is not None else len(prompt_embeds))
        device = self._execution_device
        weight_dtype = next(self.transformer.parameters()).dtype
        do_cfg = guidance_scale > 1.0                                               # PIPE.py:726
        pe, ne = self.encode_prompt(prompt, negative_prompt, do_cfg, prompt_embeds, negative_prompt_embeds,
                                    max_sequence_length, device)
        pe = [u.to(device=device, dtype=weight_dtype) for u in pe]
        in_prompt_embeds = ([u.to(device=device, dtype=weight_dtype) for u in ne] + pe) if do_cfg else pe  # PIPE.py:741

        if isinstance(self.scheduler, FlowUniPCMultistepScheduler):                # PIPE.py:750-752
            self.scheduler.set_timesteps(num_inference_steps, device=device, shift=shift)
        else:
            self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps

        # control maps -> geoada_context (PIPE.py:766-835)
        if geoada_latents is None:
            if control_video is None:
                raise ValueError("control_video (or geoada_latents) is required")
            vids = [(cv.to(torch.float32) * 2.0 - 1.0).to(dtype=weight_dtype, device=device) for cv in control_video]
            geoada_latents = self.geoada_encode_multi_frames(vids)
        geoada_latents = [z.to(device=device, dtype=weight_dtype) for z in geoada_latents]
        if mask_latents is None and mask_video is None:
            raise ValueError("mask_video (or mask_latents) is required")
        if mask_latents is None and device.type == "cuda" and weight_dtype == torch.bfloat16:
            # PIPE.py:440-488 in one HIP kernel per sample (the reference tiles the mask to 3 channels and reads channel 0)
            from .. import ops
            mv = mask_video.to(device=device, dtype=torch.float32).to(weight_dtype)
            geoada_context = [ops.geoada_context(z, m) for z, m in zip(geoada_latents, mv)]
        else:
            if mask_latents is None:
                mc = torch.tile(mask_video.to(torch.float32), [1, 3, 1, 1, 1]).to(dtype=weight_dtype, device=device)
                mask_latents = geoada_encode_masks(mc)
            mask_latents = [m.to(device=device, dtype=weight_dtype) for m in mask_latents]
            geoada_context = geoada_latent(geoada_latents, mask_latents)            # [128, T, h, w] per sample

        T, h, w = geoada_latents[0].shape[1:]
        latent_channels = getattr(getattr(self.vae, "config", None), "latent_channels", 16)
        latents = self.prepare_latents(batch_size * num_videos_per_prompt, latent_channels, (T, h, w), weight_dtype,
                                       device, generator, latents)
        seq_len = math.ceil((h * w) / (self.transformer.config.patch_size[1] * self.transformer.config.patch_size[2]) * T)
        self.transformer.num_inference_steps = num_inference_steps                  # PIPE.py:869
        self._high_noise_steps = None
        if self.transformer_2 is not None:
            self.transformer_2.num_inference_steps = num_inference_steps
            self._high_noise_steps = expert_schedule(timesteps, boundary, self.scheduler.config.num_train_timesteps)
        # the reference re-stacks this every step (PIPE.py:883-887); it is step-invariant
        geoada_context_input = torch.stack(geoada_context * 2) if do_cfg else torch.stack(geoada_context)
        self._cfg_pair_maps = geoada_context_input if (do_cfg and len(geoada_context) == 1) else None

        for i, t in enumerate(timesteps):                                           # PIPE.py:871
            if self.interrupt:
                continue
            latents = self.denoise_step(i, t, latents, in_prompt_embeds, geoada_context_input, seq_len, do_cfg,
                                        geoada_context_scale)
            if callback_on_step_end is not None:
                outs = callback_on_step_end(self, i, t, {"latents": latents})
                latents = outs.pop("latents", latents)

        if output_type == "latent":
            video = latents
        else:                                                                       # PIPE.py:550-555, 932-946
            if self.vae is None:
                raise RuntimeError("no VAE to decode with: use output_type='latent'")
            frames = self.vae.decode(latents.to(self.vae.dtype)).sample
            video = (frames / 2 + 0.5).clamp(0, 1).cpu().float()
            if hasattr(self.vae, "release_workspace"):
                self.vae.release_workspace()
        return WanPipelineOutput(videos=video)

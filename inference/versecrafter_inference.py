#!/usr/bin/env python3
"""VerseCrafter inference CLI on the MI355X engine.

Same command-line flags and defaults as the reference's inference/versecrafter_inference.py:44-69; the knobs the
reference hard-codes at module level (:89-161) are exposed as extra flags with identical defaults.  Launch as the
reference does (`torchrun --nproc-per-node=N inference/versecrafter_inference.py ...`); ulysses_degree*ring_degree
must equal N (the hybrid is run as pure Ulysses of that degree).

A full run mirrors the reference's flow (CLI.py:187-465): DiT + GeoAdapter from --transformer_path, the Wan VAE and the umT5
encoder from --model_name (or --vae_path / --text_encoder_path), the four control maps + merged mask + first frame from
--rendering_maps_path / --input_image_path, the video written to --save_path.  All three models run on the HIP engine
(versecrafter_amd.models).  The build image has no video codec library: control maps are read from frame dumps next to the .mp4 files,
through any importable decoder, or -- the .mp4 files this repo's renderer CLI writes -- through the package's own H.264 I_PCM reader;
the result is written as .mp4 the same way -- see versecrafter_amd/utils/video_io.py, utils/mp4_pcm.py.  --synthetic_inputs runs the denoising engine on random control latents /
prompt embeddings of the right shapes (no VAE, no T5, latents saved as .safetensors).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from versecrafter_amd.dist import set_multi_gpus_devices
from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
from versecrafter_amd.pipeline import WanVerseCrafterPipeline
from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler

NEGATIVE_PROMPT = (
    "Bright tones, overexposed, static, blurred details, subtitles, style, works, paintings, images, static, overall "
    "gray, worst quality, low quality, JPEG compression residue, ugly, incomplete, extra fingers, poorly drawn hands, "
    "poorly drawn faces, deformed, disfigured, misshapen limbs, fused fingers, still picture, messy background, three "
    "legs, many people in the background, walking backwards")
TEACACHE_COEFFICIENTS_14B = [8.10705460e+03, 2.13393892e+03, -3.72934672e+02, 1.66203073e+01, -4.17769401e-02]


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Video generation inference script")
    # ---- the reference's flags (CLI.py:44-69) ----
    p.add_argument("--transformer_path", type=str, default="model/VerseCrafter")
    p.add_argument("--save_path", type=str, default="dataset/inference")
    p.add_argument("--rendering_maps_path", type=str, required=True)
    p.add_argument("--prompt", type=str, required=True)
    p.add_argument("--input_image_path", type=str, required=True)
    p.add_argument("--num_inference_steps", type=int, default=50)
    p.add_argument("--sample_size", type=str, default="720,1280")
    p.add_argument("--ulysses_degree", type=int, default=2)
    p.add_argument("--ring_degree", type=int, default=2)
    p.add_argument("--cfg_degree", type=int, default=1, choices=(1, 2),
                   help="(this build) ranks that split the classifier-free-guidance pair: world = cfg_degree * ulysses_degree * "
                        "ring_degree; with 2 GPUs, `--cfg_degree 2 --ulysses_degree 1 --ring_degree 1` needs no sequence exchange")
    p.add_argument("--sp_layout", type=str, default="auto", choices=("auto", "as_given"),
                   help="(this build) auto: ulysses_degree x ring_degree runs as pure Ulysses of the product wherever the head count divides "
                        "by it (14B: 40 heads), else as the hybrid; as_given: exactly the U x R of the flags (the reference's documented "
                        "layout, inference.sh:62-71), e.g. to time 2 x 4 against Ulysses-8 on hardware")
    p.add_argument("--guidance_scale", type=float, default=5.0)
    p.add_argument("--seed", type=int, default=2025)
    p.add_argument("--fps", type=int, default=16)
    # ---- module-level constants of the reference (CLI.py:89-161), same defaults ----
    p.add_argument("--enable_teacache", type=int, default=1)
    p.add_argument("--teacache_threshold", type=float, default=0.10)
    p.add_argument("--num_skip_start_steps", type=int, default=5)
    p.add_argument("--cfg_skip_ratio", type=float, default=0.0)
    p.add_argument("--enable_riflex", type=int, default=0)
    p.add_argument("--riflex_k", type=int, default=6)
    p.add_argument("--shift", type=float, default=16)
    p.add_argument("--video_length", type=int, default=81)
    p.add_argument("--geoada_context_scale", type=float, default=1.0)
    p.add_argument("--geoada_in_dim", type=int, default=128)
    p.add_argument("--model_name", type=str, default="model/Wan2.1-T2V-14B")
    # ---- this build ----
    p.add_argument("--synthetic_inputs", action="store_true",
                   help="random control latents / prompt embeddings instead of VAE + T5 + mp4 decoding")
    p.add_argument("--synthetic_model", type=str, default=None, choices=[None, "14b", "1.3b", "tiny"],
                   help="random weights of the named architecture instead of --transformer_path")
    p.add_argument("--text_encoder_path", type=str, default=None,
                   help="umT5 checkpoint (models_t5_umt5-xxl-enc-bf16.pth or .safetensors): encode --prompt and the fixed negative prompt "
                        "with the HIP text encoder instead of random embeddings (needs --tokenizer_path)")
    p.add_argument("--tokenizer_path", type=str, default=None, help="local directory of the google/umt5-xxl tokenizer files")
    p.add_argument("--prompt_embeds_path", type=str, default=None,
                   help=".safetensors with 'prompt_embeds' [n,4096] and 'negative_prompt_embeds' [m,4096] (umT5 outputs computed "
                        "elsewhere): used instead of --text_encoder_path")
    p.add_argument("--vae_path", type=str, default=None,
                   help="Wan2.1_VAE.pth / .safetensors (default: <model_name>/Wan2.1_VAE.pth, wan_civitai.yaml:9)")
    p.add_argument("--vae_kwargs", type=str, default="{}", help="JSON overrides of the VAE hyper-parameters (tests: {\"dim\": 32})")
    p.add_argument("--output_latents", type=int, default=0, help="1: also save the final latents as .safetensors")
    p.add_argument("--transformer_high_noise_path", type=str, default=None,
                   help="(this build; BASELINE config 5) checkpoint of the HIGH-noise expert of a Wan2.2-style pair "
                        "(config/wan2.2/*.yaml: transformer_combination_type \"moe\"); --transformer_path is then the low-noise expert and "
                        "steps with t >= --boundary * 1000 run the high-noise one.  Both experts stay resident in HBM")
    p.add_argument("--boundary", type=float, default=0.875, help="config/wan2.2/wan_civitai_t2v.yaml: 0.875 (i2v: 0.900)")
    p.add_argument("--fp8_linear", type=int, default=0,
                   help="(this build; BASELINE config 5 names fp8 MFMA) 1: the blocks' nn.Linear layers run in fp8 (e4m3 weights per output "
                        "channel, activations per token, fp32 accumulation); attention and everything else stay bf16.  The reference "
                        "computes in bf16: results then differ from it by the quantisation error")
    p.add_argument("--fp8_attention", type=int, default=-1, choices=(-1, 0, 1),
                   help="(this build; BASELINE config 5 names fp8 MFMA) 1 / 0: the blocks' SELF-attention runs in fp8 -- q, k, v and the softmax "
                        "weights as e4m3 under one power-of-two scale per 32 elements, both products on the block-scaled MFMA; 1 makes the weights' "
                        "bytes from the piecewise-linear 2^x, 0 from v_exp_f32.  -1 (default): bf16 attention, as the reference")
    p.add_argument("--synthetic_high_noise_expert", action="store_true",
                   help="with --synthetic_model: a second random model (seed 1) as the high-noise expert")
    p.add_argument("--control_latents_path", type=str, default=None,
                   help=".safetensors with 'geoada_latents' [64,T,h,w] (VAE latents of the 4 control videos) and "
                        "'mask_video' [1,F,H,W] (merged mask): skips the VAE encode of the control maps")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    height, width = [int(x) for x in args.sample_size.split(",")]
    device = set_multi_gpus_devices(args.ulysses_degree, args.ring_degree, cfg_degree=args.cfg_degree,
                                    ring_as_given=args.sp_layout == "as_given")
    if device.type != "cuda":
        raise SystemExit("versecrafter_amd needs an MI355X (HIP) device: there is no CPU path")
    weight_dtype = torch.bfloat16

    if args.synthetic_model:
        dims = {"14b": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40),
                "1.3b": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30),
                "tiny": dict(dim=256, ffn_dim=512, num_heads=2, num_layers=4)}[args.synthetic_model]
        torch.manual_seed(0)
        transformer = VerseCrafterWanTransformer3DModel(geoada_in_dim=args.geoada_in_dim, param_device=device,
                                                        param_dtype=weight_dtype, **dims)
        transformer.init_weights(zero_init_outputs=False)
    else:
        transformer = VerseCrafterWanTransformer3DModel.from_pretrained(
            args.transformer_path,
            transformer_additional_kwargs={"geoada_in_dim": args.geoada_in_dim,
                                           "dict_mapping": {"in_dim": "in_channels", "dim": "hidden_size"}},
            low_cpu_mem_usage=True, torch_dtype=weight_dtype).to(device)

    transformer_2 = None
    if args.synthetic_model and args.synthetic_high_noise_expert:
        torch.manual_seed(1)
        transformer_2 = VerseCrafterWanTransformer3DModel(geoada_in_dim=args.geoada_in_dim, param_device=device,
                                                          param_dtype=weight_dtype, **dims)
        transformer_2.init_weights(zero_init_outputs=False)
    elif args.transformer_high_noise_path:
        transformer_2 = VerseCrafterWanTransformer3DModel.from_pretrained(
            args.transformer_high_noise_path,
            transformer_additional_kwargs={"geoada_in_dim": args.geoada_in_dim,
                                           "dict_mapping": {"in_dim": "in_channels", "dim": "hidden_size"}},
            low_cpu_mem_usage=True, torch_dtype=weight_dtype).to(device)

    vae = text_encoder = tokenizer = None
    vae_path = args.vae_path or os.path.join(args.model_name, "Wan2.1_VAE.pth")
    if not args.synthetic_inputs and os.path.isfile(vae_path):                      # CLI.py:220-223
        import json
        from versecrafter_amd.models import AutoencoderKLWan
        vae = AutoencoderKLWan.from_pretrained(vae_path, additional_kwargs=dict(
            temporal_compression_ratio=4, spatial_compression_ratio=8, **json.loads(args.vae_kwargs))).to(weight_dtype)
    if args.text_encoder_path:                                                      # CLI.py:238-249
        if not args.tokenizer_path:
            raise SystemExit("--text_encoder_path needs --tokenizer_path (local google/umt5-xxl tokenizer files)")
        from transformers import AutoTokenizer
        from versecrafter_amd.models import WanT5EncoderModel
        tokenizer = AutoTokenizer.from_pretrained(args.tokenizer_path)
        text_encoder = WanT5EncoderModel.from_pretrained(
            args.text_encoder_path, additional_kwargs=dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64,
                                                           num_layers=24, num_buckets=32, shared_pos=False, dropout=0.0),
            low_cpu_mem_usage=True, torch_dtype=weight_dtype).eval()
    if not args.synthetic_inputs and text_encoder is None and not args.prompt_embeds_path:
        raise SystemExit("a real run needs --text_encoder_path and --tokenizer_path (umT5 checkpoint + local tokenizer files) or "
                         "--prompt_embeds_path; otherwise pass --synthetic_inputs")
    if not args.synthetic_inputs and vae is None and not args.control_latents_path:
        raise SystemExit(f"no VAE checkpoint at {vae_path}: pass --vae_path (or --control_latents_path for pre-encoded control "
                         "maps, or --synthetic_inputs)")

    scheduler = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)  # CLI.py:252-261
    pipeline = WanVerseCrafterPipeline(tokenizer=tokenizer, text_encoder=text_encoder, vae=vae,
                                       transformer=transformer, scheduler=scheduler, transformer_2=transformer_2)
    experts = [m for m in (transformer, transformer_2) if m is not None]
    if args.ulysses_degree * args.ring_degree * args.cfg_degree > 1:
        for m in experts:
            m.enable_multi_gpus_inference()                                         # CLI.py:271-273
    pipeline.to(device)
    for m in experts:
        if args.fp8_linear:
            m.enable_fp8_linear()
        if args.fp8_attention >= 0:
            m.enable_fp8_attention(True, args.fp8_attention)
        if args.enable_teacache:                                                    # CLI.py:305-313
            m.enable_teacache(TEACACHE_COEFFICIENTS_14B, args.num_inference_steps, args.teacache_threshold,
                              num_skip_start_steps=args.num_skip_start_steps, offload=False)
        if args.cfg_skip_ratio:
            m.enable_cfg_skip(args.cfg_skip_ratio, args.num_inference_steps)
        if args.enable_riflex:
            m.enable_riflex(k=args.riflex_k, L_test=(args.video_length - 1) // 4 + 1)

    generator = torch.Generator(device=device).manual_seed(args.seed)               # CLI.py:319
    T, h, w = (args.video_length - 1) // 4 + 1, height // 8, width // 8
    g = torch.Generator().manual_seed(args.seed)
    if args.control_latents_path:
        from safetensors.torch import load_file
        cl = load_file(args.control_latents_path)
        control = dict(geoada_latents=[cl["geoada_latents"]], mask_video=cl["mask_video"][None].float())
    elif not args.synthetic_inputs:
        # CLI.py:351-403: four control maps, the merged mask (channel 0, frame 0 cleared), the first frame pasted into map 0
        from versecrafter_amd.utils.video_io import read_image, read_video
        if not os.path.isdir(args.rendering_maps_path):
            raise SystemExit(f"Annotation path not found: {args.rendering_maps_path}")
        size = (height, width)
        control_videos = []
        for name in ("background_RGB.mp4", "background_depth.mp4", "3D_gaussian_RGB.mp4", "3D_gaussian_depth.mp4"):
            try:
                control_videos.append(read_video(os.path.join(args.rendering_maps_path, name), args.video_length, size))
            except FileNotFoundError:
                # the reference appends a zero placeholder only after the first map (CLI.py:378-382) and then fails on the
                # channel count when map 0 is the missing one; say what is missing instead
                if not control_videos:
                    raise SystemExit(f"control map {name} is missing under {args.rendering_maps_path} (the reference needs all "
                                     "four: geoada_in_dim = 4 x 16 latent channels + 64 mask planes)")
                print(f"Warning: Control video not found: {name}")
                control_videos.append(torch.zeros_like(control_videos[0]))
        try:
            mask = read_video(os.path.join(args.rendering_maps_path, "merged_mask.mp4"), args.video_length, size)[:, :1]
            mask[:, :, 0] = 0.0
        except FileNotFoundError:
            mask = torch.ones_like(control_videos[0][:, :1]) * 255
        control_videos[0][:, :, 0] = read_image(args.input_image_path, size).squeeze(2)
        control = dict(control_video=control_videos, mask_video=mask)
    else:
        ctrl = torch.randn(64, T, h, w, generator=g)
        mask = (torch.rand(64, T, h, w, generator=g) < 0.5).float()
        mask[:, 0] = 0                                                              # CLI.py:395
        control = dict(geoada_latents=[ctrl], mask_latents=[mask])
    if text_encoder is not None:
        embeds = dict(prompt=args.prompt, negative_prompt=NEGATIVE_PROMPT)                     # CLI.py:421-423
    elif args.prompt_embeds_path:
        from safetensors.torch import load_file
        pe = load_file(args.prompt_embeds_path)
        embeds = dict(prompt_embeds=[pe["prompt_embeds"]], negative_prompt_embeds=[pe["negative_prompt_embeds"]])
    else:
        embeds = dict(prompt_embeds=[torch.randn(77, transformer.text_dim, generator=g)],
                      negative_prompt_embeds=[torch.randn(60, transformer.text_dim, generator=g)])
    t0 = time.time()
    latents_box = {}

    def keep_latents(pipe, i, t, kw):                     # the final latents, for --output_latents / runs without a VAE
        latents_box["latents"] = kw["latents"]
        return {}
    decode = vae is not None
    sample = pipeline(height=height, width=width, num_frames=args.video_length, generator=generator,
                      guidance_scale=args.guidance_scale, num_inference_steps=args.num_inference_steps,
                      shift=args.shift, geoada_context_scale=args.geoada_context_scale,
                      output_type="numpy" if decode else "latent", callback_on_step_end=keep_latents, boundary=args.boundary,
                      **control, **embeds).videos
    torch.cuda.synchronize()
    dt = time.time() - t0
    rank = int(os.environ.get("RANK", 0))
    if rank == 0:                                                                   # CLI.py:440-465
        os.makedirs(args.save_path, exist_ok=True)
        index = len([p for p in os.listdir(args.save_path) if p.startswith("generated_video_")])
        msg = f"{args.num_inference_steps} steps in {dt:.1f} s total (TeaCache {'on' if args.enable_teacache else 'off'})"
        if decode:
            from versecrafter_amd.utils.video_io import save_video
            out = save_video(sample, os.path.join(args.save_path, f"generated_video_{index}.mp4"), fps=args.fps)
            print(args.prompt)
            print(f"{msg}; video -> {out}")
        if not decode or args.output_latents:
            from safetensors.torch import save_file
            out = os.path.join(args.save_path, f"generated_latents_{index if decode else 0}.safetensors")
            save_file({"latents": latents_box["latents"].float().cpu().contiguous()}, out)
            print(f"{msg}; latents -> {out}")


if __name__ == "__main__":
    from versecrafter_amd.dist import SequenceParallelStall
    try:
        main()
    except SequenceParallelStall as stall:
        # the rendezvous of the engine's communicators never returned on this rank: a helper thread is still inside it.  No interpreter
        # shutdown (it would destroy the engine under that thread): say why and leave at once, non-zero; torchrun ends the other ranks
        print(f"versecrafter_inference: {stall}", file=sys.stderr, flush=True)
        os._exit(3)

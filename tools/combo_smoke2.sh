#!/bin/bash
# second set of combinations (GPU box)
cd "$(dirname "$0")/.."
CLI="python inference/versecrafter_inference.py --rendering_maps_path x --prompt p --input_image_path x.png --num_inference_steps 6 --sample_size 64,96 --synthetic_inputs --synthetic_model tiny --num_skip_start_steps 2 --output_latents 1"
echo "== single frame (video_length 1)"; $CLI --ulysses_degree 1 --ring_degree 1 --video_length 1 --save_path /tmp/c1 2>&1 | tail -2
echo "== 5 frames"; $CLI --ulysses_degree 1 --ring_degree 1 --video_length 5 --save_path /tmp/c2 2>&1 | tail -1
echo "== torchrun 2 ranks (gloo) + fp8 + two experts"; VC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 inference/versecrafter_inference.py --rendering_maps_path x --prompt p --input_image_path x.png --num_inference_steps 6 --sample_size 64,96 --video_length 9 --synthetic_inputs --synthetic_model tiny --num_skip_start_steps 2 --output_latents 1 --ulysses_degree 2 --ring_degree 1 --fp8_linear 1 --synthetic_high_noise_expert --shift 12 --save_path /tmp/c3 2>&1 | tail -2
echo "== batch of two videos with CFG (B = 4) through the pipeline API"
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
from versecrafter_amd.pipeline import WanVerseCrafterPipeline
from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
torch.manual_seed(0)
dev = torch.device("cuda", 0)
m = VerseCrafterWanTransformer3DModel(param_device=dev, param_dtype=torch.bfloat16, geoada_in_dim=128, dim=256, ffn_dim=512, num_heads=2, num_layers=4, text_dim=64, text_len=48)
m.init_weights(zero_init_outputs=False)
g = torch.Generator().manual_seed(1)
T, h, w = 3, 8, 12
def run(n):
    pipe = WanVerseCrafterPipeline(transformer=m, scheduler=FlowUniPCMultistepScheduler(shift=1))
    lat = torch.randn(2, 16, T, h, w, generator=torch.Generator().manual_seed(2)).to(dev, torch.bfloat16)[:n]
    geo = [torch.randn(64, T, h, w, generator=torch.Generator().manual_seed(3 + i)).to(dev, torch.bfloat16) for i in range(n)]
    msk = [(torch.rand(64, T, h, w, generator=torch.Generator().manual_seed(5 + i)) < 0.5).to(dev, torch.bfloat16) for i in range(n)]
    pe = [torch.randn(20 + i, 64, generator=torch.Generator().manual_seed(7 + i)).to(dev, torch.bfloat16) for i in range(n)]
    ne = [torch.randn(11 + i, 64, generator=torch.Generator().manual_seed(9 + i)).to(dev, torch.bfloat16) for i in range(n)]
    return pipe(prompt_embeds=pe, negative_prompt_embeds=ne, height=h * 8, width=w * 8, geoada_latents=geo, mask_latents=msk, num_inference_steps=4,
                guidance_scale=5.0, shift=16, latents=lat.clone(), output_type="latent").videos
two = run(2)
one = run(1)
torch.cuda.synchronize()
print("B=4 result", tuple(two.shape), bool(torch.isfinite(two.float()).all()), "first video equals the single-video run:", bool(torch.equal(two[:1], one)))
PY

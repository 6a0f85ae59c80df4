#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/combo_smoke.sh'): flag combinations of the two CLIs / the bench that no single test covers --
# cfg_skip + TeaCache + fp8 + two experts, RIFLEx, guidance off, the 1.3B workload in fp8 and bf16, a 2-rank gloo rehearsal in fp8.
set -e
cd "$(dirname "$0")/.."
CLI="python inference/versecrafter_inference.py --rendering_maps_path x --prompt p --input_image_path x.png --ulysses_degree 1 --ring_degree 1 --num_inference_steps 8 --sample_size 64,96 --video_length 9 --synthetic_inputs --synthetic_model tiny --num_skip_start_steps 2 --output_latents 1"
echo "== cfg_skip + teacache + fp8 + two experts"; $CLI --save_path /tmp/o1 --cfg_skip_ratio 0.25 --fp8_linear 1 --synthetic_high_noise_expert --shift 12 2>&1 | tail -2
echo "== the same + fp8 self-attention (pmode 1, then the v_exp_f32 variant)"; $CLI --save_path /tmp/o1b --cfg_skip_ratio 0.25 --fp8_linear 1 --fp8_attention 1 --synthetic_high_noise_expert --shift 12 2>&1 | tail -1; $CLI --save_path /tmp/o1c --fp8_attention 0 2>&1 | tail -1
echo "== riflex"; $CLI --save_path /tmp/o2 --enable_riflex 1 2>&1 | tail -2
echo "== teacache off, 1 step-size"; $CLI --save_path /tmp/o3 --enable_teacache 0 --guidance_scale 1.0 2>&1 | tail -2
echo "== bench 1.3b fp8"; python bench.py --workload wan1.3b-9f-320x512 --fp8-linear --steps 3 --warmup 2 --no-cpu-baseline --no-teacache-line 2>/dev/null | cut -c1-330
echo "== bench 1.3b bf16"; python bench.py --workload wan1.3b-9f-320x512 --steps 3 --warmup 2 --no-cpu-baseline --no-teacache-line 2>/dev/null | cut -c1-200
echo "== bench 2 ranks gloo tiny fp8"; python bench.py --gpus 2 --workload tiny --backend gloo --fp8-linear --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-330
echo "== bench 4 ranks gloo tiny4h, three layouts, fp8 linear + fp8 self-attention"; python bench.py --gpus 4 --workload tiny4h --backend gloo --fp8-linear --fp8-attn 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-500

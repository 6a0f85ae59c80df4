#!/usr/bin/env python3
"""Benchmark of the VerseCrafter denoise step on MI355X:  python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): denoise-steps/sec, Wan2.1-14B + GeoAdapter, 81 frames x 480p (480x832 -> latent
[16,21,60,104], 32760 tokens), CFG pair batched (B=2), TeaCache and cfg_skip off -- one "step" is one
iteration of the reference's sampler loop (pipeline_wan_versecrafter.py:871-925): transformer forward at B=2,
CFG combine, UniPC scheduler step.  Synthetic inputs and random weights of that architecture; all
inputs are resident in HBM before the timed region.  For N > 1 the frames x h x w token sequence is sharded
Ulysses-style over the ranks, one process per GPU, the exchanges on RCCL communicators the engine owns: total work is
fixed, so "scaling" is "strong".  `python bench.py --gpus N` without a launcher around it starts its own N ranks (torchrun's
environment contract) and forwards rank 0's line; under torchrun it runs as a rank.  Rank 0 prints ONE JSON line.

The same line carries
  roofline     : the dominant kernel class of the timed region (HIP events around every launch of the class,
                 on the launch stream) -- algorithmic FLOPs / summed duration vs the dense bf16 MFMA peak;
  breakdown    : the same for every kernel class, plus the whole-step fraction of the MFMA roofline;
  cpu_baseline : the CPU oracle (oracle/wan_oracle.py, fp32 PyTorch) timed on this box's host cores on a bounded
                 sample (one main + one adapter block at the 14B width on a 1536-token slice) and extrapolated by FLOPs.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF dense)

WORKLOADS = {
    # name: (model kwargs, frames, height, width)
    "wan14b-81f-480x832": (dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40), 81, 480, 832),
    "wan14b-49f-480x832": (dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40), 49, 480, 832),
    "wan14b-81f-720x1280": (dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40), 81, 720, 1280),   # BASELINE config 4
    "wan1.3b-9f-320x512": (dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30), 9, 320, 512),
    "tiny": (dict(dim=256, ffn_dim=512, num_heads=2, num_layers=4, text_dim=64, text_len=48), 9, 64, 96),
}


def step_flops(d, ffn, NL, NA, L, B=2, text_len=512, text_dim=4096):
    """F_step = B [ (NL+NA) F_blk + (NA+1) 2 L d^2 + F_embed ]."""
    f_blk = 8 * L * d * d + 4 * L * L * d + 4 * L * d * d + 4 * text_len * d * d + 4 * L * text_len * d + 4 * L * d * ffn
    f_embed = (2 * L * 64 * d + 2 * L * 512 * d + 2 * text_len * text_dim * d + 2 * text_len * d * d + 2 * L * d * 64 +
               (2 * 256 * d + 2 * d * d + 12 * d * d))
    return B * ((NL + NA) * f_blk + (NA + 1) * 2 * L * d * d + f_embed)


def skipped_flops(d, NL, NA, L, B, text_lens, text_len=512, text_dim=4096):
    """FLOPs of the algorithmic count (step_flops: the work as the reference performs it) that the engine does NOT execute per
    step because they are output-neutral: (1) step-invariant hoists (text embedding, control-map patch embedding, the
    cross-attention K / V projections: SURVEY 8d), (2) the identical zero-padded prompt keys of cross-attention folded into one
    key (64-key tiles), (3) the prompt-independent half of block 0 of both chains computed once for the CFG pair."""
    hoisted = B * ((NL + NA) * 4 * text_len * d * d + 2 * L * 512 * d + 2 * text_len * text_dim * d + 2 * text_len * d * d)
    folded = 0
    for n in text_lens:
        lk_eff = min(text_len, -(-(n + 1) // 64) * 64) if n < text_len - 1 else text_len
        folded += (NL + NA) * 4 * L * (text_len - lk_eff) * d
    shared = (B - 1) * 2 * (8 * L * d * d + 4 * L * L * d) if B >= 2 else 0
    return hoisted + folded + shared


def cpu_baseline(mk, f_step, L):
    """CPU oracle (oracle/wan_oracle.py, fp32 PyTorch: the "port") timed on this box's host cores on a BOUNDED sample of the
    same workload and extrapolated to steps/s by algorithmic FLOPs.  SURVEY 8d's protocol names one main block + one adapter
    block at full shape; at full shape one block is 42 TFLOP (minutes of CPU time), so the sample keeps the block pair and the
    model width but cuts the token axis: one main block, then one adapter block with its after_proj (VC.py:112-125), B = 1,
    1536 tokens.  Self-attention is 52 % of the cfg-3 step's FLOPs and only ~6 % of this sample's, and CPU attention runs below
    CPU GEMM speed, so the extrapolated rate is OPTIMISTIC for the CPU (stated in `sample`)."""
    import torch
    from oracle import wan_oracle as O
    d, ffn, heads = mk["dim"], mk["ffn_dim"], mk["num_heads"]
    text_len = mk.get("text_len", 512)
    Ls = 1536 if d >= 1024 else 256
    cfg = O.Config(dim=d, ffn_dim=ffn, num_heads=heads, num_layers=1, text_len=text_len, text_dim=mk.get("text_dim", 4096))
    g = torch.Generator().manual_seed(0)
    W = {}
    for k, shp in O.state_dict_shapes(cfg).items():
        if k.startswith(("blocks.0.", "geoada_blocks.0.")):
            W[k] = torch.randn(shp, generator=g) * (0.02 if len(shp) > 1 else 1.0)
    grid = (3, 16, Ls // 48) if Ls == 1536 else (1, 16, 16)
    x = torch.randn(1, Ls, d, generator=g)
    e0 = torch.randn(1, 6, d, generator=g) * 0.1
    ctx = torch.randn(1, text_len, d, generator=g)
    freqs = O.rope_table(128)
    f_blk = (8 * Ls * d * d + 4 * Ls * Ls * d + 4 * Ls * d * d + 4 * text_len * d * d + 4 * Ls * text_len * d + 4 * Ls * d * ffn)
    flops = 2 * f_blk + 2 * Ls * d * d                       # main block + adapter block + after_proj
    attn_share = 2 * 4 * Ls * Ls * d / flops
    nblk = mk["num_layers"] + (mk["num_layers"] + 1) // 2
    real_share = 2 * nblk * 4 * L * L * d / f_step

    def pair():
        y = O.attention_block(W, "blocks.0.", x, e0, [Ls], [grid], freqs, ctx, heads)
        c = O.attention_block(W, "geoada_blocks.0.", x, e0, [Ls], [grid], freqs, ctx, heads)
        return y, O.linear(c, W["geoada_blocks.0.after_proj.weight"], W["geoada_blocks.0.after_proj.bias"])
    times = []
    t_end = time.time() + 25.0
    with torch.no_grad():
        for it in range(4):
            t0 = time.time()
            pair()
            times.append(time.time() - t0)
            if time.time() > t_end:
                break
    best = min(times[1:]) if len(times) > 1 else times[0]
    rate = flops / best
    return {"value": rate / f_step, "unit": "denoise-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fp32: 1 main block + 1 adapter block (+ after_proj), B=1, {Ls} tokens, d={d}: {best:.2f} s = "
                      f"{rate / 1e12:.3f} TFLOP/s, extrapolated by FLOPs to a full step ({f_step / 1e15:.3f} PFLOP); "
                      f"self-attention is {attn_share:.0%} of the sample's FLOPs vs {real_share:.0%} of the real step's, so this rate is "
                      f"optimistic for the CPU",
            "host_cpus": os.cpu_count()}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="wan14b-81f-480x832", choices=sorted(WORKLOADS))
    ap.add_argument("--num_inference_steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--cfg-degree", type=int, default=int(os.environ.get("VC_BENCH_CFG_DEGREE", "0")), choices=(0, 1, 2),
                    help="ranks that split the CFG pair (0 = auto: 2 when --gpus 2, else 1)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl (= RCCL; product: the engine's own communicators over xGMI) or gloo: a REHEARSAL of the N > 1 "
                         "launch on a box with fewer GPUs than ranks -- ranks share devices round-robin, exchange buffers are "
                         "staged through host memory; the rate it prints is not a result")
    ap.add_argument("--teacache-steps", type=int, default=0,
                    help="also time N sampler steps from step 0 with TeaCache on (CLI defaults: threshold 0.10, skip-start 5) "
                         "and report them as a separate 'teacache_on' object; 0 = off (the headline metric is TeaCache-off)")
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it (no WORLD_SIZE in the environment): start N fresh rank
    processes -- one per GPU, torchrun's environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), as the reference's
    `torchrun --nproc-per-node=N` of inference.sh:62-71 -- wait for them and forward rank 0's single JSON line.
    This parent never touches the GPU (it does not even import torch); a failed rank ends the run with its exit code."""
    import subprocess
    import tempfile
    n = args.gpus
    port = int(os.environ.get("MASTER_PORT") or _free_port())
    procs, out0 = [], tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL's intra-node transport needs it on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    rc, deadline = 0, None
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                print(f"bench.py: rank {r} exited with code {c}; stopping the other ranks", file=sys.stderr)
                deadline = time.time() + 20.0                     # peers blocked in a collective never return by themselves
        if deadline is not None and time.time() > deadline:
            for r in live:
                procs[r].kill()                                   # exactly the children started above
            deadline = None
        time.sleep(0.05)
    out0.seek(0)
    lines = [ln for ln in out0.read().splitlines() if ln.strip()]
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: rank 0 printed {len(lines)} lines, expected exactly one", file=sys.stderr)
        rc = 1
    if rc == 0:
        sys.stdout.write(lines[0] + "\n")
        sys.stdout.flush()
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    run_rank(args)


def run_rank(args):
    import torch
    import torch.distributed as dist

    # Only the JSON line may reach stdout: RCCL prints a version banner on stdout when the first communicator is
    # created, other libraries may chatter too.  Route fd 1 to stderr for the whole run and keep the real stdout aside.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local = local % max(1, torch.cuda.device_count())          # ranks share devices (device_count does not initialise HIP)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("VC_BENCH_FORCE_DIST") == "1"   # the latter: rehearse the N>1 plumbing at N=1
    if use_dist:
        # host-side group (gloo): rendezvous, barrier, max-over-ranks of the time, shipping the ncclUniqueIds.  The data
        # path's RCCL communicators belong to the engine (vc_sp_init_rccl); torch's own "nccl" backend is registered for
        # CUDA tensors so that SequenceParallel recognises an RCCL-capable world, but no torch collective runs on it.
        dist.init_process_group("gloo" if rehearsal else "cpu:gloo,cuda:nccl", rank=rank, world_size=world)

    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    from versecrafter_amd.pipeline import WanVerseCrafterPipeline
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler

    mk, frames, height, width = WORKLOADS[args.workload]
    mk = dict(mk)
    T, h, w = (frames - 1) // 4 + 1, height // 8, width // 8
    L = T * (h // 2) * (w // 2)
    text_dim = mk.get("text_dim", 4096)

    # ---- model: random weights of the named architecture, identical on every rank (seed 0) ----
    torch.manual_seed(0)
    model = VerseCrafterWanTransformer3DModel(geoada_in_dim=128, param_device=dev, param_dtype=torch.bfloat16, **mk)
    model.init_weights(zero_init_outputs=False)
    # N = 2: the two samples of the CFG pair are independent units -- one per rank, no data-path collective, only the noise
    # prediction is all-gathered.  N >= 4: Ulysses over all ranks.  --cfg-degree overrides.
    cfg_degree = args.cfg_degree if args.cfg_degree else (2 if world == 2 else 1)
    if world % cfg_degree:
        raise SystemExit(f"--cfg-degree {cfg_degree} does not divide --gpus {world}")
    sp_degree = world // cfg_degree
    if use_dist:
        from versecrafter_amd import dist as vdist
        sp_group, bp_group = vdist.make_groups(sp_degree, cfg_degree)
        if sp_degree == 1 and cfg_degree > 1:
            model.enable_multi_gpus_inference()                       # batch-parallel only
        else:
            model.enable_multi_gpus_inference(vdist.SequenceParallel(sp_group, force_exchange=(world == 1)))
    scheduler = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    pipe = WanVerseCrafterPipeline(transformer=model, scheduler=scheduler)
    pipe._guidance_scale = 5.0

    # ---- synthetic inputs, seed 2025, resident in HBM ----
    g = torch.Generator(device="cpu").manual_seed(2025)
    latents = torch.randn(1, 16, T, h, w, generator=g).to(dev, torch.bfloat16)
    ctrl = torch.randn(1, 64, T, h, w, generator=g)
    mask = (torch.rand(1, 64, T, h, w, generator=g) < 0.5).float()
    mask[:, :, 0] = 0
    geoada = torch.cat([ctrl, mask], 1).to(dev, torch.bfloat16)
    geoada_in = torch.cat([geoada, geoada], 0).contiguous()
    n_un, n_co = (60, 77) if mk.get("text_len", 512) >= 77 else (20, 33)
    embeds = [torch.randn(n_un, text_dim, generator=g).to(dev, torch.bfloat16),    # uncond
              torch.randn(n_co, text_dim, generator=g).to(dev, torch.bfloat16)]    # cond
    scheduler.set_timesteps(args.num_inference_steps, device=dev, shift=16)
    ts = scheduler.timesteps
    seq_len = L
    pipe._cfg_pair_maps = geoada_in            # as __call__ does: the control maps of the CFG pair are one tensor stacked twice

    def run_steps(lat, first, n):
        for i in range(first, first + n):
            lat = pipe.denoise_step(i, ts[i], lat, embeds, geoada_in, seq_len, True, 1.0)
        return lat

    def barrier():
        torch.cuda.synchronize()                                   # this rank's queue (all engine streams) has drained ...
        if use_dist:
            dist.all_reduce(torch.zeros(1))                        # ... and so has every other rank's (host-side, gloo)

    lat = run_steps(latents, 0, args.warmup)
    barrier()
    if not args.no_profile:
        model.profile_enable(True)
    t0 = time.perf_counter()
    lat = run_steps(lat, args.warmup, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = None
    if not args.no_profile:
        prof = model.profile_read()
        model.profile_enable(False)
    finite = bool(torch.isfinite(lat.float()).all().item())
    tea = None
    if args.teacache_steps > 0:
        # TeaCache-on is a separate line.  With random weights the gate's statistics are not those of the
        # released checkpoint -- the count of skipped steps is reported next to the rate.
        coeff_14b = [8.10705460e+03, 2.13393892e+03, -3.72934672e+02, 1.66203073e+01, -4.17769401e-02]   # CLI.py:305-313
        model.enable_teacache(coeff_14b, args.num_inference_steps, 0.10, num_skip_start_steps=5, offload=False)
        scheduler.set_timesteps(args.num_inference_steps, device=dev, shift=16)
        n = min(args.teacache_steps, args.num_inference_steps)
        skipped = 0
        barrier()
        t1 = time.perf_counter()
        lt = latents
        for i in range(n):
            lt = pipe.denoise_step(i, ts[i], lt, embeds, geoada_in, seq_len, True, 1.0)
            skipped += 0 if model.should_calc else 1
        barrier()
        el = time.perf_counter() - t1
        model.disable_teacache()
        tea = {"value": n / el, "unit": "denoise-steps/s", "steps": n, "skipped_steps": skipped, "threshold": 0.10,
               "num_skip_start_steps": 5, "note": "random weights: gate statistics differ from the released checkpoint"}
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    rccl_ranks = model.sp_comm_ranks()
    if cfg_degree > 1 and not rehearsal:
        rccl_ranks = max(rccl_ranks, 1) * cfg_degree      # + the batch-parallel gather on torch's nccl (= RCCL) groups

    if rank == 0:
        NL, NA = mk["num_layers"], (mk["num_layers"] + 1) // 2
        f_step = step_flops(mk["dim"], mk["ffn_dim"], NL, NA, L, 2, mk.get("text_len", 512), text_dim)
        sps = args.steps / elapsed
        out = {
            "metric": "denoise-steps/sec Wan2.1-14B+GeoAdapter 81fx480p" if args.workload == "wan14b-81f-480x832"
                      else f"denoise-steps/sec {args.workload}",
            "value": sps, "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random weights seed 0, inputs seed 2025)",
            "config": {"workload": args.workload, "latent": [16, T, h, w], "tokens": L, "global_batch": 2,
                       "cfg": "batched pair", "guidance_scale": 5.0, "sampler": "UniPC shift 16",
                       "teacache": "off",
                       "parallelism": f"ulysses-sp{world}" if cfg_degree == 1 else f"cfg{cfg_degree} x ulysses-sp{sp_degree}",
                       "pflop_per_step": f_step / 1e15},
            "step_mfma_frac": f_step * sps / (world * PEAK_BF16_TFLOPS * 1e12),
            # the same with the output-neutral work the engine skips taken out of the numerator (see skipped_flops)
            "step_mfma_frac_executed": (f_step - skipped_flops(mk["dim"], NL, NA, L, 2, (n_un, n_co), mk.get("text_len", 512),
                                                               text_dim)) * sps / (world * PEAK_BF16_TFLOPS * 1e12),
            "outputs_finite": finite,
            # world size the engine's own RCCL communicator reports after init (ncclCommCount); 0 = no RCCL exchange ran
            "rccl_ranks": rccl_ranks,
            "transport": ("none (single rank)" if not use_dist else
                          "gloo + host-staged buffers (REHEARSAL of the launch, not a result)" if rehearsal else
                          "one CFG sample per rank, no data-path collective; the noise prediction is all-gathered over "
                          "torch.distributed's RCCL group" if sp_degree == 1 else
                          "engine-owned RCCL communicators, one per stream lane" if rccl_ranks else
                          "torch.distributed RCCL process groups, one per stream lane (engine-owned communicators unavailable)"),
        }
        if prof is not None:
            bd = {}
            for k, v in prof.items():
                if v["launches"]:
                    sec = v["ms"] / 1e3
                    bd[k] = {"launches_per_step": v["launches"] / args.steps, "ms_per_step": v["ms"] / args.steps,
                             "avg_ms": v["ms"] / v["launches"], "tflops": v["flops"] / sec / 1e12 if v["flops"] else None,
                             "gbps": v["bytes"] / sec / 1e9}
            dom = max(("attn_self", "gemm"), key=lambda k: prof[k]["ms"])
            v = prof[dom]
            ach = v["flops"] / (v["ms"] / 1e3) / 1e12
            kname = {"attn_self": "attn_fwd_pipe_kernel", "gemm": "gemm_pp_kernel"}[dom]
            # HBM bytes per launch cannot be counted inside this process: they come from the committed rocprofv3 --pmc passes
            # of this same command (profiles/traffic.json names the summary file) -- a recorded constant, labelled as such
            traffic, traffic_src = None, None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                if tj.get("workload") == args.workload and tj.get("n_gpus") == world and kname in tj:
                    traffic = tj[kname]["bytes_per_launch"]
                    traffic_src = tj.get("source")
            except (OSError, ValueError):
                pass
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                               "traffic_source": (f"recorded, not live: {traffic_src}" if traffic is not None else None),
                               "avg_launch_ms": v["ms"] / v["launches"], "launches": v["launches"]}
            if world > 1:
                out["roofline"]["note"] = ("sequence-parallel run: the two samples of the CFG pair are scheduled on separate "
                                           "streams / interleaved with exchanges, so kernel launches can overlap and the per-launch "
                                           "durations (hence 'achieved') are lower bounds; the N=1 line carries the kernel roofline")
            out["breakdown"] = bd
        if tea is not None:
            out["teacache_on"] = tea
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(mk, f_step, L)
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if use_dist:
        dist.all_reduce(torch.zeros(1))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

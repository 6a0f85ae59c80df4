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
                 sample (one main + one adapter block at the 14B width on the 8190 rows one rank of an 8-way run holds)
                 and extrapolated by FLOPs;
  teacache_on  : what a TeaCache-skipped step and a calc+store step cost on this GPU, and the steps/s that implies for the
                 CLI's skip window as a function of the number of skipped steps (the skip RATE needs the released weights).

Multi-rank start (N > 1): every rank runs in a CHILD process of a supervisor that never touches the GPU -- `python bench.py
--gpus N` supervises all N children itself; under `python -m torch.distributed.run ... bench.py --gpus N` every torchrun worker
supervises its own rank and the supervisors talk through torchrun's store.  A child marks "started" and "up" (communicators
created and one probe exchange done); when the marks do not arrive in time, or the run exceeds its budget, the supervisors
kill exactly their children and start a FRESH set with VC_SP_TRANSPORT=torch (torch.distributed's RCCL groups); if that
fails too they exit non-zero with the tail of every rank's stderr file (bench_n<N>.rank<r>.err).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP8_TFLOPS = 5000.0     # dense fp8 MFMA peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF dense)

WORKLOADS = {
    # name: (model kwargs, frames, height, width)
    "wan14b-81f-480x832": (dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40), 81, 480, 832),
    "wan14b-49f-480x832": (dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40), 49, 480, 832),
    "wan14b-81f-720x1280": (dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40), 81, 720, 1280),   # BASELINE config 4
    "wan1.3b-9f-320x512": (dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30), 9, 320, 512),
    "tiny": (dict(dim=256, ffn_dim=512, num_heads=2, num_layers=4, text_dim=64, text_len=48), 9, 64, 96),
    # four heads: on 4 ranks all THREE layouts of an N >= 4 run exist (Ulysses-4, CFG pair x Ulysses-2, Ulysses 2 x ring 2): launch rehearsals
    "tiny4h": (dict(dim=512, ffn_dim=1024, num_heads=4, num_layers=4, text_dim=64, text_len=48), 9, 64, 96),
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
    model width and takes the rows ONE RANK of an 8-way sequence-parallel run holds: L/8 tokens of both CFG samples = 8190
    rows as one sequence (B = 1): 13 TFLOP, self-attention 21 % of it (52 % of the real step; the 1536-token slice of
    earlier rounds had 4 %).  One timed pass after a small warm-up pass (thread pool, allocator); still optimistic for the
    CPU, whose attention runs below its GEMM rate, and labelled so."""
    import torch
    from oracle import wan_oracle as O
    d, ffn, heads = mk["dim"], mk["ffn_dim"], mk["num_heads"]
    text_len = mk.get("text_len", 512)
    big = d >= 1024
    Ls = int(os.environ.get("VC_BENCH_CPU_TOKENS", "0")) or (2 * -(-L // 8) if big else 256)
    grid = (1, 1, Ls)                                        # RoPE positions along w (w < 1024 rows of the table)
    if Ls > 1024:
        w = max(x for x in range(1, 1025) if Ls % x == 0)
        rest = Ls // w
        hh = max(x for x in range(1, 1025) if rest % x == 0)
        grid = (rest // hh, hh, w)
        if grid[0] > 1024:
            raise SystemExit(f"cpu_baseline: {Ls} tokens do not factor into a RoPE grid")
    cfg = O.Config(dim=d, ffn_dim=ffn, num_heads=heads, num_layers=1, text_len=text_len, text_dim=mk.get("text_dim", 4096))
    g = torch.Generator().manual_seed(0)
    W = {}
    for k, shp in O.state_dict_shapes(cfg).items():
        if k.startswith(("blocks.0.", "geoada_blocks.0.")):
            W[k] = torch.randn(shp, generator=g) * (0.02 if len(shp) > 1 else 1.0)
    freqs = O.rope_table(128)
    nblk = mk["num_layers"] + (mk["num_layers"] + 1) // 2
    real_share = 2 * nblk * 4 * L * L * d / f_step

    def pair(n, gr):
        x = torch.randn(1, n, d, generator=g)
        e0 = torch.randn(1, 6, d, generator=g) * 0.1
        ctx = torch.randn(1, text_len, d, generator=g)
        t0 = time.time()
        y = O.attention_block(W, "blocks.0.", x, e0, [n], [gr], freqs, ctx, heads)
        c = O.attention_block(W, "geoada_blocks.0.", x, e0, [n], [gr], freqs, ctx, heads)
        z = O.linear(c, W["geoada_blocks.0.after_proj.weight"], W["geoada_blocks.0.after_proj.bias"])
        return time.time() - t0, float(y.abs().mean() + z.abs().mean())

    with torch.no_grad():
        pair(128, (1, 8, 16))                                # warm-up
        best, _ = pair(Ls, grid)
    f_blk = (8 * Ls * d * d + 4 * Ls * Ls * d + 4 * Ls * d * d + 4 * text_len * d * d + 4 * Ls * text_len * d + 4 * Ls * d * ffn)
    flops = 2 * f_blk + 2 * Ls * d * d                       # main block + adapter block + after_proj
    attn_share = 2 * 4 * Ls * Ls * d / flops
    rate = flops / best
    return {"value": rate / f_step, "unit": "denoise-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fp32: 1 main block + 1 adapter block (+ after_proj), B=1, {Ls} tokens (the rows of one rank of an "
                      f"8-way run), d={d}: {best:.2f} s for {flops / 1e12:.2f} TFLOP = {rate / 1e12:.3f} TFLOP/s, extrapolated by "
                      f"FLOPs to a full step ({f_step / 1e15:.3f} PFLOP); self-attention is {attn_share:.0%} of the sample's FLOPs vs "
                      f"{real_share:.0%} of the real step's, so this rate is still optimistic for the CPU",
            "host_cpus": os.cpu_count()}


def dtype_label(args):
    """`dtype` of the JSON line: the arithmetic type(s) the step computes in.  Anything but "bf16" is an opt-in mode, never the headline."""
    lin, att = args.fp8_linear, args.fp8_attn >= 0
    if not lin and not att:
        return "bf16"
    parts = ["fp8 e4m3 linear layers (per-token x per-channel scales, fp32 accumulate)" if lin else "bf16 linear layers",
             (f"fp8 e4m3 self-attention (MX block scales, pmode {args.fp8_attn})" if att else "bf16 attention")]
    return " + ".join(parts) + ": NOT the bf16 headline"


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
    ap.add_argument("--fp8-attn", type=int, default=-1, choices=(-1, 0, 1),
                    help="run the blocks' SELF-attention in fp8 (q, k, v and the softmax weights e4m3 under MX-style block scales; NOT the "
                         "headline): 1 = the weights' bytes from the piecewise-linear 2^x, 0 = v_exp_f32; -1 (default) = bf16 attention")
    ap.add_argument("--fp8-linear", action="store_true",
                    help="run the blocks' nn.Linear layers in fp8 (BASELINE config 5's dtype; NOT the headline: the reference computes in "
                         "bf16) -- the line then says dtype 'fp8 e4m3 linear layers (fp32 accumulate) + bf16 attention'")
    ap.add_argument("--single-layout", action="store_true", help="N >= 4: time Ulysses-N only (no alternative layouts in the same run)")
    ap.add_argument("--ring-degree", type=int, default=1,
                    help="ring degree R of the sequence-parallel group (Ulysses x ring hybrid, the reference's --ring_degree): the Ulysses "
                         "degree becomes ranks / R.  Default 1: pure Ulysses -- what the 14B model's 40 heads allow on 1 / 2 / 4 / 8 GPUs")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl (= RCCL; product: the engine's own communicators over xGMI) or gloo: a REHEARSAL of the N > 1 "
                         "launch on a box with fewer GPUs than ranks -- ranks share devices round-robin, exchange buffers are "
                         "staged through host memory; the rate it prints is not a result")
    ap.add_argument("--no-teacache-line", action="store_true",
                    help="skip the separate TeaCache-on measurement (forced skipped / calc + store steps, N = 1 only)")
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ----------------------------------------------------------------------------------------------------------------------
# Supervised multi-rank start.  Nothing in this section imports torch in the self-launch mode, and nothing in it ever touches
# the GPU in either mode: GPU work happens in the children only, and a child is never re-executed -- a failed set is killed
# (exactly the PIDs started here) and a FRESH set is started.
# ----------------------------------------------------------------------------------------------------------------------
class _LocalBoard:
    """The supervisors' shared flags when ONE supervisor owns every rank (python bench.py --gpus N)."""

    def __init__(self):
        self.kv, self.cnt = {}, {}

    def set(self, k, v):
        self.kv[k] = v

    def get(self, k, timeout=0.0):
        return self.kv[k]

    def check(self, k):
        return k in self.kv

    def add(self, k, n):
        self.cnt[k] = self.cnt.get(k, 0) + n
        return self.cnt[k]


class _StoreBoard:
    """The same over torchrun's rendezvous store (one supervisor per rank; host sockets only -- no device, no process group)."""

    def __init__(self):
        from datetime import timedelta
        from torch.distributed import PrefixStore, rendezvous
        store, self.rank, self.world = next(iter(rendezvous("env://", timeout=timedelta(seconds=300))))
        self.store = PrefixStore("vc_bench_supervisor", store)
        self._td = timedelta

    def set(self, k, v):
        self.store.set(k, v)

    def get(self, k, timeout=60.0):
        self.store.wait([k], self._td(seconds=timeout))
        return self.store.get(k).decode()

    def check(self, k):
        return bool(self.store.check([k]))

    def add(self, k, n):
        return int(self.store.add(k, n))


def _tail(path, n=15):
    try:
        with open(path, errors="replace") as f:
            return "".join(f.readlines()[-n:])
    except OSError:
        return ""


def launch_ranks(args):
    """Start and supervise the rank processes (see the module docstring).  Returns the exit code.

    Modes: no WORLD_SIZE in the environment -- this process owns all N children (torchrun's environment contract: RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_*, as the reference's `torchrun --nproc-per-node=N` of inference.sh:62-71); WORLD_SIZE set
    (a torchrun worker) -- it owns the child of its own rank and agrees with the other supervisors through the store.

    Per attempt: children must mark "started" (VC_BENCH_START_TIMEOUT, default 300 s: a cold `import torch` takes minutes)
    and then "up" (VC_BENCH_UP_TIMEOUT, default 90 s after the last "started" -- 150 s on the first attempt of an N >= 4 run, which
    brings three layouts' communicators up: communicators created, probe exchange done);
    the whole run has VC_BENCH_BUDGET seconds (default 560: below the driver's limit).  A stall, or a rank that dies between
    "started" and "up", is a transport failure: attempt 2 runs fresh children with VC_SP_TRANSPORT=torch.  Any other failure,
    or a failed attempt 2, ends the run non-zero with every rank's stderr tail."""
    import shutil
    import subprocess
    import tempfile
    n = args.gpus
    t_begin = time.time()
    budget = float(os.environ.get("VC_BENCH_BUDGET", "560"))
    t_start_lim = float(os.environ.get("VC_BENCH_START_TIMEOUT", "300"))
    t_up_lim = float(os.environ.get("VC_BENCH_UP_TIMEOUT", "90"))
    # the first attempt of an N >= 4 run brings the communicators of three layouts up (and probes each): 30 s more per extra layout
    n_layouts0 = 3 if (n >= 4 and n % 2 == 0 and not args.single_layout and not args.cfg_degree and args.ring_degree <= 1) else 1
    t_up_extra0 = 30.0 * (n_layouts0 - 1) if "VC_BENCH_UP_TIMEOUT" not in os.environ else 0.0
    grace = float(os.environ.get("VC_BENCH_KILL_GRACE", "10"))
    log_dir = os.environ.get("VC_BENCH_LOG_DIR") or os.getcwd()
    child_cmd = os.environ.get("VC_BENCH_TEST_CHILD")              # tests: a stand-in rank program
    child_argv = [sys.executable, child_cmd] if child_cmd else [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    under_launcher = "WORLD_SIZE" in os.environ
    if under_launcher:
        if int(os.environ["WORLD_SIZE"]) != n:
            print(f"bench.py: --gpus {n} but WORLD_SIZE={os.environ['WORLD_SIZE']}", file=sys.stderr)
            return 2
        board = _StoreBoard()
        mine = {board.rank: int(os.environ.get("LOCAL_RANK", board.rank))}
        me = board.rank
    else:
        board = _LocalBoard()
        mine = {r: r for r in range(n)}
        me = 0
    status_dir = tempfile.mkdtemp(prefix="vc_bench_")
    err_paths = {r: os.path.join(log_dir, f"bench_n{n}.rank{r}.err") for r in mine}
    for pth in err_paths.values():
        open(pth, "w").close()
    rc_final, line = 1, None
    attempts = [None, "torch"] if os.environ.get("VC_SP_TRANSPORT") is None else [os.environ["VC_SP_TRANSPORT"]]
    started = []                     # every child this supervisor ever started: none may outlive it, whatever goes wrong here
    try:
        for a, transport in enumerate(attempts):
            # ---- a fresh rendezvous port for the children, agreed through the board
            if me == 0:
                board.set(f"port{a}", str(_free_port()))
            port = int(board.get(f"port{a}", timeout=120.0))
            procs, outs = {}, {}
            for r, local in mine.items():
                env = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_ASYNC"))}
                env.update(RANK=str(r), LOCAL_RANK=str(local), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VC_BENCH_CHILD="1",
                           VC_BENCH_STATUS_DIR=status_dir, VC_BENCH_ATTEMPT=str(a))
                # dmabuf IPC (versecrafter_amd.dist.ensure_ipc_env: the hosts of this pool refuse the legacy IPC mode)
                env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                env.setdefault("NCCL_DEBUG", "WARN")                   # RCCL's own diagnostics into the per-rank stderr file (fd 1 is routed there)
                if transport is not None:
                    env["VC_SP_TRANSPORT"] = transport
                with open(err_paths[r], "a") as ef:
                    ef.write(f"==== attempt {a} (transport {transport or 'default: engine-owned RCCL'}) rank {r} port {port}\n")
                outs[r] = open(os.path.join(status_dir, f"rank{r}.a{a}.out"), "w+")
                procs[r] = subprocess.Popen(child_argv, env=env, stdout=outs[r], stderr=open(err_paths[r], "a"))
                started.append(procs[r])
            t_spawn = time.time()
            t_all_started = None
            marked = {r: set() for r in mine}
            live, why, kind = set(mine), None, None
            fail_seen_at = None
            while True:
                now = time.time()
                exited = {r: procs[r].poll() for r in sorted(live)}
                for r in mine:                                     # marks first: they classify an exit seen in the same pass
                    for m in ("started", "up"):
                        if m not in marked[r] and os.path.exists(os.path.join(status_dir, f"rank{r}.a{a}.{m}")):
                            marked[r].add(m)
                            board.add(f"{m}{a}", 1)
                for r, c in exited.items():
                    if c is None:
                        continue
                    live.discard(r)
                    if c == 0:
                        board.add(f"done{a}", 1)
                    elif not board.check(f"fail{a}"):
                        in_bringup = "started" in marked[r] and "up" not in marked[r]
                        board.set(f"fail{a}", f"{'transport' if in_bringup else 'rank'}: rank {r} exited with code {c}")
                n_started, n_up = board.add(f"started{a}", 0), board.add(f"up{a}", 0)
                if n_started >= n and t_all_started is None:
                    t_all_started = now
                if board.check(f"fail{a}"):
                    if fail_seen_at is None:
                        fail_seen_at = now
                        why = board.get(f"fail{a}")
                        kind = why.split(":", 1)[0]
                        print(f"bench.py: attempt {a}: {why}; stopping this set of ranks", file=sys.stderr)
                    if not live or now - fail_seen_at > grace:     # peers blocked in a collective never return by themselves
                        break
                elif board.add(f"done{a}", 0) >= n:
                    break
                elif n_started < n and now - t_spawn > t_start_lim:
                    board.set(f"fail{a}", f"stall: {n_started} of {n} ranks started within {t_start_lim:.0f} s")
                elif n_started >= n and n_up < n and now - t_all_started > t_up_lim + (t_up_extra0 if a == 0 else 0.0):
                    board.set(f"fail{a}", f"stall: {n_up} of {n} ranks brought their communicators up within "
                                          f"{t_up_lim + (t_up_extra0 if a == 0 else 0.0):.0f} s of starting")
                elif now - t_begin > budget:
                    board.set(f"fail{a}", f"stall: over the run budget of {budget:.0f} s")
                time.sleep(0.1)
            for r in live:                                         # exactly the children started above
                procs[r].kill()
            for r in mine:
                procs[r].wait()
            # nobody starts the next set before every supervisor has ended this one
            board.add(f"end{a}", len(mine))
            t_w = time.time()
            while board.add(f"end{a}", 0) < n and time.time() - t_w < 60:
                time.sleep(0.1)
            if why is None:
                rc_final = 0
                if 0 in mine:
                    outs[0].seek(0)
                    lines = [ln for ln in outs[0].read().splitlines() if ln.strip()]
                    if len(lines) != 1:
                        print(f"bench.py: rank 0 printed {len(lines)} lines, expected exactly one", file=sys.stderr)
                        rc_final = 1
                    else:
                        line = lines[0]
                break
            # the first layout of this attempt was measured and saved before a LATER (alternative) layout took the run down: that line
            # is the result; what happened afterwards travels with it
            part_path = os.path.join(status_dir, f"rank0.a{a}.partial")
            if 0 in mine and os.path.exists(part_path):
                try:
                    saved = json.loads(open(part_path).read())
                    saved["alt_error"] = why
                    line, rc_final = json.dumps(saved), 0
                    print(f"bench.py: attempt {a}: {why} -- AFTER the first layout had been measured: reporting that line", file=sys.stderr)
                except (OSError, ValueError):
                    pass
            if under_launcher:
                if 0 in mine:
                    board.set(f"partial{a}", "1" if rc_final == 0 else "0")
                try:
                    if board.get(f"partial{a}", timeout=30.0) == "1":
                        rc_final = 0
                except Exception:      # noqa: BLE001 -- rank 0's supervisor is gone: nothing to report
                    pass
            if rc_final == 0:
                break
            retry = kind in ("stall", "transport") and a + 1 < len(attempts) and time.time() - t_begin < budget
            if not retry:
                break
            print(f"bench.py: starting a fresh set of ranks with VC_SP_TRANSPORT={attempts[a + 1]}", file=sys.stderr)
        if rc_final != 0:
            for r in sorted(mine):
                print(f"---- rank {r} stderr tail ({err_paths[r]}) ----\n{_tail(err_paths[r])}", file=sys.stderr)
    finally:
        for pr in started:           # a supervisor that dies (store gone, interrupt) takes exactly its own children with it
            if pr.poll() is None:
                pr.kill()
        shutil.rmtree(status_dir, ignore_errors=True)
    if rc_final == 0 and line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    return rc_final


def _mark(name):
    """Child side of the supervisor's protocol: an empty marker file (no-op without a supervisor)."""
    d = os.environ.get("VC_BENCH_STATUS_DIR")
    if d:
        open(os.path.join(d, f"rank{os.environ.get('RANK', '0')}.a{os.environ.get('VC_BENCH_ATTEMPT', '0')}.{name}"), "w").close()


def main():
    args = parse_args()
    if args.gpus > 1 and os.environ.get("VC_BENCH_CHILD") != "1":
        sys.exit(launch_ranks(args))
    run_rank(args)


T_RANK_START = time.time()


def run_rank(args):
    # dmabuf IPC before the HIP runtime starts (versecrafter_amd.dist.ensure_ipc_env says why)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    _mark("started")
    use_dist = world > 1 or os.environ.get("VC_BENCH_FORCE_DIST") == "1"   # the latter: rehearse the N>1 plumbing at N=1
    if use_dist:
        # host-side group (gloo): rendezvous, barrier, max-over-ranks of the time, shipping the ncclUniqueIds.  The data
        # path's RCCL communicators belong to the engine (vc_sp_init_rccl); torch's own "nccl" backend is registered for
        # CUDA tensors (the CFG split's gather of the noise prediction, and the torch-transport fallback, run on it).
        dist.init_process_group("gloo" if rehearsal else "cpu:gloo,cuda:nccl", rank=rank, world_size=world)

    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    from versecrafter_amd.pipeline import WanVerseCrafterPipeline
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler

    mk, frames, height, width = WORKLOADS[args.workload]
    mk = dict(mk)
    T, h, w = (frames - 1) // 4 + 1, height // 8, width // 8
    L = T * (h // 2) * (w // 2)
    text_dim = mk.get("text_dim", 4096)
    NL, NA = mk["num_layers"], (mk["num_layers"] + 1) // 2
    f_step = step_flops(mk["dim"], mk["ffn_dim"], NL, NA, L, 2, mk.get("text_len", 512), text_dim)

    torch.manual_seed(0)
    model = VerseCrafterWanTransformer3DModel(geoada_in_dim=128, param_device=dev, param_dtype=torch.bfloat16, **mk)

    # ---- layouts: (cfg_degree, sp_degree).  N = 2 measures BOTH ways of using two GPUs in this one run -- Ulysses over the two
    # ranks (north_star's curve) and one CFG sample per rank (no data-path collective: DESIGN.md 6); the faster one is the
    # headline, the other goes under "alt".  N >= 4: Ulysses over all ranks.  --cfg-degree pins one layout.
    # A layout is (cfg_degree, sp_degree, ring): ring 0 = the bench's own choice (--ring-degree, else pure Ulysses where the heads allow).
    # N >= 4 (round 4): north_star's Ulysses-N first -- its line is saved as soon as it is measured --, then, time permitting, two
    # alternatives of the SAME run: the CFG pair on two Ulysses groups of N/2 (half the all-to-all bytes and peers per exchange) and the
    # reference's documented launch line, Ulysses 2 x ring N/2 (inference.sh:62-71: 2 x 4 on 8 GPUs).  A second attempt of the supervised
    # launch (fallback transport) and --single-layout measure the first layout only.
    attempt = int(os.environ.get("VC_BENCH_ATTEMPT", "0"))
    if args.cfg_degree:
        if world % args.cfg_degree:
            raise SystemExit(f"--cfg-degree {args.cfg_degree} does not divide --gpus {world}")
        layouts = [(args.cfg_degree, world // args.cfg_degree, 0)]
    elif world == 2:
        layouts = [(1, 2, 0), (2, 1, 0)]
    elif world >= 4 and world % 2 == 0 and attempt == 0 and not args.single_layout and args.ring_degree <= 1:
        layouts = [(1, world, 0), (2, world // 2, 0)]
        if mk["num_heads"] % 2 == 0 and mk["num_heads"] % world == 0 and world // 2 <= 8:
            layouts.append((1, world, world // 2))
    else:
        layouts = [(1, world, 0)]
    groups, sps, observed = {}, {}, {}
    vdist = None
    if use_dist:
        from versecrafter_amd import dist as vdist
        for lay in layouts:                                            # process groups of every layout, created once
            groups[lay] = vdist.make_groups(lay[1], lay[0])

    def ring_of(lay):
        """--ring-degree R > 1 forces the Ulysses x ring hybrid; otherwise pure Ulysses wherever the head count divides by the group
        and the smallest ring that fits where it does not (1.3B: 12 heads on 8 ranks -> 4 x 2), as the CLI chooses."""
        if lay[1] <= 1:
            return 1
        if lay[2] > 0:
            return lay[2]
        return args.ring_degree if args.ring_degree > 1 else vdist.choose_ring_degree(lay[1], mk["num_heads"], 1)

    configured = [None]

    def configure(lay):
        """Point the model at one layout's groups.  A no-op when the model already is in that layout: re-enabling would make the
        engine destroy and re-create its communicators -- a second blocking rendezvous, outside the supervised bring-up window."""
        if not use_dist or configured[0] == lay:
            return
        configured[0] = lay
        sp_group, bp_group = groups[lay]
        vdist.use_groups(sp_group, bp_group)
        if lay[1] == 1 and lay[0] > 1:
            model.enable_multi_gpus_inference()                        # batch-parallel only: no sequence exchange
        else:
            if lay not in sps:
                sps[lay] = vdist.SequenceParallel(sp_group, force_exchange=(world == 1), ring_degree=ring_of(lay))
            model.enable_multi_gpus_inference(sps[lay])

    # ---- bring-up, BEFORE any weight exists: every layout's communicators are created and each carries one probe collective;
    # what the transports report about their size is kept for the JSON line.  Timed order = `layouts`; bring-up runs it in
    # reverse so that the model ends up configured for the first one.
    for lay in reversed(layouts):
        configure(lay)
        obs = {}
        if use_dist:
            model.attach_communicators()
            if lay[1] > 1 or world == 1:
                obs["sp"] = model.probe_exchange(dev)
            if lay[0] > 1:
                obs["cfg"] = model._bp.observed_ranks(dev)
        observed[lay] = obs
    _mark("up")

    # ---- random weights of the named architecture, identical on every rank (seed 0) ----
    model.init_weights(zero_init_outputs=False)
    if args.fp8_linear:
        model.enable_fp8_linear()
    if args.fp8_attn >= 0:
        model.enable_fp8_attention(True, args.fp8_attn)
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

    def timed(lay):
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, MAX over ranks."""
        configure(lay)
        scheduler.set_timesteps(args.num_inference_steps, device=dev, shift=16)
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
        if use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            ff = torch.tensor([1.0 if finite else 0.0])
            dist.all_reduce(ff, op=dist.ReduceOp.MIN)
            finite = bool(ff.item() > 0)
        # what the transports of this layout say about their size NOW, after they carried the timed steps
        obs = dict(observed.get(lay, {}))
        if use_dist and (lay[1] > 1 or world == 1) and model._sp is not None:
            obs["sp"] = dict(obs.get("sp", {}), ranks=model.sp_observed_ranks(dev))
        return {"layout": lay, "elapsed": elapsed, "prof": prof, "finite": finite, "observed": obs,
                "transport": getattr(model._sp, "transport", "none") if lay[1] > 1 or (world == 1 and use_dist) else "none"}

    def build_out(results, tea, final, skipped=()):
        def line_of(res):
            cfgd, spd = res["layout"][:2]
            sps_ = args.steps / res["elapsed"]
            obs = res["observed"]
            sp_ranks = int(obs.get("sp", {}).get("ranks", 0) or 0)
            cfg_ranks = int(obs.get("cfg", {}).get("ranks", 0) or 0)
            on_rccl = (not rehearsal) and use_dist
            rccl_ranks = 0
            if on_rccl:
                rccl_ranks = (sp_ranks if spd > 1 or world == 1 else 1) * (cfg_ranks if cfgd > 1 else 1)
                if spd == 1 and cfgd == 1 and world > 1:
                    rccl_ranks = 0
            if spd > 1 or (world == 1 and use_dist):
                tr = ("gloo + host-staged buffers (REHEARSAL of the launch, not a result)" if rehearsal else
                      "engine-owned RCCL communicators, one per stream lane" if res["transport"] == "rccl" else
                      "torch.distributed RCCL process groups, one per stream lane (engine-owned communicators unavailable)")
                if cfgd > 1:
                    tr += "; the CFG pair split over rank groups, noise prediction all-gathered over torch.distributed's RCCL group"
            elif cfgd > 1:
                tr = ("gloo (REHEARSAL)" if rehearsal else
                      "one CFG sample per rank, no data-path collective; the noise prediction is all-gathered over "
                      "torch.distributed's RCCL group")
            else:
                tr = "none (single rank)"
            out = {"value": sps_, "ms_per_step": 1000.0 * res["elapsed"] / args.steps,
                   "parallelism": (f"ulysses-sp{spd}" if cfgd == 1 else f"cfg{cfgd} x ulysses-sp{spd}") +
                                  (f" (ulysses {spd // ring_of(res['layout'])} x ring {ring_of(res['layout'])})" if ring_of(res["layout"]) > 1 else ""),
                   "cfg": "batched pair" if cfgd == 1 else "one sample per rank",
                   "step_mfma_frac": f_step * sps_ / (world * PEAK_BF16_TFLOPS * 1e12),
                   "outputs_finite": res["finite"],
                   # ranks counted BY the transports (ncclCommCount of the engine's communicator after the timed steps; the sum
                   # of an all-reduce of ones on the CFG group's RCCL backend): 0 = no RCCL collective ran
                   "rccl_ranks": rccl_ranks, "rccl_observed": obs, "transport": tr}
            return out
        lines = [line_of(r) for r in results]
        best = max(range(len(lines)), key=lambda i: lines[i]["value"])
        head, res = lines[best], results[best]
        sps = head["value"]
        out = {
            "metric": "denoise-steps/sec Wan2.1-14B+GeoAdapter 81fx480p" if args.workload == "wan14b-81f-480x832"
                      else f"denoise-steps/sec {args.workload}",
            "value": sps, "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": dtype_label(args), "data": "synthetic (random weights seed 0, inputs seed 2025)",
            "config": {"workload": args.workload, "latent": [16, T, h, w], "tokens": L, "global_batch": 2,
                       "cfg": head["cfg"], "guidance_scale": 5.0, "sampler": "UniPC shift 16",
                       "teacache": "off", "parallelism": head["parallelism"], "pflop_per_step": f_step / 1e15},
            "step_mfma_frac": head["step_mfma_frac"],
            # the same with the output-neutral work the engine skips taken out of the numerator (see skipped_flops)
            "step_mfma_frac_executed": (f_step - skipped_flops(mk["dim"], NL, NA, L, 2, (n_un, n_co), mk.get("text_len", 512),
                                                               text_dim)) * sps / (world * PEAK_BF16_TFLOPS * 1e12),
            "outputs_finite": head["outputs_finite"],
            "rccl_ranks": head["rccl_ranks"], "rccl_observed": head["rccl_observed"], "transport": head["transport"],
        }
        if len(lines) > 1:
            # the other layout of the same run, each timed over the same K steps after the same W warm-up steps
            out["alt"] = [dict(ln, note="not the headline: slower layout of this run") for i, ln in enumerate(lines) if i != best]
        prof = res["prof"]
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
            # fp8 modes: the dominant kernel is priced against the dense fp8 peak when it runs in fp8, and the whole step against the
            # MIXED roofline -- time the step's fp8 FLOPs need at the fp8 peak plus its bf16 FLOPs at the bf16 peak (the class shares of
            # the live profile applied to the algorithmic count) -- never against the bf16 peak alone
            fp8_cls = {"gemm": args.fp8_linear, "attn_self": args.fp8_attn >= 0}
            peak = PEAK_FP8_TFLOPS if fp8_cls.get(dom) else PEAK_BF16_TFLOPS
            if fp8_cls.get(dom):
                kname = {"attn_self": "attn_fp8_kernel", "gemm": "gemm_pp_kernel<FP8>"}[dom]
                traffic, traffic_src = None, None
                try:
                    if tj.get("workload") == args.workload and tj.get("n_gpus") == world and kname in tj:
                        traffic, traffic_src = tj[kname]["bytes_per_launch"], tj.get("source")
                except NameError:
                    pass
            if any(fp8_cls.values()):
                tot = sum(vv["flops"] for vv in prof.values()) or 1.0
                f8 = sum(prof[c]["flops"] for c, on in fp8_cls.items() if on) / tot * f_step
                t_roof = f8 / (PEAK_FP8_TFLOPS * 1e12) + (f_step - f8) / (PEAK_BF16_TFLOPS * 1e12)
                out["step_mixed_roofline_frac"] = t_roof * sps / world
                out["step_mfma_frac_note"] = ("step_mfma_frac divides by the bf16 peak and is NOT a roofline fraction of this mode; "
                                              "step_mixed_roofline_frac prices the fp8 classes at 5 PF and the rest at 2.5 PF")
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": traffic,
                               "traffic_source": (f"recorded, not live: {traffic_src}" if traffic is not None else None),
                               "avg_launch_ms": v["ms"] / v["launches"], "launches": v["launches"]}
            if world > 1:
                out["roofline"]["note"] = ("sequence-parallel run: the two samples of the CFG pair are scheduled on separate "
                                           "streams / interleaved with exchanges, so kernel launches can overlap and the per-launch "
                                           "durations (hence 'achieved') are lower bounds; the N=1 line carries the kernel roofline")
            out["breakdown"] = bd
        if tea is not None:
            out["teacache_on"] = tea
        if skipped:
            out["alt_skipped"] = [f"cfg{c} x ulysses-sp{sp_}" + (f" ring {r_}" if r_ else "") + ": not timed (the run was past its time mark)" for c, sp_, r_ in skipped]
        if final and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(mk, f_step, L)
        return out

    # Layouts are timed in order.  With alternatives (N >= 4) the first layout's line is SAVED the moment it exists (the supervisor
    # prints it if a later layout takes the run down), and an alternative is only started while the run is young enough
    # (VC_BENCH_ALT_DEADLINE seconds since this rank started, default 300; every rank takes rank 0's decision).
    results, skipped = [], []
    alt_deadline = float(os.environ.get("VC_BENCH_ALT_DEADLINE", "300"))
    for i, lay in enumerate(layouts):
        if i > 0 and world >= 4:
            go = torch.tensor([1.0 if time.time() - T_RANK_START < alt_deadline else 0.0])
            if use_dist:
                dist.broadcast(go, src=0)
            if go.item() < 0.5:
                skipped.append(lay)
                continue
        results.append(timed(lay))
        if i == 0 and len(layouts) > 1 and world >= 4 and rank == 0:
            d = os.environ.get("VC_BENCH_STATUS_DIR")
            if d:
                part = build_out(results, None, False)
                part["alt_note"] = "first layout only: the run ended before its alternative layouts were timed"
                with open(os.path.join(d, f"rank0.a{os.environ.get('VC_BENCH_ATTEMPT', '0')}.partial"), "w") as fh:
                    fh.write(json.dumps(part) + "\n")

    tea = None
    if world == 1 and not use_dist and not args.no_teacache_line:
        tea = teacache_line(args, model, pipe, scheduler, latents, embeds, geoada_in, seq_len, ts, dev)

    if rank == 0:
        real_stdout.write(json.dumps(build_out(results, tea, True, skipped)) + "\n")
        real_stdout.flush()
    if use_dist:
        dist.all_reduce(torch.zeros(1))
        dist.destroy_process_group()


def teacache_line(args, model, pipe, scheduler, latents, embeds, geoada_in, seq_len, ts, dev):
    """SURVEY 8d's separate TeaCache-on line, in the only form random weights can fill (the gate's statistics -- whether a step
    IS skipped -- belong to the released checkpoint; with random weights the time embedding's relative L1 change never drops
    under the CLI's threshold 0.10 and nothing is skipped: profiles/r02_v5_bench_teacache_50steps.json).  What a skipped step
    and a calc + store step COST is a property of the engine: the gate is forced (calc, skip, skip, calc, skip, calc) and every
    step is timed with the device drained on both sides; the implied rate over the CLI's window (50 steps, the first 5 always
    computed: CLI.py:104-116) is listed as a function of the number of skipped steps."""
    import torch
    coeff_14b = [8.10705460e+03, 2.13393892e+03, -3.72934672e+02, 1.66203073e+01, -4.17769401e-02]   # CLI.py:305-313
    n_steps = args.num_inference_steps
    model.enable_teacache(coeff_14b, n_steps, 0.10, num_skip_start_steps=5, offload=False)
    scheduler.set_timesteps(n_steps, device=dev, shift=16)
    forced = [True, False, False, True, False, True]
    it = iter(forced)
    tc = model.teacache

    def gate(e0):
        tc.previous_modulated_input = e0
        tc.should_calc = next(it)
        return tc.should_calc
    tc.gate = gate
    calc_ms, skip_ms = [], []
    lt = latents
    for i, calc in enumerate(forced):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lt = pipe.denoise_step(i, ts[i], lt, embeds, geoada_in, seq_len, True, 1.0)
        torch.cuda.synchronize()
        (calc_ms if calc else skip_ms).append(1000.0 * (time.perf_counter() - t0))
    model.disable_teacache()
    finite = bool(torch.isfinite(lt.float()).all().item())
    c = sorted(calc_ms[1:])[len(calc_ms[1:]) // 2]          # the first calc + store step also allocates the residual slot
    k = sorted(skip_ms)[len(skip_ms) // 2]
    window = n_steps - 5
    implied = {str(j): n_steps / (((n_steps - j) * c + j * k) / 1000.0) for j in (0, 5, 10, 15, 20, 25, 30) if j <= window}
    return {"calc_step_ms": c, "skipped_step_ms": k, "calc_steps_ms_all": calc_ms, "skipped_steps_ms_all": skip_ms,
            "outputs_finite": finite, "unit": "denoise-steps/s", "num_inference_steps": n_steps, "threshold": 0.10,
            "num_skip_start_steps": 5, "implied_steps_per_s_by_skipped_steps": implied,
            "note": "gate forced (calc, skip, skip, calc, skip, calc): the cost of each step type on this GPU; how MANY steps the "
                    "gate skips needs the released weights (random weights skip none at threshold 0.10)"}


if __name__ == "__main__":
    main()

"""`python bench.py --gpus N` exactly as the driver calls it: no launcher around it, the script starts its own N rank
processes (one per GPU, torchrun's environment contract -- the reference's `torchrun --nproc-per-node=N`,
inference.sh:62-71), forwards rank 0's single JSON line and fails when a rank fails."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=e, capture_output=True,
                          text=True, timeout=timeout)


def test_self_launch_propagates_rank_failure():
    """No visible GPU: every rank dies on its first assertion; the parent (which never touches the GPU) must stop the other
    ranks, print nothing on stdout and exit non-zero."""
    r = _run(["--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "0", "--backend", "gloo"],
             env={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""}, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "exited with code" in r.stderr


def test_launcher_parent_does_not_import_torch():
    """The supervisor must start the ranks before anything can touch the GPU: bench.py's module level imports neither torch nor
    the engine, launch_ranks imports nothing of either (its torchrun-mode board imports torch.distributed's STORE only -- host
    sockets, no device), and a multi-rank run_rank happens only in a child (VC_BENCH_CHILD)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for line in src.splitlines():                               # module level
        assert not line.startswith(("import torch", "from torch", "from versecrafter_amd", "import versecrafter_amd")), line
    body = src[src.index("def launch_ranks("):src.index("def _mark(")]
    stmts = [ln.strip() for ln in body.splitlines() if ln.strip().startswith(("import ", "from "))]
    assert stmts and not any("torch" in ln or "versecrafter_amd" in ln for ln in stmts), stmts
    board = src[src.index("class _StoreBoard"):src.index("def _tail(")]
    stmts = [ln.strip() for ln in board.splitlines() if ln.strip().startswith(("import ", "from "))]
    assert all(ln.startswith(("from datetime", "from torch.distributed import PrefixStore, rendezvous")) for ln in stmts), stmts
    main = src[src.index("def main("):src.index("def run_rank(")]
    assert main.index("launch_ranks(args)") < main.index("run_rank(args)") and "VC_BENCH_CHILD" in main


FAKE = os.path.join(ROOT, "tests", "_fake_bench_rank.py")
FAST = {"VC_BENCH_UP_TIMEOUT": "2", "VC_BENCH_START_TIMEOUT": "60", "VC_BENCH_KILL_GRACE": "1", "VC_BENCH_TEST_CHILD": FAKE}


def _err_files(tmp_path, n):
    return [tmp_path / f"bench_n{n}.rank{r}.err" for r in range(n)]


def test_supervisor_stall_starts_a_fresh_set_on_the_torch_transport(tmp_path):
    """First contact of a multi-rank run must produce a line, not a timeout: children that block instead of bringing their
    communicators up are killed after the bring-up deadline and a FRESH set runs with VC_SP_TRANSPORT=torch."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "3"], env=dict(FAST, FAKE_MODE="stall_first", VC_BENCH_LOG_DIR=str(tmp_path)), timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert time.time() - t0 < 60
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 3 and out["transport"] == "torch"
    assert "stall" in r.stderr and "fresh set of ranks with VC_SP_TRANSPORT=torch" in r.stderr
    for f in _err_files(tmp_path, 3):                           # one stderr file per rank, both attempts in it
        txt = f.read_text()
        assert "==== attempt 0" in txt and "==== attempt 1" in txt and "blocking in the rendezvous" in txt


def test_supervisor_gives_up_with_per_rank_tails_when_the_fallback_stalls_too(tmp_path):
    r = _run(["--gpus", "2"], env=dict(FAST, FAKE_MODE="stall_always", VC_BENCH_LOG_DIR=str(tmp_path)), timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert r.stderr.count("stall") >= 2
    assert "---- rank 0 stderr tail" in r.stderr and "---- rank 1 stderr tail" in r.stderr
    assert "blocking in the rendezvous" in r.stderr


def test_supervisor_budget_bounds_the_whole_run(tmp_path):
    import time
    t0 = time.time()
    r = _run(["--gpus", "2"], env=dict(FAST, FAKE_MODE="stall_always", VC_BENCH_UP_TIMEOUT="30", VC_BENCH_BUDGET="3",
                                       VC_BENCH_LOG_DIR=str(tmp_path)), timeout=120)
    assert r.returncode != 0 and "over the run budget" in r.stderr
    assert time.time() - t0 < 30


@pytest.mark.parametrize("mode,needle", [("hang_in_alt", "over the run budget"), ("die_in_alt", "rank 1 exited with code 5")])
def test_supervisor_reports_the_first_layout_when_an_alternative_layout_takes_the_run_down(tmp_path, mode, needle):
    """N >= 4 times Ulysses-N first and saves its line; the alternative layouts of the same run (CFG pair on two Ulysses groups, the
    reference's Ulysses 2 x ring N/2) come after it.  A hang or a death inside one of them must not cost the measured line."""
    r = _run(["--gpus", "2"], env=dict(FAST, FAKE_MODE=mode, VC_BENCH_BUDGET="6", VC_BENCH_LOG_DIR=str(tmp_path)), timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip())
    assert out["value"] == 2.5 and out["alt_note"] == "first layout only" and needle in out["alt_error"]
    assert "AFTER the first layout had been measured" in r.stderr


def test_supervisor_rank_death_during_bring_up_is_a_transport_failure(tmp_path):
    """A rank that dies between "started" and "up" (ncclCommInitRank returned an error on one side) leaves its peers blocked:
    they are killed after the grace period and the fresh set takes the torch transport."""
    r = _run(["--gpus", "2"], env=dict(FAST, FAKE_MODE="die_in_bringup", VC_BENCH_LOG_DIR=str(tmp_path)), timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["transport"] == "torch"
    assert "transport: rank 1 exited with code 3" in r.stderr


def test_supervisor_under_torchrun_one_supervisor_per_rank(tmp_path):
    """The driver's N > 1 command line: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N.  Every torchrun
    worker supervises its own rank's child; they agree through torchrun's store (stall -> everybody restarts on torch); the
    children rendezvous on a port of their own; rank 0's supervisor prints the one line."""
    e = dict(os.environ, **FAST, FAKE_MODE="stall_first", VC_BENCH_LOG_DIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["transport"] == "torch" and int(out["master_port"]) != port
    assert all(f.exists() for f in _err_files(tmp_path, 2))


@pytest.mark.gpu
def test_self_launch_two_ranks_rehearsal_on_one_gpu():
    """Two rank processes on this box's one GPU (RCCL refuses two ranks per device, so the rehearsal backend is gloo with
    host-staged exchange buffers): rendezvous, sequence-parallel engine in both ranks, barrier + max-over-ranks timing, ONE
    JSON line with n_gpus == 2 on the parent's stdout."""
    r = _run(["--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline",
              "--cfg-degree", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["parallelism"] == "ulysses-sp2" and out["scaling"] == "strong"
    assert out["outputs_finite"] is True and out["value"] > 0
    assert out["rccl_ranks"] == 0 and "REHEARSAL" in out["transport"]


@pytest.mark.gpu
def test_self_launch_two_ranks_default_measures_both_layouts():
    """--gpus 2 without --cfg-degree: BOTH ways of using two ranks are timed in the one run -- Ulysses over the two ranks
    (north_star's curve) and one CFG sample per rank (no sequence exchange, outputs gathered); the faster is the headline, the
    other sits under "alt" with its own value; --cfg-degree 2 pins the split."""
    r = _run(["--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and len(out["alt"]) == 1
    both = {out["config"]["parallelism"]: out, out["alt"][0]["parallelism"]: out["alt"][0]}
    assert set(both) == {"ulysses-sp2", "cfg2 x ulysses-sp1"}
    assert out["value"] >= out["alt"][0]["value"] > 0
    assert out["outputs_finite"] is True and out["alt"][0]["outputs_finite"] is True and "REHEARSAL" in out["transport"]
    cfg_line = both["cfg2 x ulysses-sp1"]
    assert (cfg_line.get("cfg") or cfg_line["config"]["cfg"]) == "one sample per rank"
    assert cfg_line["rccl_observed"]["cfg"] == {"ranks": 2, "backend": "gloo"}       # counted by an all-reduce, not computed
    r = _run(["--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline",
              "--cfg-degree", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip())
    assert out["config"]["parallelism"] == "cfg2 x ulysses-sp1" and "alt" not in out


@pytest.mark.gpu
def test_driver_command_line_under_torchrun_rehearsal_on_one_gpu(tmp_path):
    """The driver's N > 1 command line verbatim -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W -- with real rank children on this box's one GPU (gloo
    rehearsal): every torchrun worker supervises its own rank, the children rendezvous on their own port, bring-up probes run, both
    N = 2 layouts are timed, ONE JSON line comes out of rank 0's supervisor, and every rank leaves its stderr file."""
    import socket
    e = dict(os.environ, VC_BENCH_LOG_DIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "tiny", "--backend", "gloo", "--no-cpu-baseline"], cwd=ROOT, env=e, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["outputs_finite"] is True
    assert {out["config"]["parallelism"], out["alt"][0]["parallelism"]} == {"ulysses-sp2", "cfg2 x ulysses-sp1"}
    sp = out if out["config"]["parallelism"] == "ulysses-sp2" else out["alt"][0]
    assert sp["rccl_observed"]["sp"]["ranks"] == 2                      # counted by a collective over the lane group (gloo here)
    for rk in range(2):
        assert "==== attempt 0" in (tmp_path / f"bench_n2.rank{rk}.err").read_text()


@pytest.mark.gpu
def test_bench_ring_degree_rehearsal_on_one_gpu():
    """--ring-degree 2 on two ranks (a pure ring: Ulysses degree 1): the hybrid's bring-up probe (sub-group all-to-all + ring pass through
    the engine's entry points) and two timed steps, gloo rehearsal."""
    r = _run(["--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline",
              "--cfg-degree", "1", "--ring-degree", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip())
    assert out["config"]["parallelism"] == "ulysses-sp2 (ulysses 1 x ring 2)" and out["outputs_finite"] is True
    assert out["rccl_observed"]["sp"]["ring_degree"] == 2 and out["rccl_observed"]["sp"]["ranks"] == 2


@pytest.mark.gpu
def test_single_rank_rccl_exchange_path_through_bench():
    """N = 1 with the N > 1 plumbing forced on (VC_BENCH_FORCE_DIST=1): process group, ncclUniqueId hand-over, the engine's
    two RCCL communicators (world 1), and the whole exchange path -- pack, ncclAllToAll, segmented attention, ncclAllToAll,
    unpack, ncclAllGather -- on both block chains."""
    r = _run(["--gpus", "1", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
             env={"VC_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533", "RANK": "0",
                  "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["outputs_finite"] is True


@pytest.mark.gpu
def test_four_rank_rehearsals_on_one_gpu():
    """--gpus 4 (the driver's next point after 2): `tiny` has 2 heads, so four ranks cannot be pure Ulysses -- the bench picks the
    Ulysses 2 x ring 2 hybrid itself, as the CLI does for the 1.3B model on 8 GPUs; --cfg-degree 2 gives two Ulysses pairs."""
    r = _run(["--gpus", "4", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline", "--single-layout"],
             timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 4 and out["config"]["parallelism"] == "ulysses-sp4 (ulysses 2 x ring 2)" and "alt" not in out
    assert out["outputs_finite"] is True and out["rccl_observed"]["sp"]["ranks"] == 4 and out["rccl_observed"]["sp"]["ring_degree"] == 2
    # the driver's own command line (round 4): Ulysses over the four ranks FIRST, then the CFG pair on two Ulysses pairs in the same run;
    # the faster is the headline, the other sits under "alt" (the third layout, the reference's Ulysses 2 x ring N/2, needs a head count
    # that divides by N: not this 2-head model)
    r = _run(["--gpus", "4", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline"], timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip())
    both = {out["config"]["parallelism"]} | {a["parallelism"] for a in out["alt"]}
    assert both == {"ulysses-sp4 (ulysses 2 x ring 2)", "cfg2 x ulysses-sp2"} and len(out["alt"]) == 1
    assert out["outputs_finite"] is True and all(a["outputs_finite"] for a in out["alt"]) and "alt_error" not in out
    # a model whose heads divide by 4: all three layouts in one run, the third being the reference's documented Ulysses 2 x ring N/2
    r = _run(["--gpus", "4", "--workload", "tiny4h", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline"], timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip())
    three = {out["config"]["parallelism"]} | {a["parallelism"] for a in out["alt"]}
    assert three == {"ulysses-sp4", "cfg2 x ulysses-sp2", "ulysses-sp4 (ulysses 2 x ring 2)"}, three
    assert out["outputs_finite"] is True and all(a["outputs_finite"] for a in out["alt"])
    # ... and none of them when the run is already old (the time mark): the first layout alone, the others listed as skipped
    r = _run(["--gpus", "4", "--workload", "tiny4h", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline"],
             env={"VC_BENCH_ALT_DEADLINE": "0"}, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip())
    assert out["config"]["parallelism"] == "ulysses-sp4" and "alt" not in out and len(out["alt_skipped"]) == 2
    r = _run(["--gpus", "4", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline",
              "--cfg-degree", "2"], timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip())
    assert out["config"]["parallelism"] == "cfg2 x ulysses-sp2" and out["outputs_finite"] is True
    assert out["rccl_observed"] == {"sp": {"ranks": 2, "transport": "torch", "ring_degree": 1}, "cfg": {"ranks": 2, "backend": "gloo"}}

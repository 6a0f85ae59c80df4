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
    """The parent must start the ranks before anything can touch the GPU: bench.py's module level and launch_ranks import
    neither torch nor the engine."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for line in src.splitlines():                               # module level
        assert not line.startswith(("import torch", "from torch", "from versecrafter_amd", "import versecrafter_amd")), line
    body = src[src.index("def launch_ranks("):src.index("def main(")]
    stmts = [ln.strip() for ln in body.splitlines() if ln.strip().startswith(("import ", "from "))]
    assert stmts and not any("torch" in ln or "versecrafter_amd" in ln for ln in stmts), stmts
    main = src[src.index("def main("):src.index("def run_rank(")]
    assert main.index("launch_ranks(args)") < main.index("run_rank(args)")


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
def test_self_launch_two_ranks_default_is_one_cfg_sample_per_rank():
    """--gpus 2 without --cfg-degree: the CFG pair is split across the two ranks (no sequence exchange), outputs gathered."""
    r = _run(["--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "cfg2 x ulysses-sp1"
    assert out["outputs_finite"] is True and out["value"] > 0 and "REHEARSAL" in out["transport"]


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

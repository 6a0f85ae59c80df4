"""Process groups of dist.set_multi_gpus_devices with cfg_degree > 1 (world = cfg_degree * ulysses * ring), on CPU with gloo:
group membership, BatchParallel.gather / broadcast."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sp_degree, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from versecrafter_amd import dist as vdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vdist.set_multi_gpus_devices(sp_degree, 1, cfg_degree=world // sp_degree)
        sp, bpg = vdist.get_sp_group(), vdist.get_bp_group()
        info = dict(rank=rank,
                    sp_ranks=None if sp is None else [dist.get_global_rank(sp, i) for i in range(dist.get_world_size(sp))],
                    bp_ranks=[dist.get_global_rank(bpg, i) for i in range(dist.get_world_size(bpg))],
                    sp_world=vdist.get_sequence_parallel_world_size(), sp_rank=vdist.get_sequence_parallel_rank())
        bp = vdist.BatchParallel(bpg)
        x = torch.full((1, 3), float(rank))
        info["gathered"] = bp.gather(x)[:, 0].tolist()
        y = torch.full((2,), float(rank))
        bp.broadcast(y, bp.world_size - 1)
        info["bcast"] = y.tolist()
        q.put(info)
    finally:
        dist.destroy_process_group()


def _run(world, sp_degree):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sp_degree, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda d: d["rank"])
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    return res


def test_cfg2_groups_one_rank_per_sample():
    res = _run(2, 1)
    for r, d in enumerate(res):
        assert d["sp_ranks"] is None and d["sp_world"] == 1 and d["sp_rank"] == 0
        assert d["bp_ranks"] == [0, 1] and d["gathered"] == [0.0, 1.0] and d["bcast"] == [1.0, 1.0]


def test_cfg2_x_ulysses2_groups():
    res = _run(4, 2)
    for r, d in enumerate(res):
        base = r - r % 2
        assert d["sp_ranks"] == [base, base + 1] and d["sp_world"] == 2 and d["sp_rank"] == r % 2
        assert d["bp_ranks"] == [r % 2, r % 2 + 2]
        assert d["gathered"] == [float(r % 2), float(r % 2 + 2)]
        assert d["bcast"] == [float(r % 2 + 2)] * 2

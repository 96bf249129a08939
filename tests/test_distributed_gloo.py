"""world_size-2 test of the sharding + gather path on CPU (gloo).  The per-shard "solver" here is
the CPU oracle (tests only); on the GPU box the same code runs with the HIP kernels + RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, tmp):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    from toppra_amd import batch, distributed

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = batch.make_synthetic_batch(B, 3, 30, seed=99)
    arrays = {k: data[k] for k in ("coef", "breaks", "grid", "vlim", "alim")}

    def solve(shard):
        return orc.solve_batch(shard["coef"], shard["breaks"], shard["grid"], shard["vlim"], shard["alim"])

    local, full = distributed.solve_sharded(arrays, solve, torch.from_numpy)
    lo, hi = distributed.shard_bounds(B, world, rank)
    assert local["sd2"].shape[0] == hi - lo
    if rank == 0:
        np.save(os.path.join(tmp, "full.npy"), full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [16, 13])
def test_shard_and_gather_world2(tmp_path, B):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    from toppra_amd import batch

    port = _free_port()
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    full = np.load(os.path.join(str(tmp_path), "full.npy"))
    data = batch.make_synthetic_batch(B, 3, 30, seed=99)
    ref = orc.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    assert full.shape == ref["sd2"].shape
    assert np.array_equal(full, ref["sd2"], equal_nan=True)


def test_shard_bounds_partition():
    from toppra_amd import distributed
    for total in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            spans = [distributed.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _pipeline_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from toppra_amd import distributed

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pg = distributed.PipelinedGather(4, 3, torch.float64, "cpu")
    seen = []
    for step in range(5):  # step k's gather overlaps step k+1's "solve"
        local = torch.full((4, 3), float(10 * step + rank), dtype=torch.float64)
        pg.submit(local)
        if step % 2 == 1:
            # "sequential" placement (bench.py --gather sequential): the next solve is ordered after this gather;
            # afterwards the receive buffers hold this step's rows
            pg.order_after()
            if rank == 0:
                for r in range(world):
                    assert torch.equal(pg.bufs[r], torch.full((4, 3), float(10 * step + r), dtype=torch.float64))
    bufs = pg.finish()
    if rank == 0:
        assert len(bufs) == world
        for r in range(world):
            assert torch.equal(bufs[r], torch.full((4, 3), float(40 + r), dtype=torch.float64))
        open(os.path.join(tmp, "ok"), "w").write("1")
    else:
        assert bufs is None
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_world2(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_pipeline_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok"))

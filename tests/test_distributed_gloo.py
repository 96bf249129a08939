"""world_size-2 and -8 tests of the sharding + gather path on CPU (gloo).  The per-shard "solver" here is
the CPU oracle (tests only); on the GPU box the same code runs with the HIP kernels + RCCL.  The multi-rank control
flow of bench.py itself (gather placement calibration, pipelined gather, max-over-ranks timing, the rank-0 JSON line)
is rehearsed at world size 8 with its stub solver."""
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


@pytest.mark.parametrize("world,B", [(2, 16), (2, 13), (8, 16), (8, 13), (8, 5)])
def test_shard_and_gather(tmp_path, world, B):
    """Equal and ragged shards (13 rows over 8 ranks: 2,2,2,2,2,1,1,1; 5 rows over 8 ranks: three ranks own nothing)."""
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    from toppra_amd import batch

    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
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


@pytest.mark.parametrize("gather", ["auto", "overlap", "sequential"])
def test_bench_multi_rank_control_flow_world8(gather):
    """`bench.py --gpus 8` as the driver launches it, with the stub solver on CPU over gloo: all eight ranks must run
    the calibration / timed loop / gather in step (a mismatch in the number of collectives would hang: hence the
    timeout), rank 0 alone prints ONE JSON line with the contract's keys, the whole-job batch, the per-rank kernel
    times, the gather-alone time and the placement, and its receive buffers hold every rank's last step."""
    import json
    import subprocess

    world, port = 8, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--stub-solver", "--steps", "4",
                                       "--warmup", "3", "--batch", "37", "--gridpoints", "20", "--gather", gather],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        try:
            o, e = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("bench.py --gpus 8 --stub-solver hung (ranks out of step?)")
        assert pr.returncode == 0, e[-2000:]
        outs.append(o)
    lines = [[l for l in o.splitlines() if l.startswith("{")] for o in outs]
    assert len(lines[0]) == 1 and all(len(l) == 0 for l in lines[1:]), "rank 0 alone prints one JSON line"
    line = json.loads(lines[0][0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "per_rank_kernel_ms", "gather_alone_ms", "gather_placement"):
        assert key in line, key
    assert line["n_gpus"] == 8 and line["steps"] == 4 and line["warmup"] == 3 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 8 * 37 and "shard8+rccl_gather" in line["config"]["parallelism"]
    assert line["value"] == pytest.approx(8 * 37 * 4 / (line["ms_per_step"] * 4 * 1e-3), rel=1e-9)
    assert line["per_rank_kernel_ms"] == pytest.approx([1.0 + 0.01 * r for r in range(8)])
    assert line["gather_alone_ms"] > 0
    placement = line["gather_placement"]
    if gather == "auto":
        assert placement["chosen"] in ("overlap", "sequential") and set(placement["ms_per_step"]) == {"overlap", "sequential"}
    else:
        assert placement["chosen"] == gather
    assert line["stub_gather_delivered_every_ranks_last_step"] is True


def test_bench_launches_its_own_ranks_world8():
    """Plain `python bench.py --gpus 8 --stub-solver` with NO launcher and no rank variables in the environment (the way the
    driver runs the N = 1 bench): bench.py must start the eight ranks itself and still print exactly one JSON line."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    try:
        pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--stub-solver", "--steps", "3",
                             "--warmup", "2", "--batch", "19", "--gridpoints", "12"], env=env, capture_output=True, text=True,
                            timeout=300)
    except subprocess.TimeoutExpired:
        pytest.fail("plain `bench.py --gpus 8 --stub-solver` hung")
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["config"]["global_batch"] == 8 * 19 and line["steps"] == 3
    assert line["stub_gather_delivered_every_ranks_last_step"] is True

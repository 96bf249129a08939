"""Multi-GPU sharding of a trajectory batch: one process per GPU, no data-path collective.

Trajectories are independent, so a batch of B problems is cut into contiguous shards, every rank
solves its shard with the single-GPU kernels, and the only communication is the final gather of
``sd^2 [B_local, N+1]`` to rank 0 (RCCL over xGMI with the ``nccl`` backend; ``gloo`` on CPU for the
tests).  On the fully connected xGMI mesh a gather to one root uses the root's seven inbound
links concurrently, which is why this is a gather and not a ring all-gather.
"""
import numpy as np


def shard_bounds(total, world_size, rank):
    """Contiguous [lo, hi) of `total` items owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# Keys of a problem dict that carry one row per trajectory.  Everything else (knots, shared 1-D grid /
# breaks, scalars) passes through untouched: deciding by "leading dimension happens to equal B" would
# silently slice a shared array whose length coincides with the batch size.
PER_TRAJECTORY_KEYS = ("coef", "vlim", "alim", "sd_start", "sd_end", "waypoints", "desired_duration", "sdmin", "sdmax")
MAYBE_PER_TRAJECTORY_KEYS = ("grid", "breaks")  # per trajectory only when 2-D


def shard_problem(arrays, world_size, rank, per_trajectory_keys=PER_TRAJECTORY_KEYS):
    """Slice the per-trajectory arrays of a problem dict to this rank's contiguous shard; shared arrays
    pass through.  Returns (shard dict, (lo, hi))."""
    B = arrays["coef"].shape[0]
    lo, hi = shard_bounds(B, world_size, rank)
    out = {}
    for k, v in arrays.items():
        per_traj = v is not None and hasattr(v, "shape") and (
            (k in per_trajectory_keys and len(v.shape) >= 1) or (k in MAYBE_PER_TRAJECTORY_KEYS and len(v.shape) == 2))
        if per_traj and v.shape[0] != B:
            raise ValueError("%s has %d rows, expected one per trajectory (%d)" % (k, v.shape[0], B))
        out[k] = v[lo:hi] if per_traj else v
    return out, (lo, hi)


def gather_rows(local, total_rows, dst=0, group=None):
    """Gather row-sharded tensors ``local [B_local, ...]`` (torch, contiguous shards in rank order)
    to rank `dst`.  Returns the full ``[total_rows, ...]`` tensor on `dst`, None elsewhere.
    Ragged shards (B % world != 0) are padded to the largest shard for the collective."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    sizes = [shard_bounds(total_rows, world, r) for r in range(world)]
    rows = max(hi - lo for lo, hi in sizes)
    send = local
    if local.shape[0] != rows:
        send = torch.zeros((rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def solve_sharded(arrays, solve_fn, to_tensor, dst=0, group=None):
    """Shard -> solve locally -> gather sd^2 on `dst`.

    `solve_fn(shard_arrays) -> dict with 'sd2'` is the single-GPU solver
    (``toppra_amd.batch.solve_batch`` on device tensors); `to_tensor` converts its 'sd2' to a torch
    tensor on the collective's device.  Returns (local_result, full_sd2_or_None)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    shard, _ = shard_problem(arrays, world, rank)
    local = solve_fn(shard)
    full = gather_rows(to_tensor(local["sd2"]), arrays["coef"].shape[0], dst=dst, group=group)
    return local, full


class PipelinedGather:
    """Gather of equal row shards to `dst`, overlapped with the next step's compute.

    ``submit(local)`` issues the gather of ``local [rows, cols]`` asynchronously -- RCCL runs it on its
    own stream, ordered after the kernels that produced ``local`` -- and returns at once, so the next
    solve is launched while the previous result is still on the xGMI links.  At most one gather is in
    flight: the previous one is waited for (a stream-level wait with RCCL) before the receive buffers
    are reused.  ``finish()`` waits for the last one and returns the receive buffers on `dst`."""

    def __init__(self, rows, cols, dtype, device, dst=0, group=None):
        import torch
        import torch.distributed as dist

        self._dist, self.dst, self.group = dist, dst, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bufs = None
        if self.world > 1 and self.rank == dst:
            self.bufs = [torch.empty((rows, cols), dtype=dtype, device=device) for _ in range(self.world)]
        self._pending = None  # (work, tensor kept alive while the collective reads it)

    def submit(self, local):
        if self.world == 1:
            self._pending = (None, local)
            return
        self._wait()
        work = self._dist.gather(local, self.bufs, dst=self.dst, group=self.group, async_op=True)
        self._pending = (work, local)

    def order_after(self):
        """Order the caller's subsequent work after the gather in flight (RCCL: a stream-level wait on torch's
        current stream, the host does not block; gloo: a host wait)."""
        if self._pending is not None and self._pending[0] is not None:
            self._pending[0].wait()
            self._pending = (None, self._pending[1])

    def _wait(self):
        if self._pending is not None and self._pending[0] is not None:
            self._pending[0].wait()
        self._pending = None

    def finish(self):
        last = self._pending[1] if self._pending is not None else None
        self._wait()
        if self.world == 1:
            return [last] if last is not None else None
        return self.bufs

"""Multi-GPU: one process per GPU, frames sharded contiguously, one tiny collective for colour statistics.

Frames are independent units for every stage of the path: grain noise is a pure function
of (generator state, chunk index, element), the LUT is replicated (<= 431 KB), the 3x3 stencil never crosses a
frame, and colour-match statistics are per frame.  So a batch of F frames is cut into contiguous, chunk-aligned
ranges with no halo and no data-path collective.  The only exchange is the Lab statistics of the *reference*
frame when its rows are split across ranks: per (reference frame, channel) a (n, mean, M2) fp64 triple,
combined with an all-reduce over RCCL/xGMI (72 bytes per reference frame: latency-bound, ~10-30 us).

``torch.distributed`` is plumbing here: backend "nccl" is RCCL on ROCm; the CPU tests use "gloo".
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int, multiple_of: int = 1) -> Tuple[int, int]:
    """Contiguous [start, stop) of rank `rank`; every boundary is a multiple of `multiple_of` (RNG chunk size,
    reference batch) so that chunk-keyed noise and statistics pairing are unchanged by sharding."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    units = -(-n_items // multiple_of)           # ceil
    base, extra = divmod(units, world)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return min(u0 * multiple_of, n_items), min(u1 * multiple_of, n_items)


def row_slice(height: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows of the reference frame reduced by `rank` (BASELINE config 5: H/8 rows per rank)."""
    return shard_range(height, rank, world, 1)


def allreduce_stats(local: torch.Tensor, group=None, mode: str = "allreduce") -> torch.Tensor:
    """Combine per-rank (n, mean, M2) fp64 triples ``[..., 3]`` of disjoint pixel sets into the statistics of
    their union.

    mode="allreduce": two SUM all-reduces -- (n, n*mean) gives the global mean, then M2_r + n_r*(mean_r-mean)^2
                      gives the global M2 (Chan et al. pairwise update, summed form).
    mode="allgather": all-gather the triples and merge them in rank order (bit-deterministic by construction).
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local.clone()
    if local.dtype != torch.float64:
        raise ValueError("statistics must be float64 (n, mean, M2) triples")
    n, mean, m2 = local[..., 0], local[..., 1], local[..., 2]
    if mode == "allgather":
        from .ops import merge_stats
        world = dist.get_world_size(group)
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.contiguous(), group=group)
        return merge_stats(torch.stack(parts, dim=0))
    first = torch.stack([n, n * mean], dim=-1).contiguous()
    dist.all_reduce(first, op=dist.ReduceOp.SUM, group=group)
    n_tot = first[..., 0]
    mean_tot = first[..., 1] / n_tot
    delta = mean - mean_tot
    second = (m2 + n * delta * delta).contiguous()
    dist.all_reduce(second, op=dist.ReduceOp.SUM, group=group)
    return torch.stack([n_tot, mean_tot, second], dim=-1)


def reference_stats_sharded(reference_image: torch.Tensor, rank: int, world: int, group=None,
                            mode: str = "allreduce", cm_math=None) -> torch.Tensor:
    """Lab (mean, std+1e-5) of the reference frame(s) with the rows split across the ranks: each rank reduces
    its H/world rows on its own GPU, then one collective merges the triples.  Returns fp32 ``[R, 3, 2]``."""
    from . import ops
    r0, r1 = row_slice(int(reference_image.shape[1]), rank, world)
    if r1 > r0:
        part = ops.lab_stats(reference_image[:, r0:r1].contiguous(), cm_math)
    else:   # more ranks than rows: contribute the neutral element
        part = torch.zeros((reference_image.shape[0], 3, 3), dtype=torch.float64, device=reference_image.device)
    merged = allreduce_stats(part, group=group, mode=mode)
    return ops.finalize_stats(merged.to(reference_image.device))


def init_from_env(backend: Optional[str] = None, timeout_s: Optional[float] = None):
    """torchrun-style initialisation: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the env.  `timeout_s`: the process
    group's collective timeout (default: torch's -- 10 minutes for RCCL); a caller whose rank 0 works alone for long between two
    collectives (bench.py: output check + CPU baseline before the final barrier) sets it explicitly."""
    import os
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # torchrun (also with a single process)
    if (world > 1 or launched) and not dist.is_initialized():
        if backend is None:
            # VRGDG_DIST_BACKEND=gloo: functional runs of the N-rank flow where RCCL cannot be used (several ranks sharing the
            # one GPU of a test box: RCCL refuses duplicate devices); the product default is RCCL ("nccl" IS RCCL on ROCm)
            backend = os.environ.get("VRGDG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kwargs = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}      # bind the RCCL communicator to this rank's GPU
        if timeout_s is not None:
            import datetime
            kwargs["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local if local < torch.cuda.device_count() else 0)
    return rank, local, world

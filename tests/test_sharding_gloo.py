"""N > 1 path on CPU: world_size-2 gloo.  Frame sharding is chunk aligned and covers every frame exactly
once; the reference-frame statistics merged by the collective equal the statistics of the whole frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, load_package


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_range_properties(pkg):
    from comfyui_vrgamedevgirl_amd.sharding import shard_range
    for n, world, mult in ((2048, 8, 4), (10, 3, 4), (7, 8, 1), (0, 4, 2), (513, 8, 1), (9, 2, 4)):
        spans = [shard_range(n, r, world, mult) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all(s % mult == 0 for s, _ in spans)
        sizes = [e - s for s, e in spans]
        assert max(sizes) - min(sizes) <= 2 * mult - 1     # one unit of imbalance + a ragged last unit
    assert shard_range(2048, 3, 8, 4) == (768, 1024)      # BASELINE config 5: 256 frames per GPU
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    load_package()
    from comfyui_vrgamedevgirl_amd import sharding
    from oracle import truth64
    r, local, w = sharding.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    g = torch.Generator().manual_seed(2024)
    ref = torch.rand(2, 37, 29, 3, generator=g)          # same on every rank; odd height: uneven row split
    r0, r1 = sharding.row_slice(37, rank, world)
    lab = truth64.rgb_to_lab64(ref[:, r0:r1].numpy())
    n = float((r1 - r0) * 29)
    mean = lab.reshape(2, -1, 3).mean(axis=1)
    m2 = ((lab.reshape(2, -1, 3) - mean[:, None, :]) ** 2).sum(axis=1)
    local_stats = torch.from_numpy(np.stack([np.full_like(mean, n), mean, m2], axis=-1))      # [2,3,3] fp64
    merged = sharding.allreduce_stats(local_stats, mode=mode)
    # frame sharding of a 2048-frame job, chunk size 4
    f0, f1 = sharding.shard_range(2048, rank, world, 4)
    counts = torch.tensor([f1 - f0], dtype=torch.int64)
    dist.all_reduce(counts)
    if rank == 0:
        torch.save({"merged": merged, "frames_total": int(counts.item())}, os.path.join(out_dir, f"res_{mode}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "allgather"])
def test_reference_statistics_collective_world2(tmp_path, mode):
    from oracle import truth64
    port = _free_port()
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    res = torch.load(os.path.join(tmp_path, f"res_{mode}.pt"))
    assert res["frames_total"] == 2048
    g = torch.Generator().manual_seed(2024)
    ref = torch.rand(2, 37, 29, 3, generator=g)
    lab = truth64.rgb_to_lab64(ref.numpy()).reshape(2, -1, 3)
    mean = lab.mean(axis=1)
    m2 = ((lab - mean[:, None, :]) ** 2).sum(axis=1)
    merged = res["merged"].numpy()
    assert np.array_equal(merged[..., 0], np.full((2, 3), 37 * 29.0))
    assert np.max(np.abs(merged[..., 1] - mean)) < 1e-11
    assert np.max(np.abs(merged[..., 2] - m2) / m2) < 1e-12


def _oracle_lab_stats(images, cm_math=None):
    """CPU stand-in for ops.lab_stats (the HIP reduction) with the same contract: contiguous fp32 [F,h,W,3] in ->
    fp64 [F,3,3] = (n, mean, M2); used by the gloo test to drive reference_stats_sharded end to end without a GPU."""
    from oracle import truth64
    assert images.dtype == torch.float32 and images.is_contiguous() and images.ndim == 4 and images.shape[-1] == 3
    lab = truth64.rgb_to_lab64(images.numpy()).reshape(images.shape[0], -1, 3)
    n = float(lab.shape[1])
    mean = lab.mean(axis=1)
    m2 = ((lab - mean[:, None, :]) ** 2).sum(axis=1)
    return torch.from_numpy(np.stack([np.full_like(mean, n), mean, m2], axis=-1))


def _oracle_finalize(stats):
    n, mean, m2 = stats[..., 0], stats[..., 1], stats[..., 2]
    sd = torch.sqrt(m2 / (n - 1.0)).to(torch.float32) + np.float32(1e-5)
    return torch.stack([mean.to(torch.float32), sd], dim=-1)


def _worker_sharded(rank, world, port, mode, out_dir, height):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    load_package()
    from comfyui_vrgamedevgirl_amd import ops, sharding
    sharding.init_from_env(backend="gloo")
    seen = []

    def lab_stats(images, cm_math=None):
        seen.append(tuple(images.shape))
        return _oracle_lab_stats(images, cm_math)

    ops.lab_stats, ops.finalize_stats = lab_stats, _oracle_finalize          # the two HIP entry points of the path
    g = torch.Generator().manual_seed(77)
    ref = torch.rand(3, height, 23, 3, generator=g)                          # 3 reference frames, same on every rank
    ms = sharding.reference_stats_sharded(ref, rank, world, mode=mode)
    r0, r1 = sharding.row_slice(height, rank, world)
    assert seen == ([(3, r1 - r0, 23, 3)] if r1 > r0 else []), seen          # each rank reduced exactly its rows, once
    gathered = [torch.empty_like(ms) for _ in range(world)]
    dist.all_gather(gathered, ms)
    assert all(torch.equal(gathered[0], t) for t in gathered)                # every rank holds the same merged statistics
    if rank == 0:
        torch.save(ms, os.path.join(out_dir, f"ms_{mode}_{height}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,height", [("allreduce", 37), ("allgather", 37), ("allreduce", 1)])
def test_reference_stats_sharded_end_to_end_world2(tmp_path, mode, height):
    """sharding.reference_stats_sharded over a 2-rank gloo group: row slicing (uneven split; a rank with no rows when
    height == 1), the contiguity / dtype of what reaches the statistics op, the collective and the fp32 finalisation --
    against the statistics of the whole frames."""
    port = _free_port()
    mp.spawn(_worker_sharded, args=(2, port, mode, str(tmp_path), height), nprocs=2, join=True)
    ms = torch.load(os.path.join(tmp_path, f"ms_{mode}_{height}.pt"))
    g = torch.Generator().manual_seed(77)
    ref = torch.rand(3, height, 23, 3, generator=g)
    want = _oracle_finalize(_oracle_lab_stats(ref.contiguous()))
    assert ms.shape == (3, 3, 2) and ms.dtype == torch.float32
    # fp64 merge of two partials vs one fp64 pass: equal after rounding to fp32 up to 1 ulp on a rounding boundary
    assert torch.allclose(ms, want, rtol=2e-7, atol=0), (ms - want).abs().max()

"""N > 1 path on CPU: world_size-2 gloo.  Surfels are sharded (badslam_amd.distributed.shard_range),
every rank accumulates the per-keyframe Gauss-Newton rows of its shard (with the oracle here,
standing in for the HIP kernel which needs a GPU), the rows go through the same AllReduceHook the
product uses, and every rank must end up with the single-process sums and identical bits."""
import ctypes as C
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from badslam_amd.distributed import AllReduceHook, shard_range
    from tests import bso, scenes
    scene = scenes.synthetic_scene(3, seed=5, width=320, height=240, use_depth_residuals=True, use_descriptor_residuals=False,
                                   camera=bso.make_camera(262.5, 262.5, 160.0, 120.0, 320, 240))
    K, S = len(scene.keyframes), scene.surfels_size
    lo, hi = shard_range(S, rank, world)
    # shard view: same buffers, surfel columns [lo, hi)
    shard = bso.HostScene(scene.color_camera, scene.depth_camera, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, hi - lo)
    shard.surfels = np.ascontiguousarray(scene.surfels[:, lo:hi])
    shard.surfels_size = hi - lo
    shard.keyframes = scene.keyframes
    rows = np.zeros((K, 32), np.float32)
    for k, kf in enumerate(scene.keyframes):
        r = shard.accumulate_pose(kf)
        rows[k, :21], rows[k, 21:27] = r["H"], r["b"]
        rows[k, 28] = np.array([r["count"]], np.uint32).view(np.float32)[0]
    counts_local = rows[:, 28].copy().view(np.uint32).copy()
    rows[:, 28] = 0   # the count column is integer bits; keep it out of the float sum for this check
    hook = AllReduceHook(device=False)
    rc = hook.callback(None, rows.ctypes.data, rows.size, None)
    assert rc == 0 and hook.calls == 1
    np.save(os.path.join(out_dir, f"rows_{rank}.npy"), rows)
    np.save(os.path.join(out_dir, f"counts_{rank}.npy"), counts_local)
    if rank == 0:
        full = np.zeros((K, 27), np.float64)
        cnt = np.zeros(K, np.uint32)
        for k, kf in enumerate(scene.keyframes):
            r = scene.accumulate_pose(kf)
            full[k, :21], full[k, 21:] = r["H64"], r["b64"]
            cnt[k] = r["count"]
        np.save(os.path.join(out_dir, "full.npy"), full)
        np.save(os.path.join(out_dir, "full_counts.npy"), cnt)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_pose_rows_allreduce_gloo(tmp_path, oracle):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rows_0.npy")
    r1 = np.load(tmp_path / "rows_1.npy")
    assert np.array_equal(r0.view(np.uint32), r1.view(np.uint32)), "every rank must hold identical reduced bits"
    full = np.load(tmp_path / "full.npy")
    assert np.abs(r0[:, :27] - full).max() <= 1e-4 * np.abs(full).max()
    counts = np.load(tmp_path / "counts_0.npy").astype(np.uint64) + np.load(tmp_path / "counts_1.npy").astype(np.uint64)
    assert np.array_equal(counts, np.load(tmp_path / "full_counts.npy").astype(np.uint64))


def test_shard_ranges_partition_the_surfels():
    from badslam_amd.distributed import shard_range
    for S in (0, 1, 7, 960000, 19200001):
        for G in (1, 2, 4, 8):
            edges = [shard_range(S, g, G) for g in range(G)]
            assert edges[0][0] == 0 and edges[-1][1] == S
            assert all(edges[i][1] == edges[i + 1][0] for i in range(G - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1

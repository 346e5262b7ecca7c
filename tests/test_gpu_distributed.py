"""-m gpu: the N > 1 product path on ONE card.  Two processes share cuda:0 (RCCL needs one GPU per
rank, so the exchange runs over gloo here; on a multi-GPU node bench.py uses the same hook over
"nccl" == RCCL), each owns one surfel shard, and the HIP kernels' Gauss-Newton rows go through
badslam_amd.distributed.AllReduceHook(device=True) exactly as in bench.py.  Both ranks must end with
bit-identical poses, equal (to fp32 summation-order tolerance) to the single-process result."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    from tests import bso, scenes
    scene = scenes.synthetic_scene(4, seed=11, use_depth_residuals=True, use_descriptor_residuals=False)
    rng = np.random.default_rng(3)
    init = []
    for kf in scene.keyframes:
        x = np.concatenate([rng.choice([-1, 1], 3) * 0.004, rng.choice([-1, 1], 3) * 0.001]).astype(np.float32)
        init.append(bso.se3_mul(kf.global_T_frame, bso.se3_exp(x)))
    return scene, init


def _pose_array(poses):
    return np.array([[p.q[0], p.q[1], p.q[2], p.q[3], p.t[0], p.t[1], p.t[2]] for p in poses], np.float32)


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from badslam_amd.distributed import AllReduceHook, shard_range
    from tests import bso, gpu_util
    bso.build_oracle()
    scene, init = _scene()
    lo, hi = shard_range(scene.surfels_size, rank, world)
    shard = bso.HostScene(scene.color_camera, scene.depth_camera, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, max(1, hi - lo))
    shard.surfels = np.ascontiguousarray(scene.surfels[:, lo:hi])
    shard.active = np.ascontiguousarray(scene.active[:, lo:hi])
    shard.surfels_size = hi - lo
    shard.keyframes = scene.keyframes
    hip = gpu_util.Hip(shard.to_device("cuda:0"))
    hook = AllReduceHook(device=True)
    poses, iters, conv = hip.estimate_poses_batched(init, allreduce=hook.callback)
    assert hook.calls >= 2
    np.save(os.path.join(out_dir, f"poses_{rank}.npy"), _pose_array(poses))
    np.save(os.path.join(out_dir, f"iters_{rank}.npy"), np.array(iters))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharded_pose_estimation_matches_single_process(tmp_path, oracle):
    import torch.multiprocessing as mp
    from tests import gpu_util
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    p0, p1 = np.load(tmp_path / "poses_0.npy"), np.load(tmp_path / "poses_1.npy")
    assert np.array_equal(p0.view(np.uint32), p1.view(np.uint32)), "ranks must stay bit-identical without a broadcast"
    scene, init = _scene()
    hip = gpu_util.Hip(scene.to_device("cuda:0"))
    poses, iters, conv = hip.estimate_poses_batched(init)
    single = _pose_array(poses)
    assert all(conv)
    # same Gauss-Newton problem, different fp32 summation order: poses agree far below the 1e-4 bar
    assert np.abs(p0 - single).max() < 2e-6, np.abs(p0 - single).max()
    # and the optimisation did something: it moved the 4 mm / 1 mrad perturbed estimates back to the truth
    truth = _pose_array([kf.global_T_frame for kf in scene.keyframes])
    assert np.abs(single - truth).max() < 1e-4 < np.abs(_pose_array(init) - truth).max()

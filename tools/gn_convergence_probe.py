#!/usr/bin/env python3
"""Diagnostic: how the batched photometric pose Gauss-Newton behaves on the bench stack, as a function of how well the
descriptors are fitted (number of geometry iterations at the rendered poses before the pose problem is posed) and of the
iteration budget.  Prints, per setting, the median / max translation error against the rendered poses and the share of
keyframes that ended below the reference's step-norm threshold.  usage (GPU box): python tools/gn_convergence_probe.py [K]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import badslam_amd  # noqa: E402
from badslam_amd import abi, synthetic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
L = badslam_amd.lib()
ctx = badslam_amd.Context(0)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
dev = synthetic.TorchStack(K, "cuda:0")
cam, dp = dev.stack.camera, dev.depth_params()
sb, ab = dev.buf(dev.surfels), dev.buf(dev.active)
kfs = dev.keyframe_views()
S = dev.surfels_size
rng = np.random.default_rng(7)
inits = (abi.SE3f * K)()
for k in range(K):
    inits[k] = dev.stack.pose(k, np.concatenate([rng.choice([-1, 1], 3) * 0.005, rng.choice([-1, 1], 3) * 0.001]))[0]
truth = np.array([[*dev.stack.pose(k)[0].t] for k in range(K)])


def solve(cap):
    poses = (abi.SE3f * K)()
    C.memmove(poses, inits, C.sizeof(poses))
    iters, conv = (C.c_int32 * K)(), (C.c_int32 * K)()
    badslam_amd.check(L.bslam_estimate_frame_poses_batched(ctx.handle, stream, 1, 1, C.byref(cam), C.byref(cam), C.byref(dp), K, kfs, S, C.byref(sb), cap, poses, iters,
                                                           conv, C.cast(None, abi.ALLREDUCE_FN), None))
    err = np.abs(np.array([[*p.t] for p in poses]) - truth).max(axis=1)
    return err, np.array(iters), np.array(conv)


badslam_amd.check(L.bslam_update_surfel_activation(ctx.handle, stream, C.byref(cam), C.byref(dp), K, kfs, S, C.byref(sb), C.byref(ab)))
done = 0
for geo in (0, 1, 2, 4, 8):
    while done < geo:
        badslam_amd.check(L.bslam_optimize_geometry_iteration(ctx.handle, stream, 1, 1, C.byref(cam), C.byref(cam), C.byref(dp), K, kfs, S, C.byref(sb), C.byref(ab)))
        done += 1
    d = dev.surfels[5:7, :S].abs().mean().item()
    for cap in (5, 10, 20, 30, 60, 120):
        err, iters, conv = solve(cap)
        print(f"geometry iterations {geo}  mean |descriptor| {d:6.2f}  cap {cap:3d}:  error vs rendered pose median {np.median(err) * 1e3:6.3f} mm  max {err.max() * 1e3:6.3f} mm   "
              f"converged {conv.mean():.2f}  mean iterations {iters.mean():5.1f}", flush=True)

#!/usr/bin/env python3
"""bench.py -- full alternating-BA-iteration throughput of the HIP hot path on synthetic
640x480 keyframe stacks (BASELINE.json metric: surfel x KF residual evaluations / s).

A "step" is one alternating BA iteration over the whole stack (BS/direct_ba_alternating.cc:345-717):
surfel activation, geometry step (normals + position [+ descriptors]) and the batched pose
Gauss-Newton, every keyframe restarted from a 5 mm / 1 mrad perturbed pose.  A *pair* is one
(surfel, keyframe) visit that performs projection + association.

    python bench.py --gpus N --steps K --warmup W
N > 1: launched by torch.distributed.run, one rank per GPU; surfels are sharded (weak scaling:
every rank owns a full-size shard), keyframes replicated, the K x 32 coefficient rows are
all-reduced over RCCL once per batched GN iteration.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic bytes per pair, SURVEY.md 8(d)
B_REJECTED = 12
B_POSE = {False: 24, True: 48}       # geometry-only / photo+geo
B_ASSOC = 25
B_NORMALS = 25
B_POSITION = {False: 25, True: 49}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--keyframes", type=int, default=50)
    ap.add_argument("--photometric", type=int, default=0, help="1: photometric+geometric residuals (config 3 shape)")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  Native libraries write there as well (RCCL prints a version banner at
    # communicator creation), so file descriptor 1 points at stderr until the line is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import badslam_amd
    from badslam_amd import abi, synthetic
    from badslam_amd.distributed import AllReduceHook

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one rank per GPU.  (Rehearsal on a one-GPU box: BSLAM_BENCH_BACKEND=gloo lets several ranks share the card --
    # RCCL itself needs one GPU per rank.)
    backend = os.environ.get("BSLAM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    use_desc = bool(args.photometric)
    K = args.keyframes
    # every rank builds the same keyframe stack; rank r jitters its surfel shard differently (weak scaling)
    stack = synthetic.SyntheticStack(K, seed=0xBAD51A4)
    if world > 1:
        rng = np.random.default_rng(1000 + rank)
        stack.surfels[2] += rng.uniform(-0.001, 0.001, stack.surfels_size).astype(np.float32)
    dev = synthetic.DeviceStack(stack, device)
    S = dev.surfels_size

    L = badslam_amd.lib()
    ctx = badslam_amd.Context(dev_index)
    L.bslam_set_keyframe_cache(ctx.handle, 1)   # the bench never rewrites a keyframe image in place
    if os.environ.get("BSLAM_GEOM_KF_CHUNK"):   # tuning runs only; the default is the library's
        badslam_amd.check(L.bslam_set_geometry_keyframe_chunk(ctx.handle, int(os.environ["BSLAM_GEOM_KF_CHUNK"])))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dp = dev.depth_params()
    sb, ab = dev.buf(dev.surfels), dev.buf(dev.active)
    kfs = dev.keyframe_views()
    cam = stack.camera
    rng = np.random.default_rng(7)
    xis = [np.concatenate([rng.choice([-1, 1], 3) * 0.005, rng.choice([-1, 1], 3) * 0.001]) for _ in range(K)]
    init_poses = (abi.SE3f * K)()
    for k in range(K):
        init_poses[k] = stack.pose(k, xis[k])[0]
    hook = AllReduceHook(device=True) if world > 1 else None
    cb = hook.callback if hook else C.cast(None, abi.ALLREDUCE_FN)
    if world == 1 and os.environ.get("BSLAM_BENCH_SELF_RCCL"):
        # rehearsal of the N > 1 exchange on one GPU: a one-rank RCCL group, for which the all-reduce is the identity
        # but goes through the same torch.distributed / RCCL launch and stream hand-over (not a bench configuration)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(device))
        hook = AllReduceHook(device=True)
        cb = hook.callback
    if world == 1 and os.environ.get("BSLAM_BENCH_NOOP_HOOK"):
        # rehearsal of the N > 1 kernel sequence on one GPU: the exchange is a no-op callback (not a bench configuration)
        noop = abi.ALLREDUCE_FN(lambda user, ptr, count, stream: 0)
        cb = noop
    poses = (abi.SE3f * K)()
    iters = (C.c_int32 * K)()
    conv = (C.c_int32 * K)()

    def step():
        badslam_amd.check(L.bslam_update_surfel_activation(ctx.handle, stream, C.byref(cam), C.byref(dp), K, kfs, S, C.byref(sb), C.byref(ab)))
        badslam_amd.check(L.bslam_optimize_geometry_iteration(ctx.handle, stream, 1, int(use_desc), C.byref(cam), C.byref(cam), C.byref(dp),
                                                              K, kfs, S, C.byref(sb), C.byref(ab)))
        C.memmove(poses, init_poses, C.sizeof(poses))
        badslam_amd.check(L.bslam_estimate_frame_poses_batched(ctx.handle, stream, 1, int(use_desc), C.byref(cam), C.byref(cam), C.byref(dp),
                                                               K, kfs, S, C.byref(sb), 30, poses, iters, conv, cb, None))
        return sum(iters)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # census for the roofline accounting (untimed)
    inb, assoc = C.c_uint64(), C.c_uint64()
    badslam_amd.check(L.bslam_debug_count_pairs(ctx.handle, stream, C.byref(cam), C.byref(dp), K, kfs, S, C.byref(sb), C.byref(inb), C.byref(assoc)))
    pairs_per_pass = S * K
    frac_inb = inb.value / pairs_per_pass

    L.bslam_profile_enable(ctx.handle, 1)
    barrier()
    t0 = time.perf_counter()
    gn_iters_total = 0
    for _ in range(args.steps):
        gn_iters_total += step()
    barrier()
    dt = time.perf_counter() - t0
    launches, kms = C.c_int32(), C.c_float()
    L.bslam_profile_read(ctx.handle, 0, C.byref(launches), C.byref(kms))
    glaunches, gms = C.c_int32(), C.c_float()
    L.bslam_profile_read(ctx.handle, 1, C.byref(glaunches), C.byref(gms))
    L.bslam_profile_enable(ctx.handle, 0)

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # pairs per step on this rank: activation + normals + position passes over all K, plus one pass per
    # GN iteration of each keyframe (converged keyframes are skipped by the kernel)
    pairs_rank = args.steps * S * K * 3 + S * gn_iters_total
    pairs_total = pairs_rank * world   # weak scaling: every rank processes the same amount
    value = pairs_total / dt

    # roofline of the dominant kernel (pose_accumulate): algorithmic bytes per launch / avg launch time.
    # A launch covers the keyframes still unconverged; average active keyframes per launch:
    avg_kf_per_launch = gn_iters_total / max(1, launches.value)
    pairs_per_launch = S * avg_kf_per_launch
    bytes_per_launch = pairs_per_launch * (frac_inb * B_POSE[use_desc] + (1 - frac_inb) * B_REJECTED)
    avg_launch_s = (kms.value / 1e3) / max(1, launches.value)
    achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0

    out = {
        "metric": "surfel x keyframe residual evaluations per second, full alternating BA iteration",
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"synthetic 640x480 stack, {K} keyframes, {S} surfels per GPU (cell 4, no merge), "
                               f"{'photometric+geometric' if use_desc else 'geometry-residual-only'} alternating BA iteration "
                               f"(BASELINE.json configs[{2 if use_desc else 1}] shape)",
                   "keyframes": K, "surfels_per_gpu": S, "gn_iterations_per_step": gn_iters_total / args.steps,
                   "in_bounds_pair_fraction": frac_inb, "associated_pair_fraction": assoc.value / pairs_per_pass,
                   "parallelism": f"surfel-shard x{world}"},
        "roofline": {"bound": "hbm", "kernel": "pose_accumulate_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(use_desc, "pose_accumulate_kernel", K, S),
                     "avg_launch_us": avg_launch_s * 1e6, "launches": launches.value,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "geometry_kernel": geometry_roofline(S, K, frac_inb, assoc.value / pairs_per_pass, use_desc, glaunches.value, gms.value)},
    }

    if rank == 0 and args.cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(stack, K, use_desc, args.cpu_seconds)
    if dist.is_initialized():
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


def pmc_traffic(use_desc, kernel, K, S):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json, written
    from tools/pmc.sh runs of this same command; counters cannot be read from inside the timed process), or None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            j = json.load(f)
        w = j.get("workload", {})
        if (w.get("keyframes"), w.get("surfels_per_gpu")) != (K, S):
            return None   # the counters were collected on another workload
        return j["photo" if use_desc else "geo"].get(kernel)
    except (OSError, KeyError, ValueError):
        return None


def geometry_roofline(S, K, frac_inb, frac_assoc, use_desc, launches, total_ms):
    """Second kernel of the BA iteration: normals pass + position (or position+descriptor) pass per launch."""
    if launches == 0 or total_ms <= 0:
        return None
    pairs = S * K
    nbytes = pairs * (frac_inb * B_NORMALS + (1 - frac_inb) * B_REJECTED) + pairs * (frac_inb * B_POSITION[use_desc] + (1 - frac_inb) * B_REJECTED)
    avg_s = total_ms / 1e3 / launches
    ach = nbytes / avg_s / 1e9
    return {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "avg_launch_us": avg_s * 1e6, "launches": launches, "algorithmic_bytes_per_launch": nbytes,
            "traffic": pmc_traffic(use_desc, "geometry_kernel", K, S)}


def host_cpu_share():
    """CPUs this process may actually use: the cgroup quota if there is one (a GPU box hands out a share of its
    cores), else the affinity mask.  Oversubscribing the share only slows the baseline down."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(stack, K, use_desc, budget_s):
    """The oracle (kind "port": this repo's CPU restatement; the reference has no CPU cost
    evaluation) timed on the host cores on a bounded sample: whole pose passes
    (all surfels x all keyframes, OpenMP tasks over keyframe x surfel chunk on every host thread) of the same
    stack, repeated until ~budget_s have elapsed."""
    from badslam_amd import abi
    from tests import bso
    L = bso.lib()
    surf = stack.surfels
    sb = bso.np_buffer2d(surf)
    cf = bso.np_buffer2d(stack.cfactor)
    dp = abi.DepthParams(cf, 0.0, float(stack.raw_to_float_depth), stack.baseline_fx, stack.cell)
    kfs = (abi.KeyframeView * K)()
    for k in range(K):
        v = kfs[k]
        v.depth, v.normals = bso.np_buffer2d(stack.depth[k]), bso.np_buffer2d(stack.normals[k])
        v.radius, v.color = bso.np_buffer2d(stack.radius[k]), bso.np_buffer2d(stack.color[k])
        _, M, Rg = stack.pose(k)
        v.frame_T_global, v.global_R_frame, v.activation, v.id = M, Rg, 0, k
    Hb = np.zeros((K, 27), np.float32)
    counts = np.zeros(K, np.uint32)
    cores = host_cpu_share()
    passes, t0, used = 0, time.perf_counter(), 1
    while True:
        used = L.bso_bench_pose_pass(1, int(use_desc), C.byref(stack.camera), C.byref(stack.camera), C.byref(dp), K, kfs,
                                     stack.surfels_size, C.byref(sb), 0, Hb.ctypes.data_as(C.POINTER(C.c_float)),
                                     counts.ctypes.data_as(C.POINTER(C.c_uint32)), cores)
        passes += 1
        el = time.perf_counter() - t0
        if el > budget_s or passes >= 50:
            break
    pairs = passes * stack.surfels_size * K
    return {"value": pairs / el, "unit": "pairs/s", "cores": int(used), "kind": "port",
            "sample": f"{passes} pose-coefficient passes over {stack.surfels_size} surfels x {K} keyframes "
                      f"({el:.1f} s, OpenMP over keyframe x surfel-chunk tasks)"}


if __name__ == "__main__":
    main()

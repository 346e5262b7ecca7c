#!/usr/bin/env python3
"""bench.py -- full alternating-BA-iteration throughput of the HIP hot path on synthetic
640x480 keyframe stacks (BASELINE.json metric: surfel x KF residual evaluations / s).

Headline workload (the default): BASELINE.json configs[2] -- full photometric + geometric BA, 300 keyframes,
5.76 M surfels on one MI355X.  A "step" is one alternating BA iteration over the whole stack
(BS/direct_ba_alternating.cc:345-717): surfel activation, geometry step (normals + position + descriptors) and the
batched pose Gauss-Newton, every keyframe restarted from a 5 mm / 1 mrad perturbed pose.  A *pair* is one
(surfel, keyframe) visit that performs projection + association; only pairs that were actually
evaluated are counted (the activation pass stops at a surfel's first associated keyframe, the geometry passes skip
inactive surfels, converged keyframes leave the pose loop).

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU.  Started under a launcher (torch.distributed.run: WORLD_SIZE / RANK / LOCAL_RANK in the environment)
this process IS one rank; started bare, it spawns the N ranks itself as fresh child processes -- before anything in the parent
touches a GPU -- relays rank 0's JSON line and fails if any rank fails.  Surfels are sharded, keyframes replicated, the K x 32
coefficient rows are all-reduced over RCCL once per batched GN iteration.  `value` is the weak-scaled configs[2] workload
(every rank owns a full 5.76 M-surfel shard; N = 1 is the single-GPU headline); `config.strong` is the FIXED configs[4]
problem (1000 keyframes x 19.2 M surfels, geometry only) split over the N ranks with shard_range.  On one GPU the line also
carries `config.secondary` (configs[1]: 50 keyframes, geometry only), `config.trajectory` and `config.survey` (the other two
stacks of SURVEY.md 8d), a `pcg` block (PCG scheme) and `cpu_baseline` (the oracle on the host cores, N threads and 1 thread).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import socket
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
SIMD_COUNT = 1024       # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9
# algorithmic bytes per pair, SURVEY.md 8(d)
B_REJECTED = 12
B_POSE = {False: 24, True: 48}       # geometry-only / photo+geo
B_ASSOC = 25
B_NORMALS = 25
B_POSITION = {False: 25, True: 49}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--keyframes", type=int, default=300)
    ap.add_argument("--photometric", type=int, default=1, help="1: photometric+geometric residuals (configs[2]); 0: geometry only")
    ap.add_argument("--secondary", type=int, default=1, help="also measure configs[1] (50 keyframes, geometry only) on one GPU")
    ap.add_argument("--pcg", type=int, default=1, help="also measure the PCG scheme (one GPU)")
    ap.add_argument("--scene", default="dense", choices=["dense", "survey", "trajectory"],
                    help="synthetic stack of the headline block (badslam_amd/synthetic.py); the bench line of record uses dense")
    ap.add_argument("--trajectory", type=int, default=1, help="also measure the smooth-trajectory stack (one GPU): config.trajectory")
    ap.add_argument("--survey", type=int, default=1, help="also measure the dense stack with the survey's pose ranges (one GPU): config.survey")
    ap.add_argument("--strong", type=int, default=1, help="also measure the fixed configs[4] problem split over the ranks: config.strong")
    ap.add_argument("--strong-keyframes", type=int, default=1000)
    ap.add_argument("--culling", type=int, default=1, help="block-level frustum culling in the pair kernels (0: off, A/B runs)")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target seconds for each of the two CPU-baseline runs")
    return ap.parse_args()


class Env:
    pass


WORKLOAD_NAMES = {
    "dense": "synthetic 640x480 stack",
    "survey": "synthetic 640x480 stack with the pose ranges of SURVEY.md 8(d) (xi_t +-1.5 m, xi_r +-0.7 rad)",
    "trajectory": "synthetic 640x480 smooth-trajectory stack (one lap through a room, SURVEY.md 8d)",
}


def build_stack(env, K, kind, rendered_in_hbm, shard=None):
    """The workload's keyframe stack and this rank's surfels in HBM.  Returns (device stack, host stack or None).
    rendered_in_hbm = False: numpy on the host (the headline of record, kept bit-identical since round 1, and what the CPU
    baseline reads); True: torch ops on the device (seconds instead of minutes at K = 300 / 1000).
    shard = (rank, world): this rank keeps the surfel columns shard_range(S, rank, world) of the FIXED problem (strong scaling);
    None with world > 1: every rank keeps a full-size shard with its own jitter (weak scaling)."""
    import torch
    from badslam_amd import synthetic
    from badslam_amd.distributed import shard_range
    host = None
    if rendered_in_hbm:
        dev = synthetic.TorchStack(K, env.device, kind=kind)
    else:
        host = synthetic.SyntheticStack(K, seed=0xBAD51A4, kind=kind)
        if env.world > 1 and shard is None:   # weak scaling: rank r jitters its full-size surfel shard differently
            rng = np.random.default_rng(1000 + env.rank)
            host.surfels[2] += rng.uniform(-0.001, 0.001, host.surfels_size).astype(np.float32)
        dev = synthetic.DeviceStack(host, env.device)
    dev.total_surfels = dev.surfels_size
    if shard is not None:
        lo, hi = shard_range(dev.surfels_size, shard[0], shard[1])
        dev.surfels = dev.surfels[:, lo:hi].contiguous()
        dev.surfels_size = hi - lo
        dev.active = torch.ones((1, max(1, dev.surfels_size)), dtype=torch.uint8, device=env.device)
    return dev, host


def measure(env, K, use_desc, steps, warmup, kind="dense", rendered_in_hbm=False, shard=None, config_index=None):
    """Times `steps` BA iterations on a synthetic stack of K keyframes; returns the result dict of that workload."""
    import torch
    import torch.distributed as dist
    import badslam_amd
    from badslam_amd import abi
    L, ctx, world, rank = env.L, env.ctx, env.world, env.rank
    dev, host = build_stack(env, K, kind, rendered_in_hbm, shard)
    stack = dev.stack
    S = dev.surfels_size
    badslam_amd.check(L.bslam_invalidate_keyframe_cache(ctx.handle))
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
    poses = (abi.SE3f * K)()
    iters = (C.c_int32 * K)()
    conv = (C.c_int32 * K)()
    hist = {}
    # Photometric workloads: every step solves the SAME problem -- the persistent surfel rows are restored from a pristine copy at
    # the start of a step (a device-to-device copy of 8 rows on the BA stream, inside the timed region: 0.1 ms at 5.76 M surfels).
    # Their Gauss-Newton loops run into the reference's cap of 30, so a surfel state that drifted over the steps would change the
    # work per step.  The geometry-only loops converge in 2 - 3 iterations whatever the state: no restore there.
    restore = bool(use_desc)
    pristine = dev.surfels[:8].clone() if restore else None

    def step():
        if restore:
            dev.surfels[:8].copy_(pristine)
        badslam_amd.check(L.bslam_update_surfel_activation(ctx.handle, stream, C.byref(cam), C.byref(dp), K, kfs, S, C.byref(sb), C.byref(ab)))
        badslam_amd.check(L.bslam_optimize_geometry_iteration(ctx.handle, stream, 1, int(use_desc), C.byref(cam), C.byref(cam), C.byref(dp),
                                                              K, kfs, S, C.byref(sb), C.byref(ab)))
        C.memmove(poses, init_poses, C.sizeof(poses))
        badslam_amd.check(L.bslam_estimate_frame_poses_batched(ctx.handle, stream, 1, int(use_desc), C.byref(cam), C.byref(cam), C.byref(dp),
                                                               K, kfs, S, C.byref(sb), 30, poses, iters, conv, env.cb, None))
        return list(iters), sum(conv)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    # census for the roofline accounting (untimed), on the surfels as the pose loop sees them
    inb, assoc = C.c_uint64(), C.c_uint64()
    badslam_amd.check(L.bslam_debug_count_pairs(ctx.handle, stream, C.byref(cam), C.byref(dp), K, kfs, S, C.byref(sb), C.byref(inb), C.byref(assoc)))
    frac_inb = inb.value / max(1, S * K)
    frac_assoc = assoc.value / max(1, S * K)

    badslam_amd.check(L.bslam_profile_enable(ctx.handle, 1))
    barrier()
    t0 = time.perf_counter()
    gn_iters_total = converged_total = 0
    for _ in range(steps):
        its, nconv = step()
        gn_iters_total += sum(its)
        converged_total += nconv
        for v in its:
            hist[v] = hist.get(v, 0) + 1
    barrier()
    dt = time.perf_counter() - t0
    prof = {}
    for tag, name in ((0, "pose"), (1, "geometry"), (4, "activation"), (5, "exchange"), (6, "pose_reduce")):
        launches, kms = C.c_int32(), C.c_float()
        badslam_amd.check(L.bslam_profile_read(ctx.handle, tag, C.byref(launches), C.byref(kms)))
        prof[name] = (launches.value, kms.value)
    counters = (C.c_uint64 * 8)()
    badslam_amd.check(L.bslam_profile_read_counters(ctx.handle, counters))
    badslam_amd.check(L.bslam_profile_enable(ctx.handle, 0))

    # pairs COVERED on this rank over the timed steps: the activation pass's visits (it stops at a surfel's first associated
    # keyframe), the two geometry passes over the active surfels, one pass per GN iteration of each unconverged keyframe.
    # The reference spends a thread on every one of them (BS/kernel_opt_pose.cu:263-275); here a (work slot, keyframe) pair
    # whose surfels provably cannot project into the image is decided by one block-level test (`culled_pair_fraction`).
    pairs_activation = int(counters[0])                 # visited until the first associated keyframe
    active_surfel_steps = int(counters[1])              # sum over steps of the surfels set active
    pairs_geometry = 2 * K * active_surfel_steps        # normals pass + position(/descriptor) pass, active surfels only
    pairs_pose = S * gn_iters_total                     # one pass per GN iteration of each unconverged keyframe
    pairs_rank = pairs_activation + pairs_geometry + pairs_pose
    pairs_all = pairs_rank * world                      # weak scaling: every rank processes the same amount
    if world > 1:
        red = torch.tensor([dt], dtype=torch.float64, device=env.control_device)
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        dt = float(red.item())
        if shard is not None:                           # strong scaling: the ranks' shares differ by a surfel or two
            tot = torch.tensor([float(pairs_rank)], dtype=torch.float64, device=env.control_device)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            pairs_all = float(tot.item())
    value = pairs_all / dt

    pose_launches, pose_ms = prof["pose"]
    # dominant kernel: pose_accumulate.  A launch covers the keyframes still unconverged.
    avg_kf_per_launch = gn_iters_total / max(1, pose_launches)
    bytes_per_launch = S * avg_kf_per_launch * (frac_inb * B_POSE[use_desc] + (1 - frac_inb) * B_REJECTED)
    avg_launch_s = (pose_ms / 1e3) / max(1, pose_launches)
    achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    pmc = pmc_entry(use_desc, K, S, kind)
    roof = {"bound": "valu", "kernel": "pose_accumulate_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_value(pmc, "pose_accumulate_kernel", "hbm_bytes"),
            "convention": "achieved = ALGORITHMIC bytes per launch (SURVEY.md 8d: in-bounds pairs at the full figure, pairs rejected at "
                          "z <= 0 / bounds at 12 B) / average launch time, priced against the HBM peak as the contract asks; the PMC "
                          "counters say the kernel is VALU-issue bound, not HBM bound (valu_issue_frac, hbm_traffic_frac)",
            "avg_launch_us": avg_launch_s * 1e6, "launches": pose_launches, "keyframes_per_launch": avg_kf_per_launch,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "pmc_source": (pmc or {}).get("source"), "pmc_csrc_hash": (pmc or {}).get("csrc_hash")}
    valu = pmc_value(pmc, "pose_accumulate_kernel", "valu_wave_insts_per_pair")
    if valu is not None and avg_launch_s > 0:
        # SQ_INSTS_VALU counts wave instructions; a wave64 op occupies a SIMD32 for >= 2 cycles (packed / fp64 / DPP more)
        roof["valu_issue_frac"] = valu * S * avg_kf_per_launch * 2 / (SIMD_COUNT * CLOCK_HZ * avg_launch_s)
    if roof["traffic"] is not None and avg_launch_s > 0:
        # counted per average launch of the PMC run: scale by pairs to this run's average launch
        per_pair = roof["traffic"] / (pmc_value(pmc, "pose_accumulate_kernel", "pairs_per_launch") or (S * K))
        roof["traffic"] = per_pair * S * avg_kf_per_launch
        roof["traffic_raw"] = (pmc_value(pmc, "pose_accumulate_kernel", "hbm_bytes_raw") or 0) / (pmc_value(pmc, "pose_accumulate_kernel", "pairs_per_launch") or (S * K)) * S * avg_kf_per_launch
        roof["hbm_traffic_frac"] = roof["traffic"] / avg_launch_s / 1e9 / HBM_PEAK_GBS
    geo_launches, geo_ms = prof["geometry"]
    roof["geometry_kernel"] = geometry_roofline(S, K, frac_inb, use_desc, steps, active_surfel_steps, geo_ms, pmc)
    act_launches, act_ms = prof["activation"]
    if act_launches:
        roof["activation_kernel"] = {"avg_launch_us": act_ms * 1e3 / act_launches, "pairs_visited_per_launch": pairs_activation / act_launches,
                                     "achieved": pairs_activation / act_launches * (frac_inb * B_ASSOC + (1 - frac_inb) * B_REJECTED) / (act_ms / 1e3 / act_launches) / 1e9}
    red_launches, red_ms = prof["pose_reduce"]
    if red_launches:
        roof["pose_reduce_solve_kernel"] = {"avg_launch_us": red_ms * 1e3 / red_launches, "launches": red_launches}
    exchange = None
    ex_launches, ex_ms = prof["exchange"]
    if ex_launches:
        exchange = {"allreduce_us_per_gn_iteration": ex_ms * 1e3 / ex_launches, "allreduces": ex_launches, "floats_per_allreduce": K * 32,
                    "note": "HIP events on the BA stream around the in-place K x 32 float sum of this rank"}
    cull = cull_stats(env)
    label = WORKLOAD_NAMES[kind]
    if shard is not None:
        per_gpu = f"{dev.total_surfels} surfels split over {world} GPU(s) ({S} on this rank)"
    else:
        per_gpu = f"{S} surfels per GPU"
    cfg_name = f" (BASELINE.json configs[{config_index}])" if config_index is not None else ""
    return {
        "value": value, "ms_per_step": dt / steps * 1e3, "S": S, "K": K, "stack": host, "dev": dev,
        "config": {"workload": f"{label}, {K} keyframes, {per_gpu} (cell 4, no merge), "
                               f"{'photometric+geometric' if use_desc else 'geometry-residual-only'} alternating BA iteration{cfg_name}",
                   "scene": kind, "keyframes": K, "surfels_per_gpu": S, "surfels_total": dev.total_surfels * (world if shard is None else 1),
                   "gn_iterations_per_step": gn_iters_total / steps,
                   "gn_iteration_histogram": {str(k): v for k, v in sorted(hist.items())},
                   "keyframes_converged_fraction": converged_total / (steps * K),
                   "gn_convergence_note": ("photometric loops converge slowly (about 37 iterations on average on the dense stack) to a fixed point a few mm beside the "
                                           "rendered pose -- the reference's descriptor Jacobian is approximate by construction (BS/cost_function.cuh:188-190) -- so many "
                                           "keyframes end at the reference's cap of 30 (BS/direct_ba_alternating.cc:130); the oracle's sequential loop does the same on the "
                                           "same data: tests/test_gpu_large_configs.py::test_photometric_gauss_newton_on_the_bench_stack") if use_desc else None,
                   "pairs_per_step": {"activation_visited": pairs_activation / steps, "geometry": pairs_geometry / steps, "pose": pairs_pose / steps},
                   "active_surfel_fraction": active_surfel_steps / max(1, steps * S),
                   "in_bounds_pair_fraction": frac_inb, "associated_pair_fraction": frac_assoc,
                   "culled_pair_fraction": cull,
                   "surfels_restored_per_step": restore,
                   "parallelism": f"surfel-shard x{world}", "exchange": env.exchange, "exchange_timing": exchange},
        "roofline": roof,
    }


def cull_stats(env):
    """Fraction of the (work slot, keyframe) pairs of the last pose launch that the block-level frustum test rejected, or None
    when the library was not built with the counter / culling is off."""
    L = env.L
    if not hasattr(L, "bslam_debug_cull_stats"):
        return None
    tested, culled = C.c_uint64(), C.c_uint64()
    if L.bslam_debug_cull_stats(env.ctx.handle, C.byref(tested), C.byref(culled)) != 0 or tested.value == 0:
        return None
    return culled.value / tested.value


# ---------------------------------------------------------------------------------------------------------------------
# Launching.  `python bench.py --gpus N` with no launcher in the environment starts its own N ranks.
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpu_count():
    """Counting devices does not initialise the GPU in this process (torch.cuda.device_count)."""
    import torch
    return torch.cuda.device_count()


def spawn_ranks(args, argv):
    """Parent of a bare `bench.py --gpus N`: N fresh child processes, one rank each (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, exactly what torch.distributed.run would set).  The parent never touches a GPU; it relays
    rank 0's JSON line and exits non-zero if any rank does, or if the line does not report N ranks."""
    n = args.gpus
    backend = os.environ.get("BSLAM_BENCH_BACKEND", "native")
    dry = bool(os.environ.get("BSLAM_BENCH_DRY_RUN"))
    if backend != "gloo" and not dry:
        have = visible_gpu_count()
        if have < n:
            raise SystemExit(f"bench.py: --gpus {n} needs {n} GPUs, {have} visible (one rank per GPU; BSLAM_BENCH_BACKEND=gloo "
                             f"rehearses N ranks on fewer cards)")
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BSLAM_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # rank 0's stdout is read to the end first (a few hundred bytes); the others are then collected.  A rank that dies early
    # takes the group down: the survivors' barriers time out inside torch.distributed, so they are stopped here by PID.
    failed = None
    out0 = ""
    pending = set(range(n))
    import selectors
    sel = selectors.DefaultSelector()
    sel.register(procs[0].stdout, selectors.EVENT_READ)
    stdout_open = True
    while pending:
        if stdout_open:
            for key, _ in sel.select(timeout=0.2):
                chunk = key.fileobj.readline()
                if chunk == "":
                    sel.unregister(key.fileobj)
                    stdout_open = False
                else:
                    out0 += chunk
        else:
            time.sleep(0.2)
        for r in list(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and failed is None:
                failed = (r, rc)
        if failed is not None:
            break
    if failed is not None:
        for r in pending:
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        raise SystemExit(f"bench.py: rank {failed[0]} of {n} exited with code {failed[1]}; no result")
    if stdout_open:
        out0 += procs[0].stdout.read()
    line = None
    for cand in out0.splitlines():
        if cand.startswith("{"):
            line = cand
    if line is None:
        raise SystemExit("bench.py: rank 0 printed no JSON line")
    got = json.loads(line).get("n_gpus")
    if got != n:
        raise SystemExit(f"bench.py: asked for {n} ranks, the result line reports n_gpus = {got}")
    print(line, flush=True)


def dry_run_rank(args):
    """BSLAM_BENCH_DRY_RUN=1: the launch path without a GPU (tests/test_bench_launch_cpu.py) -- a gloo group, one sum across
    the ranks, and a line whose n_gpus is the size of the group that actually formed."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("BSLAM_BENCH_DRY_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    joined = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        joined = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "dry run of the launch path", "value": 0.0, "n_gpus": joined, "dry_run": True}), flush=True)


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        return spawn_ranks(args, sys.argv[1:])          # nothing in this process has touched a GPU
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it bare (it spawns its own ranks) or under "
                         f"a launcher with --nproc-per-node {args.gpus}")
    if os.environ.get("BSLAM_BENCH_DRY_RUN"):
        return dry_run_rank(args)
    run_rank(args, world)


def run_rank(args, world):
    # The contract is ONE JSON line on stdout.  Native libraries write there as well (RCCL prints a version banner at
    # communicator creation), so file descriptor 1 points at stderr until the line is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import badslam_amd
    from badslam_amd import abi
    from badslam_amd.distributed import AllReduceHook

    env = Env()
    env.rank = rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env.world = world
    # One rank per GPU.  Default exchange ("native"): the library's own RCCL communicator (bslam_comm_init) -- the K x 32
    # coefficient rows are summed by ncclAllReduce on the BA stream, no callback; torch.distributed (gloo) only carries the
    # 128-byte communicator id, the barriers and the max over ranks of the wall time.  BSLAM_BENCH_BACKEND=nccl | gloo selects the
    # older callback path through torch.distributed.all_reduce instead (gloo: rehearsal of N ranks sharing one card).
    backend = os.environ.get("BSLAM_BENCH_BACKEND", "native")
    cards = torch.cuda.device_count()
    if backend != "gloo" and world > cards:
        raise SystemExit(f"bench.py: rank {rank}: {world} ranks but {cards} GPU(s) visible (one rank per GPU)")
    dev_index = local_rank if backend != "gloo" else local_rank % max(1, cards)
    torch.cuda.set_device(dev_index)
    env.device = device = f"cuda:{dev_index}"
    env.control_device = device if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    env.L = L = badslam_amd.lib()
    env.ctx = ctx = badslam_amd.Context(dev_index)
    L.bslam_set_keyframe_cache(ctx.handle, 1)   # the bench never rewrites a keyframe image in place
    if hasattr(L, "bslam_set_culling"):
        badslam_amd.check(L.bslam_set_culling(ctx.handle, int(args.culling)))
    if os.environ.get("BSLAM_GEOM_KF_CHUNK"):   # tuning runs only; the default is the library's
        badslam_amd.check(L.bslam_set_geometry_keyframe_chunk(ctx.handle, int(os.environ["BSLAM_GEOM_KF_CHUNK"])))
    hook = None
    env.cb = C.cast(None, abi.ALLREDUCE_FN)
    env.exchange = "none (1 rank)"
    joined = 1                                    # ranks of the group the exchange actually runs over
    if world > 1 and backend == "native":
        ok = 1
        try:
            uid = [badslam_amd.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(uid[0], rank, world)
        except Exception as e:   # reported, and agreed upon by all ranks below
            ok = 0
            print(f"rank {rank}: bslam_comm_init failed: {e!r}", file=sys.stderr, flush=True)
        agreed = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        # every rank reaches this point and creates the fallback group together, used or not (collective call)
        fallback_group = dist.new_group(backend="nccl") if int(agreed.item()) != 1 else None
        if int(agreed.item()) == 1:
            joined = ctx.comm_query()[1]          # ncclCommCount of the library's communicator
            env.exchange = f"{joined} ranks: RCCL ncclAllReduce inside libbadslam_hip (bslam_comm_init), in place on the BA stream"
        else:
            # some rank could not create the library's communicator: every rank falls back to the callback path over
            # torch.distributed's RCCL group, and the line says so (config.exchange)
            if ok:
                ctx.comm_destroy()
            hook = AllReduceHook(device=True, group=fallback_group)
            env.cb = hook.callback
            joined = dist.get_world_size(fallback_group)
            env.exchange = f"{joined} ranks: torch.distributed.all_reduce callback (nccl) -- FALLBACK: bslam_comm_init failed on some rank"
    elif world > 1:
        hook = AllReduceHook(device=True)
        env.cb = hook.callback
        joined = dist.get_world_size()
        env.exchange = f"{joined} ranks: torch.distributed.all_reduce callback ({backend})" + (", ranks share cards (rehearsal)" if backend == "gloo" and cards < world else "")
    if world == 1 and os.environ.get("BSLAM_BENCH_SELF_RCCL"):
        # rehearsal of the N > 1 exchange on one GPU: a one-rank communicator, for which the all-reduce is the identity but
        # goes through the same kernel sequence and RCCL launch (not a bench configuration)
        ctx.comm_init(badslam_amd.comm_unique_id(), 0, 1)
        env.exchange = "1 rank: RCCL, one-rank rehearsal"
    if world == 1 and os.environ.get("BSLAM_BENCH_NOOP_HOOK"):
        # rehearsal of the N > 1 kernel sequence on one GPU: the exchange is a no-op callback (not a bench configuration)
        noop = abi.ALLREDUCE_FN(lambda user, ptr, count, stream: 0)
        env.cb = noop
    if joined != args.gpus:
        raise SystemExit(f"bench.py: rank {rank}: --gpus {args.gpus} but the exchange group has {joined} rank(s)")

    use_desc = bool(args.photometric)
    only = os.environ.get("BSLAM_BENCH_ONLY")     # profiling runs: "strong" = nothing but the strong-scaling block
    default_workload = args.keyframes == 300 and use_desc and args.scene == "dense"
    head = measure(env, args.keyframes, use_desc, args.steps, args.warmup, kind=args.scene, rendered_in_hbm=(args.scene != "dense"),
                   config_index=(2 if use_desc else 1) if args.scene == "dense" else None) if only != "strong" else None
    out = {"metric": "surfel x keyframe residual evaluations per second, full alternating BA iteration"}
    if head is not None:
        out.update({"value": head["value"], "unit": "pairs/s", "n_gpus": joined, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "f32", "data": "synthetic", "config": head["config"], "roofline": head["roofline"]})
        head["dev"] = None                        # the stack's HBM is released before the next block is built
    single = world == 1 and rank == 0
    small_stack = None
    headline_stack = head["stack"] if head is not None else None
    if single and head is not None and args.secondary and not (args.keyframes == 50 and not use_desc):
        sec = measure(env, 50, False, 20, 3, config_index=1)
        small_stack = sec["stack"]
        out["config"]["secondary"] = {"value": sec["value"], "unit": "pairs/s", "ms_per_step": sec["ms_per_step"], "steps": 20, "warmup": 3,
                                      "config": sec["config"], "roofline": sec["roofline"]}
        del sec
    # the other two stacks of SURVEY.md 8(d): same K, same residuals, rendered in HBM
    for flag, kind in ((args.trajectory, "trajectory"), (args.survey, "survey")):
        if single and head is not None and flag and default_workload:
            torch.cuda.empty_cache()
            r = measure(env, 300, True, 5, 2, kind=kind, rendered_in_hbm=True)
            out["config"][kind] = {"value": r["value"], "unit": "pairs/s", "ms_per_step": r["ms_per_step"], "steps": 5, "warmup": 2,
                                   "config": r["config"], "roofline": r["roofline"]}
            del r
    if args.strong and (default_workload or only == "strong"):
        # BASELINE.json configs[4]: ONE fixed problem (1000 keyframes x 19.2 M surfels, geometry only) split over the ranks
        torch.cuda.empty_cache()
        r = measure(env, args.strong_keyframes, False, 3, 1, kind="dense", rendered_in_hbm=True, shard=(rank, world), config_index=4)
        strong = {"value": r["value"], "unit": "pairs/s", "ms_per_step": r["ms_per_step"], "steps": 3, "warmup": 1, "scaling": "strong",
                  "n_gpus": joined, "config": r["config"], "roofline": r["roofline"]}
        del r
        if head is None:
            out.update({"value": strong["value"], "unit": "pairs/s", "n_gpus": joined, "steps": 3, "warmup": 1, "ms_per_step": strong["ms_per_step"],
                        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                        "config": strong["config"], "roofline": strong["roofline"]})
        else:
            out["config"]["strong"] = strong
    torch.cuda.empty_cache()
    if single and args.pcg and headline_stack is not None:
        out["pcg"] = {"headline_stack": pcg_block(headline_stack, use_desc, dev_index, out["config"]["in_bounds_pair_fraction"],
                                                  pmc=pmc_entry(use_desc, head["K"], head["S"], args.scene))}
        if small_stack is not None:
            out["pcg"]["secondary_stack"] = pcg_block(small_stack, False, dev_index, out["config"]["secondary"]["config"]["in_bounds_pair_fraction"],
                                                      pmc=pmc_entry(False, 50, small_stack.surfels_size, "dense"))
    if single and args.cpu_baseline and headline_stack is not None:
        out["cpu_baseline"] = cpu_baseline(headline_stack, head["K"], use_desc, args.cpu_seconds)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


def csrc_hash():
    """Hash of the kernel sources: PMC figures in profiles/pmc_traffic.json are only quoted for the code they were taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "badslam_amd", "csrc")
    for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_entry(use_desc, K, S, kind="dense"):
    """Per-kernel PMC figures of this exact workload from the committed counter passes (profiles/pmc_traffic.json, written from
    tools/pmc.sh runs of this same command; counters cannot be read from inside the timed process), or None.  An entry is
    stamped with the hash of csrc/ it was measured on: after a kernel change the stale counters are NOT quoted."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            j = json.load(f)
        for w in j.get("workloads", []):
            if (w.get("keyframes"), w.get("surfels_per_gpu"), bool(w.get("photometric")), w.get("scene", "dense")) == (K, S, bool(use_desc), kind):
                if w.get("csrc_hash") != csrc_hash():
                    print(f"bench.py: profiles/pmc_traffic.json entry {w.get('source')!r} was measured on csrc {w.get('csrc_hash')}, "
                          f"the tree is {csrc_hash()}: counters not quoted (rerun tools/pmc.sh + tools/pmc_traffic.py)", file=sys.stderr)
                    return None
                return w
    except (OSError, KeyError, ValueError):
        pass
    return None


def pmc_value(entry, kernel, key):
    if not entry:
        return None
    return entry.get("kernels", {}).get(kernel, {}).get(key)


def geometry_roofline(S, K, frac_inb, use_desc, steps, active_surfel_steps, total_ms, pmc):
    """Second kernel of the BA iteration: normals pass + position (or position+descriptor) pass per step, over the active
    surfels (one or several launches per step: resident grids x keyframe chunks)."""
    if total_ms <= 0 or steps == 0:
        return None
    pairs = (active_surfel_steps / steps) * K
    nbytes = pairs * (frac_inb * B_NORMALS + (1 - frac_inb) * B_REJECTED) + pairs * (frac_inb * B_POSITION[use_desc] + (1 - frac_inb) * B_REJECTED)
    per_step_s = total_ms / 1e3 / steps
    ach = nbytes / per_step_s / 1e9
    # same convention as the dominant kernel: ALGORITHMIC gather bytes against the HBM peak.  With the per-surfel work order most
    # of those gathers are served by L1 / L2 (see traffic), so the fraction can pass 1: it is not a physical utilisation.
    return {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "us_per_step": per_step_s * 1e6, "algorithmic_bytes_per_step": nbytes,
            "traffic": pmc_value(pmc, "geometry_kernel", "hbm_bytes"), "traffic_raw": pmc_value(pmc, "geometry_kernel", "hbm_bytes_raw")}


def pcg_block(stack, use_desc, dev_index, frac_inb, iterations=2, pmc=None):
    """PCG scheme (BS/direct_ba_pcg.cc:229-471) on the same stack through the C++ host class: poses + geometry, one outer
    iteration per call.  Pairs: the normals pass, PCGInit and one PCGStep1 pass per inner step, each over all S x K pairs.
    frac_inb: the fraction of pairs passing z > 0 / bounds on this stack (census of the alternating block), which prices the
    kernels' algorithmic bytes as SURVEY.md 8(d) says (rejected pairs stop after 12 B)."""
    import torch
    import badslam_amd
    from badslam_amd.direct_ba import DirectBA
    L = badslam_amd.lib()
    K, S = stack.K, stack.surfels_size
    cam = stack.camera
    ba = DirectBA(S, float(stack.raw_to_float_depth), stack.baseline_fx, stack.cell, 0.8, 1, 1, 1, cam, cam, 0, True, use_desc, device=dev_index)
    ba.set_options(pcg_gauge_keyframe=0)
    rng = np.random.default_rng(7)
    for k in range(K):
        xi = np.concatenate([rng.choice([-1, 1], 3) * 0.005, rng.choice([-1, 1], 3) * 0.001])
        T = stack.pose(k, xi if k else None)[0]
        ba.AddKeyframe(k, 0.3, 6.0, stack.depth[k], stack.normals[k], stack.radius[k], stack.color[k], T)
    ba.SetSurfels(stack.surfels[:8], S)
    h = ba.context_handle()
    ba.BundleAdjustment(False, False, False, True, True, 1, 1, True, 0, K - 1, True)   # warm-up
    torch.cuda.synchronize()
    badslam_amd.check(L.bslam_profile_enable(h, 1))
    t0 = time.perf_counter()
    for _ in range(iterations):
        ba.BundleAdjustment(False, False, False, True, True, 1, 1, True, 0, K - 1, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iterations
    res = {}
    for tag, name in ((2, "pcg_init_kernel"), (3, "pcg_step1_kernel")):
        launches, kms = C.c_int32(), C.c_float()
        badslam_amd.check(L.bslam_profile_read(h, tag, C.byref(launches), C.byref(kms)))
        res[name] = (launches.value, kms.value)
    badslam_amd.check(L.bslam_profile_enable(h, 0))
    ba.close()
    step1_launches, step1_ms = res["pcg_step1_kernel"]
    init_launches, init_ms = res["pcg_init_kernel"]
    inner = step1_launches / iterations
    pairs = S * K * (1 + 1 + inner)                       # normals + init + step1 passes per outer iteration
    out = {"keyframes": K, "surfels": S, "photometric": bool(use_desc), "ms_per_ba_iteration": dt * 1e3, "inner_steps_per_iteration": inner,
           "value": pairs / dt, "unit": "pairs/s"}
    if step1_launches:
        nbytes = S * K * (frac_inb * B_POSE[use_desc] + (1 - frac_inb) * B_REJECTED)
        avg = step1_ms / 1e3 / step1_launches
        out["pcg_step1_kernel"] = {"avg_launch_us": avg * 1e6, "launches": step1_launches, "achieved": nbytes / avg / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": nbytes / avg / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes,
                                   "in_bounds_pair_fraction": frac_inb, "traffic": pmc_value(pmc, "pcg_step1_kernel", "hbm_bytes"),
                                   "valu_wave_insts_per_pair": pmc_value(pmc, "pcg_step1_kernel", "valu_wave_insts_per_pair"),
                                   "note": "in-bounds pairs at the pose / PCG figure of SURVEY.md 8(d), rejected pairs at 12 B (census of the alternating block on the same stack)"}
    if init_launches:
        nbytes = S * K * (frac_inb * B_POSE[use_desc] + (1 - frac_inb) * B_REJECTED)
        avg = init_ms / 1e3 / init_launches
        out["pcg_init_kernel"] = {"avg_launch_us": avg * 1e6, "launches": init_launches, "achieved": nbytes / avg / 1e9,
                                  "frac": nbytes / avg / 1e9 / HBM_PEAK_GBS}
    return out


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
    """The oracle (kind "port": this repo's CPU restatement; the reference has no CPU cost evaluation)
    timed on the host cores on a bounded sample of the SAME workload: a strided subset of the stack's surfels goes through
    one full BA iteration's passes -- activation, geometry iteration, one pose-coefficient pass over all keyframes
    (oracle/bso_bench.c) -- once on all host threads (OpenMP over surfel ranges) and once on one thread."""
    from badslam_amd import abi
    from tests import bso
    L = bso.lib()
    cf = bso.np_buffer2d(stack.cfactor)
    dp = abi.DepthParams(cf, 0.0, float(stack.raw_to_float_depth), stack.baseline_fx, stack.cell)
    kfs = (abi.KeyframeView * K)()
    for k in range(K):
        v = kfs[k]
        v.depth, v.normals = bso.np_buffer2d(stack.depth[k]), bso.np_buffer2d(stack.normals[k])
        v.radius, v.color = bso.np_buffer2d(stack.radius[k]), bso.np_buffer2d(stack.color[k])
        _, M, Rg = stack.pose(k)
        v.frame_T_global, v.global_R_frame, v.activation, v.id = M, Rg, abi.KF_ACTIVE, k
    Hb = np.zeros((K, 27), np.float32)
    cores = host_cpu_share()

    def run(count, threads):
        stride = max(1, stack.surfels_size // count)
        surf = np.ascontiguousarray(stack.surfels[:, ::stride][:, :count])
        n = surf.shape[1]
        active = np.ones((1, n), np.uint8)
        sb, ab = bso.np_buffer2d(surf), bso.np_buffer2d(active)
        pairs3 = (C.c_uint64 * 3)()
        t0 = time.perf_counter()
        used = L.bso_bench_ba_iteration(1, int(use_desc), C.byref(stack.camera), C.byref(stack.camera), C.byref(dp), K, kfs, n, C.byref(sb),
                                        C.byref(ab), 0, threads, pairs3, Hb.ctypes.data_as(C.POINTER(C.c_float)))
        el = time.perf_counter() - t0
        return sum(pairs3), el, used, n

    pairs, el, _, n = run(2048, 1)                                  # pilot: one-thread rate
    rate1 = pairs / el
    per_surfel = pairs / n
    n1 = int(min(stack.surfels_size, max(2048, rate1 * budget_s / per_surfel)))
    p1, e1, _, n1 = run(n1, 1)
    nN = int(min(stack.surfels_size, max(2048 * cores, (p1 / e1) * cores * 0.8 * budget_s / per_surfel)))
    pN, eN, used, nN = run(nN, cores)
    what = "activation + geometry iteration + one pose-coefficient pass over all keyframes"
    return {"value": pN / eN, "unit": "pairs/s", "cores": int(used), "kind": "port",
            "sample": f"every {max(1, stack.surfels_size // nN)}th surfel of the headline stack ({nN} surfels x {K} keyframes): {what}; "
                      f"{pN} pairs in {eN:.1f} s on {used} OpenMP threads",
            "one_core": {"value": p1 / e1, "unit": "pairs/s", "cores": 1,
                         "sample": f"{n1} surfels x {K} keyframes, same passes; {p1} pairs in {e1:.1f} s"}}


if __name__ == "__main__":
    main()

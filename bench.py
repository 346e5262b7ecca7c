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
N > 1: launched by torch.distributed.run, one rank per GPU; surfels are sharded (weak scaling: every rank owns a
full-size shard), keyframes replicated, the K x 32 coefficient rows are all-reduced over RCCL once per batched GN
iteration.  On one GPU the line also carries `config.secondary` (configs[1]: 50 keyframes, geometry only), a `pcg`
block (PCG scheme on the same stacks) and `cpu_baseline` (the oracle on the host cores, N threads and 1 thread).
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
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target seconds for each of the two CPU-baseline runs")
    return ap.parse_args()


class Env:
    pass


def measure(env, K, use_desc, steps, warmup, stack=None):
    """Times `steps` BA iterations on a synthetic stack of K keyframes; returns the result dict of that workload."""
    import torch
    import torch.distributed as dist
    import badslam_amd
    from badslam_amd import abi, synthetic
    L, ctx, world, rank, device = env.L, env.ctx, env.world, env.rank, env.device
    if stack is None:
        stack = synthetic.SyntheticStack(K, seed=0xBAD51A4)
    if world > 1:   # every rank builds the same keyframe stack; rank r jitters its surfel shard differently (weak scaling)
        rng = np.random.default_rng(1000 + rank)
        stack.surfels[2] += rng.uniform(-0.001, 0.001, stack.surfels_size).astype(np.float32)
    dev = synthetic.DeviceStack(stack, device)
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

    def step():
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
    # census for the roofline accounting (untimed)
    inb, assoc = C.c_uint64(), C.c_uint64()
    badslam_amd.check(L.bslam_debug_count_pairs(ctx.handle, stream, C.byref(cam), C.byref(dp), K, kfs, S, C.byref(sb), C.byref(inb), C.byref(assoc)))
    frac_inb = inb.value / (S * K)
    frac_assoc = assoc.value / (S * K)

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
    for tag, name in ((0, "pose"), (1, "geometry"), (4, "activation")):
        launches, kms = C.c_int32(), C.c_float()
        badslam_amd.check(L.bslam_profile_read(ctx.handle, tag, C.byref(launches), C.byref(kms)))
        prof[name] = (launches.value, kms.value)
    counters = (C.c_uint64 * 8)()
    badslam_amd.check(L.bslam_profile_read_counters(ctx.handle, counters))
    badslam_amd.check(L.bslam_profile_enable(ctx.handle, 0))
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=env.control_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # pairs EVALUATED on this rank over the timed steps
    pairs_activation = int(counters[0])                 # visited until the first associated keyframe
    active_surfel_steps = int(counters[1])              # sum over steps of the surfels set active
    pairs_geometry = 2 * K * active_surfel_steps        # normals pass + position(/descriptor) pass, active surfels only
    pairs_pose = S * gn_iters_total                     # one pass per GN iteration of each unconverged keyframe
    pairs_rank = pairs_activation + pairs_geometry + pairs_pose
    value = pairs_rank * world / dt                     # weak scaling: every rank processes the same amount

    pose_launches, pose_ms = prof["pose"]
    # dominant kernel: pose_accumulate.  A launch covers the keyframes still unconverged.
    avg_kf_per_launch = gn_iters_total / max(1, pose_launches)
    bytes_per_launch = S * avg_kf_per_launch * (frac_inb * B_POSE[use_desc] + (1 - frac_inb) * B_REJECTED)
    avg_launch_s = (pose_ms / 1e3) / max(1, pose_launches)
    achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    pmc = pmc_entry(use_desc, K, S)
    roof = {"bound": "valu", "kernel": "pose_accumulate_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_value(pmc, "pose_accumulate_kernel", "hbm_bytes"),
            "convention": "achieved = ALGORITHMIC bytes per launch (SURVEY.md 8d) / average launch time, priced against the HBM peak as the "
                          "contract asks; the PMC counters say the kernel is VALU-issue bound, not HBM bound (valu_issue_frac, hbm_traffic_frac)",
            "avg_launch_us": avg_launch_s * 1e6, "launches": pose_launches, "keyframes_per_launch": avg_kf_per_launch,
            "algorithmic_bytes_per_launch": bytes_per_launch}
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
    return {
        "value": value, "ms_per_step": dt / steps * 1e3, "S": S, "K": K, "stack": stack,
        "config": {"workload": f"synthetic 640x480 stack, {K} keyframes, {S} surfels per GPU (cell 4, no merge), "
                               f"{'photometric+geometric' if use_desc else 'geometry-residual-only'} alternating BA iteration "
                               f"(BASELINE.json configs[{2 if use_desc else 1}])",
                   "keyframes": K, "surfels_per_gpu": S, "gn_iterations_per_step": gn_iters_total / steps,
                   "gn_iteration_histogram": {str(k): v for k, v in sorted(hist.items())},
                   "keyframes_converged_fraction": converged_total / (steps * K),
                   "pairs_per_step": {"activation_visited": pairs_activation / steps, "geometry": pairs_geometry / steps, "pose": pairs_pose / steps},
                   "active_surfel_fraction": active_surfel_steps / (steps * S),
                   "in_bounds_pair_fraction": frac_inb, "associated_pair_fraction": frac_assoc,
                   "parallelism": f"surfel-shard x{world}", "exchange": env.exchange},
        "roofline": roof,
    }


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
    from badslam_amd import abi
    from badslam_amd.distributed import AllReduceHook

    env = Env()
    env.rank = rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # One rank per GPU.  Default exchange ("native"): the library's own RCCL communicator (bslam_comm_init) -- the K x 32
    # coefficient rows are summed by ncclAllReduce on the BA stream, no callback; torch.distributed (gloo) only carries the
    # 128-byte communicator id, the barriers and the max over ranks of the wall time.  BSLAM_BENCH_BACKEND=nccl | gloo selects the
    # older callback path through torch.distributed.all_reduce instead (gloo: rehearsal of N ranks sharing one card).
    backend = os.environ.get("BSLAM_BENCH_BACKEND", "native")
    dev_index = local_rank if backend != "gloo" else local_rank % max(1, torch.cuda.device_count())
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
    if os.environ.get("BSLAM_GEOM_KF_CHUNK"):   # tuning runs only; the default is the library's
        badslam_amd.check(L.bslam_set_geometry_keyframe_chunk(ctx.handle, int(os.environ["BSLAM_GEOM_KF_CHUNK"])))
    hook = None
    env.cb = C.cast(None, abi.ALLREDUCE_FN)
    env.exchange = "none"
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
        if int(agreed.item()) == 1:
            env.exchange = "RCCL ncclAllReduce inside libbadslam_hip (bslam_comm_init), on the BA stream"
        else:
            # some rank could not create the library's communicator: every rank falls back to the callback path over
            # torch.distributed's RCCL group, and the line says so (config.exchange)
            if ok:
                ctx.comm_destroy()
            hook = AllReduceHook(device=True, group=dist.new_group(backend="nccl"))
            env.cb = hook.callback
            env.exchange = "torch.distributed.all_reduce callback (nccl) -- FALLBACK: bslam_comm_init failed on some rank"
    elif world > 1:
        hook = AllReduceHook(device=True)
        env.cb = hook.callback
        env.exchange = f"torch.distributed.all_reduce callback ({backend})"
    if world == 1 and os.environ.get("BSLAM_BENCH_SELF_RCCL"):
        # rehearsal of the N > 1 exchange on one GPU: a one-rank communicator, for which the all-reduce is the identity but
        # goes through the same kernel sequence and RCCL launch (not a bench configuration)
        ctx.comm_init(badslam_amd.comm_unique_id(), 0, 1)
        env.exchange = "RCCL, one-rank rehearsal"
    if world == 1 and os.environ.get("BSLAM_BENCH_NOOP_HOOK"):
        # rehearsal of the N > 1 kernel sequence on one GPU: the exchange is a no-op callback (not a bench configuration)
        noop = abi.ALLREDUCE_FN(lambda user, ptr, count, stream: 0)
        env.cb = noop

    use_desc = bool(args.photometric)
    head = measure(env, args.keyframes, use_desc, args.steps, args.warmup)
    out = {
        "metric": "surfel x keyframe residual evaluations per second, full alternating BA iteration",
        "value": head["value"], "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "config": head["config"], "roofline": head["roofline"],
    }
    single = world == 1 and rank == 0
    small_stack = None
    if single and args.secondary and not (args.keyframes == 50 and not use_desc):
        sec = measure(env, 50, False, 20, 3)
        small_stack = sec["stack"]
        out["config"]["secondary"] = {"value": sec["value"], "unit": "pairs/s", "ms_per_step": sec["ms_per_step"], "steps": 20, "warmup": 3,
                                      "config": sec["config"], "roofline": sec["roofline"]}
    if single and args.pcg:
        out["pcg"] = {"headline_stack": pcg_block(head["stack"], use_desc, dev_index)}
        if small_stack is not None:
            out["pcg"]["secondary_stack"] = pcg_block(small_stack, False, dev_index)
    if single and args.cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(head["stack"], head["K"], use_desc, args.cpu_seconds)
    if dist.is_initialized():
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


def pmc_entry(use_desc, K, S):
    """Per-kernel PMC figures of this exact workload from the committed counter passes (profiles/pmc_traffic.json, written from
    tools/pmc.sh runs of this same command; counters cannot be read from inside the timed process), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            j = json.load(f)
        for w in j.get("workloads", []):
            if (w.get("keyframes"), w.get("surfels_per_gpu"), bool(w.get("photometric"))) == (K, S, bool(use_desc)):
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


def pcg_block(stack, use_desc, dev_index, iterations=2):
    """PCG scheme (BS/direct_ba_pcg.cc:229-471) on the same stack through the C++ host class: poses + geometry, one outer
    iteration per call.  Pairs: the normals pass, PCGInit and one PCGStep1 pass per inner step, each over all S x K pairs."""
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
        nbytes = S * K * B_POSE[use_desc]                  # upper bound: every pair priced as in-bounds (no census on this path)
        avg = step1_ms / 1e3 / step1_launches
        out["pcg_step1_kernel"] = {"avg_launch_us": avg * 1e6, "launches": step1_launches, "achieved": nbytes / avg / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": nbytes / avg / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes,
                                   "note": "all S x K pairs priced at the in-bounds figure of SURVEY.md 8(d)"}
    if init_launches:
        out["pcg_init_kernel"] = {"avg_launch_us": init_ms * 1e3 / init_launches, "launches": init_launches}
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

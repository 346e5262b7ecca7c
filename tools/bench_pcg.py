#!/usr/bin/env python3
"""PCG-scheme BA iteration on the synthetic stack of bench.py through the C++ host class
(badslam_amd/host/direct_ba.*).  Tuning tool: prints ms per outer iteration; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=50)
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--photometric", type=int, default=0)
    ap.add_argument("--alternating", type=int, default=0)
    ap.add_argument("--inner", type=int, default=30)
    args = ap.parse_args()
    import torch  # noqa: F401  (HIP runtime first)
    from badslam_amd import synthetic
    from badslam_amd.direct_ba import DirectBA
    K = args.keyframes
    stack = synthetic.SyntheticStack(K, seed=0xBAD51A4)
    cam = stack.camera
    ba = DirectBA(stack.surfels_size, float(stack.raw_to_float_depth), stack.baseline_fx, stack.cell, 0.8, 1, 1, 1,
                  cam, cam, 0, True, bool(args.photometric))
    ba.set_options(pcg_gauge_keyframe=0)
    rng = np.random.default_rng(7)
    for k in range(K):
        xi = np.concatenate([rng.choice([-1, 1], 3) * 0.005, rng.choice([-1, 1], 3) * 0.001])
        T = stack.pose(k, xi if k else None)[0]
        ba.AddKeyframe(k, 0.3, 6.0, stack.depth[k], stack.normals[k], stack.radius[k], stack.color[k], T)
    ba.SetSurfels(stack.surfels[:8], stack.surfels_size)
    use_pcg = not args.alternating
    ba.BundleAdjustment(False, False, False, True, True, 1, 1, use_pcg, 0, K - 1, True, args.inner)   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iterations):
        ba.BundleAdjustment(False, False, False, True, True, 1, 1, use_pcg, 0, K - 1, True, args.inner)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iterations
    print(f"{'PCG' if use_pcg else 'alternating'} BA iteration: {dt * 1e3:.3f} ms  (K={K}, S={stack.surfels_size}, photometric={args.photometric})")


if __name__ == "__main__":
    main()

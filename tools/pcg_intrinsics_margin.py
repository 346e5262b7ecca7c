#!/usr/bin/env python3
"""Run-to-run spread of the PCG intrinsics known-answer test (float atomics on the cfactor cells):
prints the final intrinsics error of N repetitions of tests/test_gpu_direct_ba.py::test_intrinsics_optimization_with_geometric_residual[True]."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import bso, scenes                                   # noqa: E402
from tests.test_gpu_direct_ba import intrinsics_test_ba         # noqa: E402

bso.build_oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for rep in range(n):
    true = scenes.intrinsics_test_camera()
    scene = scenes.intrinsics_scene(36, seed=0, cell=2, max_surfels=1000 * 1000, camera=true, create_surfels=False)
    ba = intrinsics_test_ba(scene)
    d = scenes.distorted_camera(true)
    ba.set_intrinsics(None, [d.fx, d.fy, d.cx, d.cy], 0.0)
    for i in range(100):
        ba.BundleAdjustment(True, False, False, False, False, 1, 10, True, 0, len(scene.keyframes) - 1, i != 0)
    _, dc, a = ba.intrinsics()
    err = np.abs(dc - np.array([true.fx, true.fy, true.cx, true.cy], np.float32))
    print(rep, err, float(a), flush=True)

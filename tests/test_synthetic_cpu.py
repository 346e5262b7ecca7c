"""The synthetic stacks of bench.py (badslam_amd/synthetic.py) on CPU: the dense stack -- the headline workload
since round 1 -- stays what it was, bit for bit; the trajectory stack really is a trajectory (every pixel measured, depths inside
the sensor range, smooth motion, a keyframe sees a bounded part of the scene); the survey stack uses the survey's pose ranges."""
import hashlib

import numpy as np

from badslam_amd import synthetic


def test_dense_stack_is_unchanged():
    st = synthetic.SyntheticStack(3, seed=0xBAD51A4)
    h = hashlib.sha256()
    for a in (st.surfels, st.depth, st.normals, st.radius, st.color):
        h.update(np.ascontiguousarray(a).tobytes())
    assert st.surfels_size == 57600
    assert h.hexdigest() == "863f896341a485192433c419fca561f7642576658ba204cd7723058dbcb16f50"


def in_view_fraction(st, keyframes):
    p = st.surfels[:3].astype(np.float64)
    cam = st.camera
    out = []
    for k in keyframes:
        R, t = st.R[k], st.t[k]
        loc = R.T @ (p - t[:, None])
        z = loc[2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u, v = cam.fx * loc[0] / z + cam.cx, cam.fy * loc[1] / z + cam.cy
        out.append(((z > 0) & (u >= 0) & (v >= 0) & (u < cam.width) & (v < cam.height)).mean())
    return float(np.mean(out))


def test_trajectory_stack_sees_a_bounded_part_of_the_scene():
    K = 40
    st = synthetic.SyntheticStack(K, kind="trajectory")
    assert st.surfels_size == 19200 * K                                   # every cell of every keyframe has a measurement
    depth = st.depth.astype(np.float64) * float(st.raw_to_float_depth)
    inner = depth[:, 1:-1, 1:-1]
    assert (st.depth[:, 1:-1, 1:-1] < 32768).all() and 1.0 < inner.min() and inner.max() < 6.0
    steps = [np.linalg.norm(st.t[k + 1] - st.t[k]) for k in range(K - 1)]
    assert max(steps) < 2.5 * (2 * np.pi * 2.5 / K)                        # one smooth lap: no jumps
    turn = [np.degrees(np.arccos(np.clip((np.trace(st.R[k].T @ st.R[k + 1]) - 1) / 2, -1, 1))) for k in range(K - 1)]
    assert max(turn) < 3 * 360.0 / K
    f = in_view_fraction(st, range(0, K, 5))
    assert 0.05 < f < 0.2, f                                               # about a tenth of the room per keyframe
    dense = synthetic.SyntheticStack(6)
    assert in_view_fraction(dense, range(6)) > 0.6                         # ... against most of the scene on the dense stack


def test_survey_stack_uses_the_reference_tests_pose_ranges():
    st = synthetic.SyntheticStack(30, kind="survey")
    t = np.array(st.t)
    assert np.abs(t).max() > 0.8 and np.abs(t).max() <= 1.5 * 1.8          # exp() of a +-1.5 m / +-0.7 rad twist
    assert 0.1 < in_view_fraction(st, range(0, 30, 3)) < 0.6

#!/usr/bin/env python3
"""Writes tests/golden/oracle_regression.json: a few outputs of the ORACLE on small seeded scenes.

These are NOT reference outputs (the reference cannot run here and ships no vectors for this path):
they freeze the oracle's behaviour at the state in which it passed the reference's known-answer tests, so that a later
change to oracle/ that alters an integer decision or a sum shows up in the CPU suite (tests/test_golden_regression.py).
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import bso, scenes  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def compute():
    bso.build_oracle()
    cam = bso.make_camera(262.5, 262.5, 160.0, 120.0, 320, 240)
    out = {}
    scene = scenes.synthetic_scene(3, seed=21, width=320, height=240, cell=4, camera=cam, use_depth_residuals=True, use_descriptor_residuals=True)
    out["surfels_size"] = int(scene.surfels_size)
    out["keyframe_depth_sha"] = [digest(kf.depth) for kf in scene.keyframes]
    out["keyframe_normals_sha"] = [digest(kf.normals) for kf in scene.keyframes]
    out["keyframe_radius_sha"] = [digest(kf.radius) for kf in scene.keyframes]
    out["surfel_normals_sha"] = digest(scene.surfels[3, :scene.surfels_size].view(np.uint32))
    out["association_sha"] = [digest(scene.association(kf)) for kf in scene.keyframes]
    r = scene.accumulate_pose(scene.keyframes[1])
    out["pose_count"] = int(r["count"])
    out["pose_H64"] = [float(v) for v in r["H64"]]
    out["pose_b64"] = [float(v) for v in r["b64"]]
    scene.update_activation()
    out["active_count"] = int((scene.active[0, :scene.surfels_size] & 1).sum())
    scene.optimize_geometry_iteration()
    out["geometry_normals_sha"] = digest(scene.surfels[3, :scene.surfels_size].view(np.uint32))
    out["geometry_position_sum"] = [float(scene.surfels[i, :scene.surfels_size].astype(np.float64).sum()) for i in range(3)]
    # lifecycle
    count = scene.surfels_size
    for kf in scene.keyframes:
        count = scene.merge_surfels(kf, 0.8, count)
    out["after_merge"] = int(count)
    count = scene.delete_surfels_and_update_radii(2, count)
    out["after_delete"] = int(count)
    scene.compact_surfels(count)
    out["compacted_x_sha"] = digest(scene.surfels[0, :count].view(np.uint32))
    # odometry
    two = scenes.synthetic_scene(2, seed=13, width=320, height=240, cell=4, camera=cam, use_depth_residuals=True, use_descriptor_residuals=True,
                                 translation_range=0.03, rotation_range=0.02)
    est, its = two.track_frame_pairwise(two.keyframes[1], two.keyframes[0], bso.se3_identity(), num_scales=4)
    out["tracking_iterations"] = [int(i) for i in its]
    out["tracking_pose"] = [float(v) for v in bso.se3_to_np(est)]
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.json")
    with open(path, "w") as f:
        json.dump(compute(), f, indent=1)
    print("wrote", path)

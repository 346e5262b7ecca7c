"""Pins the ORACLE against the reference's own known-answer tests: the
reference ships no golden vectors for this path, only GPU integration tests with
closed-form answers; each is re-stated here with the reference's tolerance."""
import numpy as np
import pytest

from tests import bso, scenes


def pose_error(est, gt):
    # (global_tr_frame_estimate.inverse() * global_tr_frame).log()
    return bso.se3_log(bso.se3_mul(bso.se3_inverse(est), gt))


@pytest.fixture(scope="module")
def geo_scene(oracle):
    return scenes.pose_geometric_scene(seed=0)


def test_pose_optimization_with_geometric_residual(geo_scene):
    """BS/test/test_pose_optimization_geometric_residual.cc:50-171, tolerance 1.1e-6 (:168)."""
    scene, kf = geo_scene
    assert 150000 < scene.surfels_size < 180000
    gt = bso.se3_identity()
    worst = 0.0
    for off in scenes.offsets_13(0.005, 0.001):
        init = bso.se3_mul(off, bso.se3_inverse(gt))
        est, iters, conv = scene.estimate_frame_pose(kf, init)
        err = pose_error(est, gt)
        worst = max(worst, float(np.abs(err).max()))
        assert conv and iters <= 30
    assert worst < 1.1e-6, worst


def test_pose_optimization_color_only_cues(oracle):
    """BS/test/test_pose_optimization_photometric_residual.cc:50-178, tolerance 8e-5 (:175)."""
    scene, kf, gt = scenes.pose_photometric_scene(seed=0)
    worst = 0.0
    for off in scenes.offsets_13(0.0005, 0.001):
        init = bso.se3_mul(gt, off)
        est, iters, conv = scene.estimate_frame_pose(kf, init)
        err = pose_error(est, gt)
        worst = max(worst, float(np.abs(err).max()))
    assert worst < 8e-5, worst

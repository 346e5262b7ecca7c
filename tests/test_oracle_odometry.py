"""CPU: the oracle's restatement of TrackFramePairwise (oracle/bso_odometry.c), the checker of tests/test_gpu_odometry.py.
The reference has no unit test of its tracker: the known answer is the relative pose two synthetic views of the plane
scene were rendered with."""
import numpy as np
import pytest

from tests import bso, scenes


@pytest.mark.parametrize("use_desc", [False, True])
def test_track_frame_pairwise_recovers_the_rendered_motion(oracle, use_desc):
    cam = bso.make_camera(262.5, 262.5, 160.0, 120.0, 320, 240)
    scene = scenes.synthetic_scene(2, seed=13, width=320, height=240, cell=4, camera=cam, use_depth_residuals=True, use_descriptor_residuals=use_desc,
                                   translation_range=0.03, rotation_range=0.02)
    base, tracked = scene.keyframes
    truth = bso.se3_mul(bso.se3_inverse(base.global_T_frame), tracked.global_T_frame)
    est, its = scene.track_frame_pairwise(tracked, base, bso.se3_identity(), num_scales=4)
    err = np.abs(bso.se3_log(bso.se3_mul(bso.se3_inverse(est), truth))).max()
    assert np.abs(bso.se3_log(truth)).max() > 0.01 and err < (1e-4 if use_desc else 2e-5), (err, its)
    assert its[3] > its[0]                                   # most of the work happens at the coarsest scale
    # a wildly wrong first candidate loses against the identity at the coarsest scale (BS/pairwise_frame_tracking.cc:428-489)
    bad = bso.se3_exp(np.array([0.6, -0.5, 0.4, 0.3, -0.3, 0.2], np.float32))
    est2, _ = scene.track_frame_pairwise(tracked, base, bad, bso.se3_identity(), num_scales=4, test_different_initial_estimates=True)
    assert np.abs(bso.se3_log(bso.se3_mul(bso.se3_inverse(est2), truth))).max() < (1e-4 if use_desc else 2e-5)


@pytest.mark.parametrize("use_gradmag,use_pyramid_level_0", [(True, True), (False, False), (True, False)])
def test_tracker_variants_recover_the_rendered_motion(oracle, use_gradmag, use_pyramid_level_0):
    """The two switches the reference's tracker takes besides the residual types (BS/pairwise_frame_tracking.cc:164-166):
    use_gradmag (one colour residual on Sobel gradient-magnitude images, BS/kernel_opt_pose.cu:713-937,1173-1338; exercised by
    BS/test/test_pairwise_frame_tracking.cc:425-547, which logs errors without a bar) and use_pyramid_level_0 = false (the tracked
    frame enters at level 1 through CalibrateAndDownsampleImagesCUDA, level 0 is not tracked)."""
    cam = bso.make_camera(262.5, 262.5, 160.0, 120.0, 320, 240)
    scene = scenes.synthetic_scene(2, seed=13, width=320, height=240, cell=4, camera=cam, use_depth_residuals=True, use_descriptor_residuals=True,
                                   translation_range=0.03, rotation_range=0.02)
    base, tracked = scene.keyframes
    truth = bso.se3_mul(bso.se3_inverse(base.global_T_frame), tracked.global_T_frame)
    bso.lib().bso_set_tracking_variant(int(use_gradmag), int(use_pyramid_level_0))
    try:
        est, its = scene.track_frame_pairwise(tracked, base, bso.se3_identity(), num_scales=4)
    finally:
        bso.lib().bso_set_tracking_variant(0, 1)
    err = np.abs(bso.se3_log(bso.se3_mul(bso.se3_inverse(est), truth))).max()
    assert err < 3e-4, (err, its)
    if not use_pyramid_level_0:
        assert its[0] == 0 and its[1] > 0           # level 0 is skipped

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


def surfel_depth_errors(scene, kf, new_depth):
    """BS/test/test_geometry_optimization_geometric_residual.cc:184-208."""
    n = scene.surfels_size
    M = np.array(list(bso.se3_matrix3x4(bso.se3_inverse(kf.global_T_frame)).m), np.float64).reshape(3, 4)
    p = M[:, :3] @ scene.surfels[:3, :n].astype(np.float64) + M[:, 3:4]
    cam = scene.depth_camera
    u = cam.fx * p[0] / p[2] + cam.cx
    v = cam.fy * p[1] / p[2] + cam.cy
    ok = (p[2] > 0) & (u >= 0) & (v >= 0) & (u < cam.width) & (v < cam.height)
    px, py = u[ok].astype(int), v[ok].astype(int)
    valid = new_depth[py, px] < 32768
    expected = scene.raw_to_float_depth * new_depth[py, px].astype(np.float64)
    return np.abs(p[2][ok] - expected)[valid]


def test_alternating_geometry_optimization_with_geometric_residual(oracle):
    """Optimization.AlternatingGeometryOptimizationWithGeometricResidual
    (BS/test/test_geometry_optimization_geometric_residual.cc:50-216): 10 x 10 alternating BA
    iterations (geometry only) put every surfel on the perturbed depth map within 1e-4 m.  The
    reference's end-of-scheme surfel deletion is out of scope, so never-associated surfels (if any)
    are tolerated at <= 0.05 %."""
    scene, kf, new_depth = scenes.geometry_geometric_scene(seed=0)
    assert (surfel_depth_errors(scene, kf, new_depth) > 1e-4).mean() > 0.5
    for _ in range(100):
        scene.update_activation()
        scene.optimize_geometry_iteration()
    err = surfel_depth_errors(scene, kf, new_depth)
    assert (err > 1e-4).sum() <= 0.0005 * err.size, ((err > 1e-4).sum(), err.size)


def test_pcg_geometry_optimization_with_geometric_residual(oracle):
    """Optimization.PCGGeometryOptimizationWithGeometricResidual (same file, :220): the PCG scheme
    reaches the same bar; the loop is stopped as soon as it does (the reference runs 100 outer
    iterations unconditionally)."""
    from tests import oracle_ba
    scene, kf, new_depth = scenes.geometry_geometric_scene(seed=0)
    ok = False
    for it in range(12):
        oracle_ba.pcg_ba_iteration(scene, optimize_poses=False, optimize_geometry=True)
        err = surfel_depth_errors(scene, kf, new_depth)
        if (err > 1e-4).sum() <= 0.0005 * err.size:
            ok = True
            break
    assert ok, ((err > 1e-4).sum(), err.size)


def plane_depth_census(scene, kf1, expected_z, surfels=None):
    """ComputeError of BS/test/test_geometry_optimization_photometric_residual.cc:50-125:
    (num_correct, num_fails) at 1e-3 m over the surfels that project into keyframe 1."""
    n = scene.surfels_size
    S = scene.surfels[:3, :n] if surfels is None else surfels[:3]
    M = np.array(list(bso.se3_matrix3x4(bso.se3_inverse(kf1.global_T_frame)).m), np.float64).reshape(3, 4)
    p = M[:, :3] @ S.astype(np.float64) + M[:, 3:4]
    cam = scene.depth_camera
    u = cam.fx * p[0] / p[2] + cam.cx
    v = cam.fy * p[1] / p[2] + cam.cy
    ok = (p[2] > 0) & (u >= 0) & (v >= 0) & (u < cam.width) & (v < cam.height)
    err = np.abs(p[2][ok] - expected_z)
    return int((err <= 1e-3).sum()), int((err > 1e-3).sum())


def test_alternating_geometry_optimization_with_photometric_residual(oracle):
    """Optimization.AlternatingGeometryOptimizationWithPhotometricResidual
    (BS/test/test_geometry_optimization_photometric_residual.cc:128-275): 60 alternating iterations
    with descriptor residuals only; the reference's bar is num_correct >= 100000, num_fails <= 75000."""
    scene, kf0, kf1, ez = scenes.geometry_photometric_scene()
    c0, f0 = plane_depth_census(scene, kf1, ez)
    assert c0 < 100000 and f0 > 75000     # the noisy start does not meet the bar
    for _ in range(60):
        scene.update_activation()
        scene.optimize_geometry_iteration()
    correct, fails = plane_depth_census(scene, kf1, ez)
    assert correct >= 100000 and fails <= 75000, (correct, fails)


def test_pcg_geometry_optimization_with_photometric_residual(oracle):
    """Optimization.PCGGeometryOptimizationWithPhotometricResidual (same file, :277): the PCG scheme
    meets the same bar; stopped as soon as it does (the reference runs 60 iterations)."""
    from tests import oracle_ba
    scene, kf0, kf1, ez = scenes.geometry_photometric_scene()
    ok = False
    for _ in range(6):
        oracle_ba.pcg_ba_iteration(scene, optimize_poses=False, optimize_geometry=True)
        correct, fails = plane_depth_census(scene, kf1, ez)
        if correct >= 100000 and fails <= 75000:
            ok = True
            break
    assert ok, (correct, fails)


def camera_errors(est, true):
    return [abs(est.fx - true.fx), abs(est.fy - true.fy), abs(est.cx - true.cx), abs(est.cy - true.cy)]


def test_intrinsics_optimization_with_photometric_residual(oracle):
    """Optimization.AlternatingIntrinsicsOptimizationWithPhotometricResidual as the reference runs it
    (BS/test/test_intrinsics_optimization_photometric_residual.cc:104-265) at FULL size on the oracle: true camera
    {0.5h, 0.45h, 0.5w - 0.5, 0.5h - 0.5} (:112), 12 keyframes of 20 planes, descriptor residuals only, surfels created with the
    observation filter from every keyframe with the true camera (:216-218), colour camera then set to true + (+0.5, -0.6, +1.23,
    -2.17) px (:150, :219); 10 x BundleAdjustment(colour intrinsics only, do_surfel_updates, min 1 / max 10 iterations) = 10 single
    steps (no pose optimisation: the loop exits after min_iterations, BS/direct_ba_alternating.cc:693-700), the first call
    with increase_ba_iteration_count = false; bars 0.03 px (fx, fy) / 0.15 px (cx, cy) (:262-265)."""
    from tests import oracle_ba
    true = scenes.intrinsics_test_camera()
    scene = scenes.intrinsics_scene(12, seed=0, cell=2, max_surfels=1000 * 1000, photometric=True, camera=true, filter_new_surfels=True)
    assert scene.surfels_size > 300000
    scene.color_camera = scenes.distorted_camera(true)
    ba = oracle_ba.OracleAlternatingBA(scene)
    for i in range(10):
        ba.bundle_adjustment(False, True, True, False, 1, i != 0)
    err = camera_errors(scene.color_camera, true)
    assert err[0] < 0.03 and err[1] < 0.03 and err[2] < 0.15 and err[3] < 0.15, err


def test_intrinsics_optimization_with_geometric_residual(oracle):
    """Optimization.AlternatingIntrinsicsOptimizationWithGeometricResidual
    (BS/test/test_intrinsics_optimization_geometric_residual.cc:371-553) at FULL size on the oracle: true camera
    {0.5h, 0.45h, ...} (:378), 36 keyframes, depth residuals only, filtered surfel creation with the true camera (:515-517), depth
    camera then set to true + (+0.5, -0.6, +1.23, -2.17) px (:416, :518), BundleAdjustment(depth intrinsics only, no surfel
    updates, min 1 / max 10) per call; bar 1e-3 px (:539-542).  The reference issues 100 calls; the estimate is inside the bar
    after 10 and then moves by < 1e-4 px per call (1 mm depth quantisation), so the CPU run stops after 14 (the full 100 run
    through the HIP path in tests/test_gpu_direct_ba.py)."""
    from tests import oracle_ba
    true = scenes.intrinsics_test_camera()
    scene = scenes.intrinsics_scene(36, seed=0, cell=2, max_surfels=1000 * 1000, camera=true, filter_new_surfels=True)
    scene.depth_camera = scenes.distorted_camera(true)
    ba = oracle_ba.OracleAlternatingBA(scene)
    for i in range(14):
        ba.bundle_adjustment(True, False, False, False, 1, i != 0)
    err = camera_errors(scene.depth_camera, true)
    assert max(err) < 1e-3, err
    # (a and the cfactors are optimised along: the cfactors settle at ~5e-5, fitting the 1 mm depth quantisation, which leaves
    # `a` all but unconstrained -- it drifts to about -2 against its weak prior and stays there.  The reference does not test them.)
    assert np.abs(scene.cfactor).max() < 2e-3


def test_depth_deformation_optimization_with_geometric_residual_reduced(oracle):
    """Optimization.AlternatingDepthDeformationOptimizationWithGeometricResidual
    (BS/test/test_intrinsics_optimization_geometric_residual.cc:178-368) on the oracle at reduced size (160x120; the full
    640x480 x 400 calls run through the HIP path): 12 keyframes whose depth is distorted with a = 0.03, cfactor = 0.005
    through the inverse deformation model (:111-125); BA starts without surfels, a = 0, cfactor = 0; every call does surfel
    updates + geometry and, from the second call on, depth intrinsics (:322-337); bars |a - 0.03| < 1e-2 and
    |cfactor(cell) - 0.005| < 1e-3 (:345-347; the reference probes cell (50, 50), here the central cell of the smaller image)."""
    from tests import oracle_ba
    w, h = 160, 120
    true_a, true_cf = 0.03, 0.005
    scene = scenes.intrinsics_scene(12, seed=0, width=w, height=h, cell=2, max_surfels=200000, distortion=(true_a, true_cf), create_surfels=False)
    ba = oracle_ba.OracleAlternatingBA(scene)
    for i in range(400):
        ba.bundle_adjustment(i != 0, False, True, True, 1, i != 0)
    cf = scene.cfactor
    assert scene.surfels_size > 5000
    assert abs(scene.a - true_a) < 1e-2, scene.a
    assert abs(cf[cf.shape[0] // 2, cf.shape[1] // 2] - true_cf) < 1e-3, cf[cf.shape[0] // 2, cf.shape[1] // 2]

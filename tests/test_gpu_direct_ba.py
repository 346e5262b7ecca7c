"""-m gpu: the reference's known-answer BA tests driven through the C++ host class
(badslam_amd/host/direct_ba.*) over the HIP kernels, plus host-loop parity against an
oracle-driven restatement of the same loops."""
import numpy as np
import pytest

from badslam_amd import abi
from tests import bso, scenes

pytestmark = pytest.mark.gpu


def make_ba(scene, **kw):
    from badslam_amd.direct_ba import DirectBA
    ba = DirectBA(scene.max_surfels, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, 0.8, 1, 1, 1,
                  scene.color_camera, scene.depth_camera, 0, scene.use_depth_residuals, scene.use_descriptor_residuals)
    ba.set_options(texture_mode=scene.tex_mode, **kw)
    for kf in scene.keyframes:
        ba.AddKeyframe(kf.id, max(kf.min_depth, 1e-3), max(kf.max_depth, 1e-2), kf.depth, kf.normals, kf.radius, kf.color, kf.global_T_frame)
    ba.SetSurfels(scene.surfels[:8], scene.surfels_size)
    return ba


def pose_error(est, gt):
    return bso.se3_log(bso.se3_mul(bso.se3_inverse(est), gt))


def test_estimate_frame_pose_known_answer(oracle):
    """Optimization.PoseOptimizationWithGeometricResidual through DirectBA::EstimateFramePose
    (BS/test/test_pose_optimization_geometric_residual.cc:134-170), bar 1.1e-6."""
    scene, kf = scenes.pose_geometric_scene(seed=0)
    ba = make_ba(scene)
    gt = bso.se3_identity()
    worst = 0.0
    for off in scenes.offsets_13(0.005, 0.001):
        est = ba.EstimateFramePose(0, bso.se3_mul(off, bso.se3_inverse(gt)))
        worst = max(worst, float(np.abs(pose_error(est, gt)).max()))
    assert worst < 1.1e-6, worst


def depth_check(scene, kf, new_depth, surfels):
    """BS/test/test_geometry_optimization_geometric_residual.cc:184-208 (only surfels that project
    into the image are checked there as well)."""
    M = np.array(list(bso.se3_matrix3x4(bso.se3_inverse(kf.global_T_frame)).m), np.float64).reshape(3, 4)
    p = M[:, :3] @ surfels[:3].astype(np.float64) + M[:, 3:4]
    cam = scene.depth_camera
    u = cam.fx * p[0] / p[2] + cam.cx
    v = cam.fy * p[1] / p[2] + cam.cy
    ok = (p[2] > 0) & (u >= 0) & (v >= 0) & (u < cam.width) & (v < cam.height)
    px, py = u[ok].astype(int), v[ok].astype(int)
    expected = scene.raw_to_float_depth * new_depth[py, px].astype(np.float64)
    valid = new_depth[py, px] < 32768
    err = np.abs(p[2][ok] - expected)
    return err[valid]


@pytest.mark.parametrize("use_pcg", [False, True])
def test_geometry_optimization_with_geometric_residual(oracle, use_pcg):
    """{Alternating,PCG}GeometryOptimizationWithGeometricResidual
    (BS/test/test_geometry_optimization_geometric_residual.cc:50-220): after 10 x 10 BA iterations
    every surfel sits on the (perturbed) measured depth within 1e-4 m."""
    scene, kf, new_depth = scenes.geometry_geometric_scene(seed=0)
    ba = make_ba(scene, pcg_gauge_keyframe=0)
    before = depth_check(scene, kf, new_depth, scene.surfels[:8, :scene.surfels_size])
    assert (before > 1e-4).mean() > 0.5
    for _ in range(10):
        ba.BundleAdjustment(False, False, False, False, True, 10, 10, use_pcg, 0, 0, True)
    err = depth_check(scene, kf, new_depth, ba.GetSurfels(8))
    # The reference deletes unobserved / outlier surfels in PerformBASchemeEndTasks (out of scope here),
    # so a handful of never-associated surfels may remain; everything that is associated must be on the surface.
    fails = int((err > 1e-4).sum())
    assert fails <= 0.0005 * err.size, (fails, err.size, float(np.sort(err)[-10:].mean()))


@pytest.mark.parametrize("use_pcg", [False, True])
def test_geometry_optimization_with_photometric_residual(oracle, use_pcg):
    """{Alternating,PCG}GeometryOptimizationWithPhotometricResidual
    (BS/test/test_geometry_optimization_photometric_residual.cc:128-280): 60 BA iterations with
    descriptor residuals only over two keyframes; bar num_correct >= 100000, num_fails <= 75000."""
    from tests.test_oracle_known_answers import plane_depth_census
    scene, kf0, kf1, ez = scenes.geometry_photometric_scene()
    ba = make_ba(scene, pcg_gauge_keyframe=0)
    ba.BundleAdjustment(False, False, False, False, True, 60, 60, use_pcg, 0, 1, True)
    correct, fails = plane_depth_census(scene, kf1, ez, ba.GetSurfels(8))
    assert correct >= 100000 and fails <= 75000, (correct, fails)
    if not use_pcg:
        # same loop on the oracle: the census must agree closely (threshold flips only)
        for _ in range(60):
            scene.update_activation()
            scene.optimize_geometry_iteration()
        oc, of = plane_depth_census(scene, kf1, ez)
        assert abs(oc - correct) <= 0.002 * (oc + of), (oc, of, correct, fails)


def test_do_surfel_updates_lifecycle_through_the_host_loop(oracle, tmp_path):
    """BundleAdjustment(do_surfel_updates = true): the BA creates (filtered), merges, deletes and compacts the surfels
    itself (BS/direct_ba_alternating.cc:396-533, BS/direct_ba.cc:566-653).  Same geometry known-answer as above, but
    starting without any surfel."""
    scene, kf, new_depth = scenes.geometry_geometric_scene(seed=0)
    scene.surfels_size = 0
    ba = make_ba(scene, pcg_gauge_keyframe=0)
    assert ba.surfels_size() == 0
    timings = str(tmp_path / "timings.txt")
    ba.set_timings_file(timings)
    for _ in range(10):
        ba.BundleAdjustment(False, False, True, False, True, 10, 10, False, 0, 0, True)
    ba.set_timings_file(None)
    # --save_timings lines of one BA iteration with surfel updates and geometry optimisation, in the reference's order
    # (BS/direct_ba_alternating.cc:630-688): header, creation, activation, geometry, merge, compaction
    lines = open(timings).read().split("\n")
    assert lines[0].startswith("BA_count ") and " inner_iteration 0 " in lines[0] and " keyframe_count 1 " in lines[0]
    assert [ln.split()[0] for ln in lines[1:6]] == ["BA_surfel_creation", "BA_surfel_activation", "BA_geometry_optimization",
                                                  "BA_initial_surfel_merge", "BA_surfel_compaction"]
    assert all(float(ln.split()[1]) >= 0.0 for ln in lines[1:6])
    assert sum(ln.startswith("BA_count") for ln in lines) >= 10
    n = ba.surfels_size()
    assert n > 250000, n
    surf = ba.GetSurfels(8)
    assert not np.isnan(surf[0]).any()                       # compacted: no deleted surfel left
    err = depth_check(scene, kf, new_depth, surf)
    assert (err > 1e-4).sum() == 0, ((err > 1e-4).sum(), err.size)   # end-of-scheme deletion removes the never-associated ones


@pytest.mark.parametrize("use_pcg", [False, True])
def test_depth_deformation_optimization_with_geometric_residual(oracle, use_pcg):
    """{Alternating,PCG}DepthDeformationOptimizationWithGeometricResidual
    (BS/test/test_intrinsics_optimization_geometric_residual.cc:286-368): 12 keyframes of 20 planes whose depth is
    distorted with a = 0.03, cfactor = 0.005; 400 (alternating) / 20 (PCG) BA calls with surfel updates, geometry and --
    from the second call on -- depth intrinsics; bars |a - 0.03| < 1e-2 and |cfactor(50, 50) - 0.005| < 1e-3 (:361-365)."""
    true_a, true_cf = 0.03, 0.005
    scene = scenes.intrinsics_scene(12, seed=0, cell=2, max_surfels=1000 * 1000, distortion=(true_a, true_cf), create_surfels=False)
    from badslam_amd.direct_ba import DirectBA
    ba = DirectBA(scene.max_surfels, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, 0.8, 2, 2, 2,
                  scene.color_camera, scene.depth_camera, 0, True, False)
    ba.set_options(texture_mode=scene.tex_mode, pcg_gauge_keyframe=0)
    for kf in scene.keyframes:
        ba.AddKeyframe(kf.id, max(kf.min_depth, 1e-3), max(kf.max_depth, 1e-2), kf.depth, kf.normals, kf.radius, kf.color, kf.global_T_frame)
    K = len(scene.keyframes)
    for i in range(20 if use_pcg else 400):
        ba.BundleAdjustment(i != 0, False, True, False, True, 1, 10, use_pcg, 0, K - 1, i != 0)
    _, _, a = ba.intrinsics()
    cf = ba.cfactor(scene.cfactor.shape)
    assert ba.surfels_size() > 100000
    assert abs(a - true_a) < 1e-2, a
    assert abs(cf[50, 50] - true_cf) < 1e-3, cf[50, 50]


def oracle_alternating_iteration(scene, covis):
    """One iteration of BS/direct_ba_alternating.cc:345-717 with the oracle's kernels
    (whole window, no surfel updates, sequential EstimateFramePose)."""
    scene.update_activation()
    scene.optimize_geometry_iteration()
    num_converged = 0
    for kf in scene.keyframes:
        if kf.activation == abi.KF_INACTIVE:
            num_converged += 1
            continue
        est, _, _ = scene.estimate_frame_pose(kf, kf.global_T_frame)
        diff = bso.se3_log(bso.se3_mul(bso.se3_inverse(kf.global_T_frame), est))
        moved = not bso.lib().bso_is_scale1_pose_estimation_converged(bso.fptr(diff))
        kf.global_T_frame = est
        kf.activation = abi.KF_ACTIVE if moved else abi.KF_INACTIVE
        num_converged += 0 if moved else 1
    done = num_converged == len(scene.keyframes)
    if not done:
        for kf in scene.keyframes:   # DetermineCovisibleActiveKeyframes
            if kf.activation == abi.KF_ACTIVE:
                for j in covis[kf.id]:
                    if scene.keyframes[j].activation == abi.KF_INACTIVE:
                        scene.keyframes[j].activation = abi.KF_COVISIBLE_ACTIVE
    return done


@pytest.mark.parametrize("batched", [True, False])
def test_alternating_ba_matches_oracle_loop(oracle, batched):
    # (scheme end tasks off: the oracle loop below restates the iteration body only)
    scene = scenes.synthetic_scene(4, seed=41, use_depth_residuals=True, use_descriptor_residuals=False)
    rng = np.random.default_rng(4)
    n = scene.surfels_size
    scene.surfels[2, :n] += rng.uniform(-0.002, 0.002, n).astype(np.float32)
    for kf in scene.keyframes:
        x = np.concatenate([rng.uniform(-0.003, 0.003, 3), rng.uniform(-0.001, 0.001, 3)]).astype(np.float32)
        kf.global_T_frame = bso.se3_mul(kf.global_T_frame, bso.se3_exp(x))
    ba = make_ba(scene, batched_pose_optimization=batched, scheme_end_tasks=False)
    covis = {kf.id: ba.keyframe_covisibility(kf.id) for kf in scene.keyframes}
    assert all(len(v) == len(scene.keyframes) - 1 for v in covis.values()), covis   # all frusta overlap in this scene
    # first iteration on its own: identical inputs on both sides -> integer outputs must be identical
    ba.BundleAdjustment(False, False, False, True, True, 1, 1, False, 0, len(scene.keyframes) - 1, True)
    oracle_alternating_iteration(scene, covis)
    first = ba.GetSurfels(8)
    assert np.array_equal(first[3].view(np.uint32), scene.surfels[3, :n].view(np.uint32))
    assert np.array_equal(ba.GetActiveSurfels() & 1, scene.active[0, :n] & 1)
    assert np.abs(first[:3] - scene.surfels[:3, :n]).max() < 1e-6
    iters, conv = ba.BundleAdjustment(False, False, False, True, True, 1, 2, False, 0, len(scene.keyframes) - 1, True)
    ref_iters = 0
    for _ in range(2):
        ref_iters += 1
        if oracle_alternating_iteration(scene, covis):
            break
    assert iters == ref_iters
    for kf in scene.keyframes:
        d = np.abs(bso.se3_to_np(ba.keyframe_pose(kf.id)) - bso.se3_to_np(kf.global_T_frame)).max()
        assert d < 1e-4, (kf.id, d)
        assert ba.keyframe_activation(kf.id) == kf.activation
    got = ba.GetSurfels(8)
    ref = scene.surfels[:8, :n]
    # Over several iterations the poses of the two runs differ in the last bits (different summation
    # order of H/b), so a surfel sitting exactly on an association / 10-bit rounding threshold can
    # fall on the other side: tolerate <= 0.5 % such surfels, everything else must agree.
    normal_mismatch = (got[3].view(np.uint32) != ref[3].view(np.uint32)).mean()
    assert normal_mismatch <= 5e-3, normal_mismatch
    pos_bad = (np.abs(got[:3] - ref[:3]).max(axis=0) > 1e-4 * max(1.0, np.abs(ref[:3]).max())).mean()
    assert pos_bad <= 5e-3, pos_bad
    act_mismatch = ((ba.GetActiveSurfels() & 1) != (scene.active[0, :n] & 1)).mean()
    assert act_mismatch <= 5e-3, act_mismatch


def test_intrinsics_step_matches_oracle(oracle):
    """bslam_optimize_intrinsics vs the oracle's OptimizeIntrinsicsCUDA restatement, one step from a
    distorted depth camera with non-zero a / cfactor (so that every Jacobian column is exercised)."""
    from tests import gpu_util
    true = scenes.intrinsics_test_camera()
    scene = scenes.intrinsics_scene(4, seed=3, width=640, height=480, cell=4, max_surfels=200000, camera=true)
    scene.depth_camera = scenes.distorted_camera(true, 8.0)
    scene.a = 0.01
    scene.cfactor[:, :] = np.random.default_rng(0).uniform(-0.002, 0.002, scene.cfactor.shape).astype(np.float32)
    hip = gpu_util.Hip(scene.to_device())
    out_c, out_d, a = hip.optimize_intrinsics(True, False)
    scene.optimize_intrinsics(True, False)
    ref = scene.depth_camera
    for got, exp in ((out_d.fx, ref.fx), (out_d.fy, ref.fy), (out_d.cx, ref.cx), (out_d.cy, ref.cy)):
        assert abs(got - exp) <= 1e-4 * abs(exp), (got, exp)
    # `a` is the weakly constrained unknown of the 5x5 system (the reference's own test allows 1e-2 on it,
    # BS/test/test_intrinsics_optimization_geometric_residual.cc:347): the solve amplifies the rounding of the
    # fp32 sums (serial on the oracle, tree / atomics on the device) more than for the camera parameters
    assert abs(a - scene.a) <= 1e-3 * max(abs(scene.a), 1e-2), (a, scene.a)
    cf = hip.d.cfactor.cpu().numpy()
    # per-cell updates inherit the sensitivity of `a` (offset -= B_a * x1_a): compared at 1e-3 of the largest cfactor
    cf_err = np.abs(cf - scene.cfactor).max()
    assert cf_err <= 1e-3 * max(np.abs(scene.cfactor).max(), 1e-3), (cf_err, np.abs(scene.cfactor).max())
    assert np.abs(scene.cfactor).max() > 1e-4


def test_color_intrinsics_step_matches_oracle(oracle):
    """The colour-intrinsics step (BS/kernel_opt_intrinsics.cu:130-166,198-216, .cc:251-280) vs the oracle with its H, b summed
    in float64: fx, fy, cx, cy after one step from a camera 4 px off agree to 2e-4 px."""
    from tests import gpu_util
    true = scenes.intrinsics_test_camera()
    scene = scenes.intrinsics_scene(4, seed=3, width=640, height=480, cell=2, max_surfels=400000, photometric=True, camera=true)
    scene.color_camera = scenes.distorted_camera(true, 3.0)
    hip = gpu_util.Hip(scene.to_device())
    out_c, _, _ = hip.optimize_intrinsics(False, True)
    bso.lib().bso_set_intrinsics_sum64(1)
    try:
        scene.optimize_intrinsics(False, True)
    finally:
        bso.lib().bso_set_intrinsics_sum64(0)
    ref = scene.color_camera
    d = scenes.distorted_camera(true, 3.0)
    assert abs(ref.fx - d.fx) > 0.1 and abs(ref.cy - d.cy) > 0.1      # the step moved the camera
    for got, exp in ((out_c.fx, ref.fx), (out_c.fy, ref.fy), (out_c.cx, ref.cx), (out_c.cy, ref.cy)):
        assert abs(got - exp) <= 2e-4, (got, exp)


def intrinsics_test_ba(scene):
    """DirectBA as the two intrinsics tests construct it (merge factor 0.8, observation counts 2 / 2 / 2,
    BS/test/test_intrinsics_optimization_geometric_residual.cc:421-436), keyframes added, then the surfels created from every
    keyframe with the observation filter while the cameras are still the true ones (:515-517)."""
    from badslam_amd.direct_ba import DirectBA
    ba = DirectBA(scene.max_surfels, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, 0.8, 2, 2, 2,
                  scene.color_camera, scene.depth_camera, 0, scene.use_depth_residuals, scene.use_descriptor_residuals)
    ba.set_options(texture_mode=scene.tex_mode, pcg_gauge_keyframe=0)
    for kf in scene.keyframes:
        ba.AddKeyframe(kf.id, max(kf.min_depth, 1e-3), max(kf.max_depth, 1e-2), kf.depth, kf.normals, kf.radius, kf.color, kf.global_T_frame)
    for kf in scene.keyframes:
        ba.CreateSurfelsForKeyframe(True, kf.id)
    return ba


@pytest.mark.parametrize("use_pcg", [False, True])
def test_intrinsics_optimization_with_geometric_residual(oracle, use_pcg):
    """{Alternating,PCG}IntrinsicsOptimizationWithGeometricResidual exactly as the reference runs it
    (BS/test/test_intrinsics_optimization_geometric_residual.cc:371-553): true camera {0.5h, 0.45h, 0.5w - 0.5, 0.5h - 0.5} (:378),
    36 keyframes of 20 planes, filtered surfel creation with the true camera, depth camera then set to true + (+0.5, -0.6,
    +1.23, -2.17) px (:416, :518), 100 x BundleAdjustment(depth intrinsics only, no surfel updates, min 1 / max 10, first call
    without increase_ba_iteration_count); bar 1e-3 px on fx, fy, cx, cy (:539-542)."""
    true = scenes.intrinsics_test_camera()
    scene = scenes.intrinsics_scene(36, seed=0, cell=2, max_surfels=1000 * 1000, camera=true, create_surfels=False)
    ba = intrinsics_test_ba(scene)
    assert ba.surfels_size() > 500000
    d = scenes.distorted_camera(true)
    ba.set_intrinsics(None, [d.fx, d.fy, d.cx, d.cy], 0.0)
    for i in range(100):
        ba.BundleAdjustment(True, False, False, False, False, 1, 10, use_pcg, 0, len(scene.keyframes) - 1, i != 0)
    _, dc, a = ba.intrinsics()
    err = np.abs(dc - np.array([true.fx, true.fy, true.cx, true.cy], np.float32))
    assert err.max() < 1e-3, (err, a)


@pytest.mark.parametrize("use_pcg", [False, True])
def test_intrinsics_optimization_with_photometric_residual(oracle, use_pcg):
    """{Alternating,PCG}IntrinsicsOptimizationWithPhotometricResidual exactly as the reference runs it
    (BS/test/test_intrinsics_optimization_photometric_residual.cc:104-265): true camera {0.5h, 0.45h, ...} (:112), 12 keyframes,
    descriptor residuals only, filtered surfel creation with the true camera (:216-218), colour camera then set to true +
    (+0.5, -0.6, +1.23, -2.17) px (:150, :219), 10 x BundleAdjustment(colour intrinsics only, do_surfel_updates, min 1 / max 10)
    -- 10 single steps, since without pose optimisation the loop ends after min_iterations (BS/direct_ba_alternating.cc:693-700) --
    bars 0.03 px (fx, fy) and 0.15 px (cx, cy) (:262-265)."""
    true = scenes.intrinsics_test_camera()
    scene = scenes.intrinsics_scene(12, seed=0, cell=2, max_surfels=1000 * 1000, photometric=True, camera=true, create_surfels=False)
    ba = intrinsics_test_ba(scene)
    assert ba.surfels_size() > 300000
    created = ba.surfels_size()
    d = scenes.distorted_camera(true)
    ba.set_intrinsics([d.fx, d.fy, d.cx, d.cy], None, 0.0)
    trace = []
    for i in range(10):
        its, _ = ba.BundleAdjustment(False, True, True, False, False, 1, 10, use_pcg, 0, len(scene.keyframes) - 1, i != 0)
        assert its == 1
        trace.append(ba.intrinsics()[0].copy())
    cc = trace[-1]
    err = np.abs(cc - np.array([true.fx, true.fy, true.cx, true.cy], np.float32))
    assert err[0] < 0.03 and err[1] < 0.03 and err[2] < 0.15 and err[3] < 0.15, err
    if not use_pcg:
        # the same schedule on the oracle (tests/oracle_ba.py) with the host class's co-visibility lists, step by step: the two
        # trajectories agree far below the bars
        from tests import oracle_ba
        covis = {kf.id: ba.keyframe_covisibility(kf.id) for kf in scene.keyframes}
        oscene = scenes.intrinsics_scene(12, seed=0, cell=2, max_surfels=1000 * 1000, photometric=True, camera=true, filter_new_surfels=True,
                                         covisibility=covis)
        assert oscene.surfels_size == created
        oscene.color_camera = d
        oba = oracle_ba.OracleAlternatingBA(oscene, covisibility=covis)
        # The oracle's H, b as float64 sums of its fp32 terms: its default serial fp32 sum over ~4e6 terms is itself ~1e-3
        # off (6e-3 px on this step), the device's tree / fp64-atomic sums are not.
        bso.lib().bso_set_intrinsics_sum64(1)
        try:
            for i in range(10):
                oba.bundle_adjustment(False, True, True, False, 1, i != 0)
                oc = oscene.color_camera
                assert np.abs(trace[i] - np.array([oc.fx, oc.fy, oc.cx, oc.cy], np.float32)).max() < 3e-4, (i, trace[i], oc.fx, oc.fy, oc.cx, oc.cy)
        finally:
            bso.lib().bso_set_intrinsics_sum64(0)


def test_save_and_load_calibration_through_direct_ba(oracle, tmp_path):
    """SaveCalibration / LoadCalibration (BS/io.cc:570-700) on the host class: intrinsics, a and the device cfactor image."""
    scene = scenes.synthetic_scene(2, seed=3, cell=4)
    ba = make_ba(scene)
    d = scene.depth_camera
    ba.set_intrinsics([d.fx + 1.5, d.fy - 2.5, d.cx + 0.25, d.cy - 0.75], [d.fx - 0.5, d.fy + 0.5, d.cx + 1.0, d.cy + 2.0], 0.0175)
    ba.SaveCalibration(tmp_path / "c")
    other = make_ba(scene)
    other.LoadCalibration(tmp_path / "c")
    c1, d1, a1 = ba.intrinsics()
    c2, d2, a2 = other.intrinsics()
    assert np.allclose(c1, c2, atol=2e-3) and np.allclose(d1, d2, atol=2e-3) and abs(a1 - a2) < 1e-7   # 6 significant digits in the files
    assert np.array_equal(other.cfactor(scene.cfactor.shape), ba.cfactor(scene.cfactor.shape))


def test_save_and_load_state_through_direct_ba(oracle, tmp_path):
    """The DirectBA part of SaveState / LoadState (BS/io.cc:38-536): after a few BA iterations the state goes to a
    version-1 file; a fresh DirectBA with the same keyframe images continues from it exactly like the original."""
    scene = scenes.synthetic_scene(3, seed=8, cell=4)
    rng = np.random.default_rng(1)
    for kf in scene.keyframes[1:]:
        kf.global_T_frame = bso.se3_mul(kf.global_T_frame, bso.se3_exp(np.concatenate([rng.uniform(-0.003, 0.003, 3), rng.uniform(-0.001, 0.001, 3)]).astype(np.float32)))
    ba = make_ba(scene)
    ba.BundleAdjustment(False, False, False, True, True, 2, 2, False, 0, 2, True)
    ba.SaveState(tmp_path / "s.state", frame_count=3)
    other = make_ba(scene)                      # same images, the scene's (unoptimised) poses and surfels
    other.LoadState(tmp_path / "s.state")
    assert other.surfels_size() == ba.surfels_size()
    assert np.array_equal(other.GetSurfels(8).view(np.uint32), ba.GetSurfels(8).view(np.uint32))
    for k in range(3):
        assert np.array_equal(bso.se3_to_np(other.keyframe_pose(k)), bso.se3_to_np(ba.keyframe_pose(k)))
    # both continue identically
    ba.BundleAdjustment(False, False, False, True, True, 1, 1, False, 0, 2, True)
    other.BundleAdjustment(False, False, False, True, True, 1, 1, False, 0, 2, True)
    assert np.array_equal(other.GetSurfels(8).view(np.uint32), ba.GetSurfels(8).view(np.uint32))
    for k in range(3):
        assert np.array_equal(bso.se3_to_np(other.keyframe_pose(k)), bso.se3_to_np(ba.keyframe_pose(k)))


def test_keyframe_management_colors_and_export(oracle):
    """The rest of DirectBA's outer seam (BS/direct_ba.h:95-126): DeleteKeyframe, UpdateKeyframeCoVisibility,
    MergeKeyframes, AssignColors, ExportToPointCloud -- and BundleAdjustment with a deleted keyframe in the list
    (null entries are skipped, BS/kernel_opt_geometry.cc:115)."""
    scene = scenes.synthetic_scene(6, seed=9, use_depth_residuals=True, use_descriptor_residuals=True)
    K = len(scene.keyframes)
    ba = make_ba(scene)
    # co-visibility from AddKeyframe is symmetric and free of self entries (BS/direct_ba.cc:231-249)
    lists = [ba.keyframe_covisibility(k) for k in range(K)]
    for k in range(K):
        assert k not in lists[k] and all(k in lists[j] for j in lists[k])
    assert all(len(l) > 0 for l in lists)

    # AssignColors through the host class == the oracle on the same scene, byte for byte
    ba.AssignColors()
    scene.assign_colors()
    n = scene.surfels_size
    assert np.array_equal(ba.GetSurfels(8)[5, :n].view(np.uint32), scene.surfels[5, :n].view(np.uint32))

    # DeleteKeyframe: gone from every list; everything that walks the keyframes keeps working
    ba.DeleteKeyframe(2)
    assert ba.keyframe_is_deleted(2) and not ba.keyframe_is_deleted(1)
    assert all(2 not in ba.keyframe_covisibility(k) for k in range(K) if k != 2)
    with pytest.raises(Exception):
        ba.DeleteKeyframe(2)
    before = np.array([dba_pose7(ba, k) for k in range(K) if k != 2])
    ba.BundleAdjustment(False, False, False, True, True, 2, 4, False, 0, K - 1, True)
    after = np.array([dba_pose7(ba, k) for k in range(K) if k != 2])
    surfels = ba.GetSurfels(8)
    assert np.isfinite(after).all() and np.abs(after - before).max() < 1e-2
    assert np.isfinite(surfels[:3, :ba.surfels_size()]).all()
    ba.AssignColors()     # the table without keyframe 2

    # UpdateKeyframeCoVisibility after a pose change: a keyframe moved 100 m away sees nobody any more -- except
    # itself, twice (the reference does not skip the keyframe itself, BS/direct_ba.cc:724-735)
    T = ba.keyframe_pose(3)
    T.t[0] += 100.0
    ba.set_keyframe_pose(3, T)
    ba.UpdateKeyframeCoVisibility(3)
    assert ba.keyframe_covisibility(3) == [3, 3]
    assert all(3 not in ba.keyframe_covisibility(k) for k in range(K) if k not in (2, 3))

    # ExportToPointCloud: the non-NaN surfels with colours and re-normalised normals
    pos, col, nrm = ba.ExportToPointCloud()
    rows = ba.GetSurfels(8)[:, :ba.surfels_size()]
    valid = ~np.isnan(rows[0])
    assert len(pos) == valid.sum() == len(col) == len(nrm)
    assert np.array_equal(pos, rows[:3, valid].T)
    packed = rows[5, valid].view(np.uint32)
    assert np.array_equal(col, np.stack([packed & 0xff, (packed >> 8) & 0xff, (packed >> 16) & 0xff], axis=1).astype(np.uint8))
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1).max() < 1e-6


def dba_pose7(ba, k):
    from badslam_amd import direct_ba as dba
    return dba.pose7(ba.keyframe_pose(k))


def test_merge_keyframes_deletes_the_closest_inner_keyframes(oracle):
    """MergeKeyframes (BS/direct_ba.cc:251-338): candidates are keyframes within 0.3 m and 45 degrees of the next one,
    ranked by distance to previous + next; keyframe 0 is never deleted, a keyframe whose neighbour went in the same call
    is skipped."""
    scene = scenes.synthetic_scene(7, seed=10, use_depth_residuals=True, use_descriptor_residuals=False)
    ba = make_ba(scene)
    base = ba.keyframe_pose(0)
    # keyframes on a line: gaps 0.10, 0.02, 0.03, 0.25, 0.05, 0.50 m (the last gap is beyond the 0.3 m limit)
    x = np.cumsum([0.0, 0.10, 0.02, 0.03, 0.25, 0.05, 0.50])
    for k in range(7):
        T = ba.keyframe_pose(0)
        T.t[0] = base.t[0] + float(x[k])
        ba.set_keyframe_pose(k, T)
    # candidates (id: prev gap + next gap): 1: 0.12, 2: 0.05, 3: 0.28, 4: 0.30; 5 has no valid next gap (0.5 m), 6 is last
    deleted = ba.MergeKeyframes(2)
    # sorted: 2 (0.05), 1 (0.12): 2 goes first, then 1 is skipped because its next keyframe (2) is gone
    assert deleted == [2]
    assert ba.keyframe_is_deleted(2) and not ba.keyframe_is_deleted(0)
    deleted = ba.MergeKeyframes(10)
    # now 1: 0.10 + 0.05 = 0.15, 3: 0.05 + 0.25 = 0.30, 4: 0.25 + 0.05 = 0.30 -> 1 goes, then 3 (prev 1 gone -> skipped), 4 (prev 3 alive, next 5 alive) goes
    assert 0 not in deleted and 6 not in deleted and 1 in deleted
    assert all(ba.keyframe_is_deleted(k) for k in deleted)
    assert ba.MergeKeyframes(0) == []

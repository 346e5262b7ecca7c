"""-m gpu parity tests: HIP path (through the C ABI) vs the CPU oracle on identical inputs.

Bars (BASELINE.json north_star): association pixels / activation masks bit-exact; residuals,
H/b coefficients and optimised poses within 1e-4 relative (tolerances written at each assert).
"""
import ctypes as C

import numpy as np

import badslam_amd
import pytest

from badslam_amd import abi
from tests import bso, scenes

pytestmark = pytest.mark.gpu

REL = 1e-4   # north_star tolerance for floating-point outputs


def rel_close(a, b, scale=None, rel=REL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = np.abs(b).max() if scale is None else scale
    return np.abs(a - b).max() <= rel * max(scale, 1e-30)


def pose_error(est, gt):
    return bso.se3_log(bso.se3_mul(bso.se3_inverse(est), gt))


@pytest.fixture(scope="module")
def hip_mod():
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need a GPU"
    from tests import gpu_util
    return gpu_util


@pytest.fixture(scope="module")
def geo(hip_mod, oracle):
    scene, kf = scenes.pose_geometric_scene(seed=0)
    return scene, kf, hip_mod.Hip(scene.to_device())


@pytest.fixture(scope="module")
def photo(hip_mod, oracle):
    scene, kf, gt = scenes.pose_photometric_scene(seed=0)
    return scene, kf, gt, hip_mod.Hip(scene.to_device())


@pytest.fixture(scope="module")
def multi(hip_mod, oracle):
    scene = scenes.synthetic_scene(6, seed=3, use_depth_residuals=True, use_descriptor_residuals=True)
    return scene, hip_mod.Hip(scene.to_device())


def perturbed(kf, x):
    """frame_T_global of global_T_frame * exp(x)."""
    T = bso.se3_mul(kf.global_T_frame, bso.se3_exp(x))
    return T, bso.se3_matrix3x4(bso.se3_inverse(T))


# ----------------------------------------------------------------------------- association

def test_association_bit_exact_single_keyframe(geo):
    scene, kf, hip = geo
    ref = scene.association(kf)
    got = hip.association(0)
    assert (ref != 0xFFFFFFFF).sum() > 100000
    assert np.array_equal(ref, got)


def test_association_bit_exact_multi_keyframe(multi):
    scene, hip = multi
    total = 0
    for k, kf in enumerate(scene.keyframes):
        ref = scene.association(kf)
        got = hip.association(k)
        total += int((ref != 0xFFFFFFFF).sum())
        mism = int((ref != got).sum())
        assert mism == 0, f"keyframe {k}: {mism} of {ref.size} association results differ"
    assert total > scene.surfels_size   # surfels are seen by several keyframes


# ----------------------------------------------------------------------------- residuals, H/b

def test_depth_residuals_per_surfel(geo):
    scene, kf, hip = geo
    _, M = perturbed(kf, np.array([0.004, -0.003, 0.002, 0.001, -0.0005, 0.0008], np.float32))
    # probe uses the keyframe's own pose; check that one, then H/b at the perturbed pose below
    ref = scene.accumulate_pose(kf, per_surfel=True)["per_surfel"]
    got = hip.residual_probe(0)
    assert np.array_equal(ref[:, 6], got[:, 6])
    assert rel_close(got[:, 0], ref[:, 0], scale=max(1.0, np.abs(ref[:, 0]).max()))
    assert rel_close(got[:, 1], ref[:, 1], scale=1.0)


def test_pose_coefficients_geometric(geo):
    scene, kf, hip = geo
    _, M = perturbed(kf, np.array([0.004, -0.003, 0.002, 0.001, -0.0005, 0.0008], np.float32))
    ref = scene.accumulate_pose(kf, frame_T_global=M)
    got = hip.accumulate_pose(0, frame_T_global=M)
    assert got["count"] == ref["count"] and ref["count"] > 100000
    # against the float64 sum of the oracle's fp32 terms, relative to the largest coefficient
    assert rel_close(got["H"], ref["H64"])
    assert rel_close(got["b"], ref["b64"])
    assert rel_close(got["cost"], ref["cost"], scale=max(1.0, abs(ref["cost"])), rel=1e-3)   # fp32 serial sum of 1e5 terms on the oracle side


def test_pose_coefficients_photometric(photo):
    scene, kf, gt, hip = photo
    _, M = perturbed(kf, np.array([0.0004, -0.0003, 0.0002, 0.0008, -0.0005, 0.0006], np.float32))
    ref = scene.accumulate_pose(kf, frame_T_global=M, per_surfel=True)
    got = hip.accumulate_pose(0, frame_T_global=M)
    assert got["count"] == ref["count"] and ref["count"] > 250000
    assert rel_close(got["H"], ref["H64"])
    assert rel_close(got["b"], ref["b64"])


def test_photometric_residuals_per_surfel(photo):
    scene, kf, gt, hip = photo
    ref = scene.accumulate_pose(kf, per_surfel=True)["per_surfel"]
    got = hip.residual_probe(0)
    assert np.array_equal(ref[:, 6], got[:, 6])
    scale = max(1.0, np.abs(ref[:, [2, 4]]).max())
    assert rel_close(got[:, 2], ref[:, 2], scale=scale)
    assert rel_close(got[:, 4], ref[:, 4], scale=scale)
    assert rel_close(got[:, 3], ref[:, 3], scale=1e-2)
    assert rel_close(got[:, 5], ref[:, 5], scale=1e-2)


def test_batched_coefficients_match_per_keyframe_oracle(multi):
    scene, hip = multi
    Hb, counts = hip.accumulate_pose_batched()
    for k, kf in enumerate(scene.keyframes):
        ref = scene.accumulate_pose(kf)
        assert counts[k] == ref["count"]
        assert rel_close(Hb[k, :21], ref["H64"]), k
        assert rel_close(Hb[k, 21:], ref["b64"], scale=np.abs(ref["b64"]).max() + 1e-3 * np.abs(ref["H64"]).max()), k


def test_batched_equals_single_keyframe_entry_point(multi):
    """The reference-shaped per-keyframe entry point and the batched one run the same kernel: deterministic sums, and -- in
    the same work order (index order; by default long keyframe lists use a per-surfel Morton order, short ones a granule
    order) -- identical bits."""
    scene, hip = multi
    Hb, counts = hip.accumulate_pose_batched()
    for k in range(len(scene.keyframes)):
        one = hip.accumulate_pose(k)
        assert rel_close(one["H"], Hb[k, :21], rel=1e-5) and one["count"] == counts[k]
    badslam_amd.check(hip.L.bslam_set_xcd_schedule(hip.ctx.handle, 0))
    try:
        Hi, ci = hip.accumulate_pose_batched()
        for k in range(len(scene.keyframes)):
            one = hip.accumulate_pose(k)
            assert np.array_equal(one["H"], Hi[k, :21]) and np.array_equal(one["b"], Hi[k, 21:])
        assert np.array_equal(ci, counts)
    finally:
        badslam_amd.check(hip.L.bslam_set_xcd_schedule(hip.ctx.handle, 1))
    Hb2, _ = hip.accumulate_pose_batched()
    assert np.array_equal(Hb, Hb2), "sums must be reproducible run to run"


# ----------------------------------------------------------------------------- pose Gauss-Newton

def test_known_answer_pose_geometric_through_hip(geo):
    """BS/test/test_pose_optimization_geometric_residual.cc:134-170 with the HIP path."""
    scene, kf, hip = geo
    gt = bso.se3_identity()
    worst = 0.0
    for off in scenes.offsets_13(0.005, 0.001):
        init = bso.se3_mul(off, bso.se3_inverse(gt))
        poses, iters, conv = hip.estimate_poses_batched([init])
        ref, ref_iters, ref_conv = scene.estimate_frame_pose(kf, init)
        err = pose_error(poses[0], gt)
        worst = max(worst, float(np.abs(err).max()))
        assert conv[0] == 1 and iters[0] <= 30
        # optimised pose vs the oracle's: 1e-4 relative on the 7 SE3 numbers (|q| = 1, |t| ~ 5e-3..)
        assert np.abs(bso.se3_to_np(poses[0]) - bso.se3_to_np(ref)).max() < 1e-6
    assert worst < 1.1e-6, worst


def test_known_answer_pose_photometric_through_hip(photo):
    """BS/test/test_pose_optimization_photometric_residual.cc:141-177 with the HIP path."""
    scene, kf, gt, hip = photo
    worst = 0.0
    for off in scenes.offsets_13(0.0005, 0.001)[:5]:
        init = bso.se3_mul(gt, off)
        poses, iters, conv = hip.estimate_poses_batched([init])
        err = pose_error(poses[0], gt)
        worst = max(worst, float(np.abs(err).max()))
    assert worst < 8e-5, worst


def test_batched_pose_estimation_matches_sequential_oracle(multi):
    scene, hip = multi
    rng = np.random.default_rng(5)
    inits = []
    for kf in scene.keyframes:
        x = np.concatenate([rng.uniform(-0.004, 0.004, 3), rng.uniform(-0.001, 0.001, 3)]).astype(np.float32)
        inits.append(bso.se3_mul(kf.global_T_frame, bso.se3_exp(x)))
    poses, iters, conv = hip.estimate_poses_batched(inits)
    for k, kf in enumerate(scene.keyframes):
        ref, ref_iters, ref_conv = scene.estimate_frame_pose(kf, inits[k])
        d = np.abs(bso.se3_to_np(poses[k]) - bso.se3_to_np(ref)).max()
        assert d < 1e-4 * max(1.0, np.abs(bso.se3_to_np(ref)).max()), (k, d)
        assert conv[k] == int(ref_conv)


def test_inactive_keyframes_are_not_optimised(multi):
    scene, hip = multi
    inits = [bso.se3_mul(kf.global_T_frame, bso.se3_exp(np.array([0.003, 0, 0, 0, 0, 0], np.float32))) for kf in scene.keyframes]
    acts = [abi.KF_INACTIVE if k % 2 else abi.KF_ACTIVE for k in range(len(inits))]
    poses, iters, conv = hip.estimate_poses_batched(inits, activations=acts)
    for k in range(len(inits)):
        if acts[k] == abi.KF_INACTIVE:
            assert iters[k] == 0 and np.array_equal(bso.se3_to_np(poses[k]), bso.se3_to_np(inits[k]))
        else:
            assert iters[k] > 0


# ----------------------------------------------------------------------------- activation / geometry

def fresh_multi(hip_mod, use_desc, seed=11, K=4):
    scene = scenes.synthetic_scene(K, seed=seed, use_depth_residuals=True, use_descriptor_residuals=use_desc)
    # move the surfels off their surfaces a little so that the geometry step has work to do
    rng = np.random.default_rng(seed)
    n = scene.surfels_size
    scene.surfels[2, :n] += rng.uniform(-0.004, 0.004, n).astype(np.float32)
    return scene, hip_mod.Hip(scene.to_device())


def test_activation_mask_bit_exact(hip_mod, oracle):
    scene, hip = fresh_multi(hip_mod, False)
    scene.keyframes[1].activation = abi.KF_INACTIVE
    scene.keyframes[2].activation = abi.KF_COVISIBLE_ACTIVE
    scene.active[0, :scene.surfels_size] = np.random.default_rng(0).integers(0, 4, scene.surfels_size).astype(np.uint8)
    hip.d.active.copy_(hip.torch.from_numpy(scene.active))
    scene.update_activation()
    hip.update_activation()
    got = hip.d.active_np()
    assert np.array_equal(got[0, :scene.surfels_size], scene.active[0, :scene.surfels_size])
    assert 0 < (got[0, :scene.surfels_size] & 1).sum() < scene.surfels_size


@pytest.mark.parametrize("use_desc", [False, True])
def test_geometry_iteration_matches_oracle(hip_mod, oracle, use_desc):
    scene, hip = fresh_multi(hip_mod, use_desc)
    scene.keyframes[3].activation = abi.KF_INACTIVE
    scene.active[0, ::7] = 0
    hip.d.active.copy_(hip.torch.from_numpy(scene.active))
    before = scene.surfels[:8, :scene.surfels_size].copy()
    scene.optimize_geometry_iteration()
    hip.optimize_geometry_iteration()
    got = hip.d.surfels_np()[:8, :scene.surfels_size]
    ref = scene.surfels[:8, :scene.surfels_size]
    # normals are integers (3 x s10): bit-exact
    assert np.array_equal(got[3].view(np.uint32), ref[3].view(np.uint32))
    assert not np.array_equal(ref[:3], before[:3]), "the step must move surfels"
    # positions / descriptors: same fp32 operations in the same order -> expect identical bits;
    # the hard bar is 1e-4 relative
    for row in (0, 1, 2, 6, 7):
        assert rel_close(got[row], ref[row], scale=max(1.0, np.abs(ref[row]).max())), row
    exact = sum(np.array_equal(got[r].view(np.uint32), ref[r].view(np.uint32)) for r in (0, 1, 2, 6, 7))
    print("bit-exact rows:", exact, "of 5")
    # inactive surfels untouched
    assert np.array_equal(got[:, ::7].view(np.uint32), before[:, ::7].view(np.uint32))


def test_update_normals_only(hip_mod, oracle):
    scene, hip = fresh_multi(hip_mod, False, seed=12)
    scene.update_normals()
    hip.update_normals()
    got = hip.d.surfels_np()[:8, :scene.surfels_size].view(np.uint32)
    ref = scene.surfels[:8, :scene.surfels_size].view(np.uint32)
    diff = {r: int((got[r] != ref[r]).sum()) for r in range(8)}
    bad = np.nonzero(got[3] != ref[3])[0][:5]
    detail = [(int(i), hex(int(got[3, i])), hex(int(ref[3, i]))) for i in bad]
    assert not any(diff.values()), (diff, detail)


# ----------------------------------------------------------------------------- edge cases

def test_argument_checks(multi):
    import badslam_amd
    scene, hip = multi
    dp, sb = hip.d.depth_params(), hip.d.surfel_buf()
    v = hip.d.keyframe_view(0)
    H, b = np.zeros(21, np.float32), np.zeros(6, np.float32)
    f = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    L = badslam_amd.lib()
    from tests.gpu_util import stream_ptr
    # neither residual type: the reference CHECK()s (BS/kernel_opt_pose.cc:58)
    rc = L.bslam_accumulate_pose_estimation_coeffs(hip.ctx.handle, stream_ptr(), 0, 0, C.byref(scene.color_camera), C.byref(scene.depth_camera),
                                                   C.byref(dp), C.byref(v.depth), C.byref(v.normals), C.byref(v.color), C.byref(v.frame_T_global),
                                                   scene.surfels_size, C.byref(sb), 0, None, None, f(H), f(b))
    assert rc == -1
    # surfels_size == 0: CHECK_GT (BS/kernel_opt_pose.cc:61)
    rc = L.bslam_accumulate_pose_estimation_coeffs(hip.ctx.handle, stream_ptr(), 1, 0, C.byref(scene.color_camera), C.byref(scene.depth_camera),
                                                   C.byref(dp), C.byref(v.depth), C.byref(v.normals), C.byref(v.color), C.byref(v.frame_T_global),
                                                   0, C.byref(sb), 0, None, None, f(H), f(b))
    assert rc == -1
    # surfels_size larger than the buffer must be refused, not launched
    rc = L.bslam_accumulate_pose_estimation_coeffs(hip.ctx.handle, stream_ptr(), 1, 0, C.byref(scene.color_camera), C.byref(scene.depth_camera),
                                                   C.byref(dp), C.byref(v.depth), C.byref(v.normals), C.byref(v.color), C.byref(v.frame_T_global),
                                                   scene.max_surfels + 1, C.byref(sb), 0, None, None, f(H), f(b))
    assert rc == -1


def test_ragged_surfel_count(hip_mod, oracle):
    """surfels_size not a multiple of the tile: the tail threads must not contribute."""
    scene = scenes.synthetic_scene(2, seed=21, use_depth_residuals=True, use_descriptor_residuals=False)
    scene.surfels_size = scene.surfels_size - 333
    hip = hip_mod.Hip(scene.to_device())
    for k, kf in enumerate(scene.keyframes):
        ref = scene.accumulate_pose(kf)
        got = hip.accumulate_pose(k)
        assert got["count"] == ref["count"]
        assert rel_close(got["H"], ref["H64"])


def test_normal_decode_is_correctly_rounded_over_the_whole_domain():
    """All 65536 u16 normal codes through the kernels' decode (specialised correctly rounded sqrt) against an
    independent evaluation of BS/util.cuh:120-130 in the oracle's shape: z = -sqrt(max(0, fma(-y, y, fma(-x, x, 1))))."""
    import ctypes as C
    import torch
    import badslam_amd
    L = badslam_amd.lib()
    ctx = badslam_amd.Context(0)
    out = np.zeros((65536, 3), np.float32)
    badslam_amd.check(L.bslam_debug_decode_normals(ctx.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                   out.ctypes.data_as(C.POINTER(C.c_float))))
    codes = np.arange(65536, dtype=np.uint32)
    x = (codes & 0xff).astype(np.uint8).view(np.int8).astype(np.float32) * np.float32(1.0 / 127)
    y = ((codes >> 8) & 0xff).astype(np.uint8).view(np.int8).astype(np.float32) * np.float32(1.0 / 127)
    # exact products / sums in 64-bit-mantissa long double (x^2 is a multiple of 2^-60), one rounding per fma
    xl, yl = x.astype(np.longdouble), y.astype(np.longdouble)
    assert np.finfo(np.longdouble).nmant >= 63
    t = (np.longdouble(1) - xl * xl).astype(np.float32)
    z2 = (t.astype(np.longdouble) - yl * yl).astype(np.float32)
    z = -np.sqrt(np.maximum(z2, np.float32(0)))
    assert np.array_equal(out[:, 0].view(np.uint32), x.view(np.uint32))
    assert np.array_equal(out[:, 1].view(np.uint32), y.view(np.uint32))
    assert np.array_equal(out[:, 2].view(np.uint32), z.astype(np.float32).view(np.uint32))
    assert (z2 > 0).sum() > 40000 and (z2 <= 0).sum() > 10000      # both branches of the clamp are exercised


@pytest.mark.parametrize("tex_mode", ["fixed", "float"])
def test_assign_colors_bit_exact(oracle, tex_mode):
    """AssignColorsCUDA (BS/kernel_assign_colors.cu): the uchar4 colour row after one launch over all keyframes equals
    the oracle's per-keyframe accumulation byte for byte; surfels nobody observes keep their colour."""
    from badslam_amd import abi
    from tests import gpu_util
    mode = abi.TEX_FIXED_POINT_1_8 if tex_mode == "fixed" else abi.TEX_EXACT_FLOAT
    scene = scenes.synthetic_scene(4, seed=5, use_depth_residuals=True, use_descriptor_residuals=True, tex_mode=mode)
    n = scene.surfels_size
    # a few surfels far away from everything: no observation, colour must survive
    scene.surfels[0:3, n - 50:n] += 100.0
    sentinel = np.uint32(0x11223344)
    scene.surfels[5, :n] = np.full(n, sentinel, np.uint32).view(np.float32)
    # the keyframe colour images carry luma in all channels in the generator; make the channels differ
    rng = np.random.default_rng(0)
    for kf in scene.keyframes:
        kf.color[..., 0] = (kf.color[..., 3].astype(np.int32) * 3 // 4).astype(np.uint8)
        kf.color[..., 1] = 255 - kf.color[..., 3]
        kf.color[..., 2] = rng.integers(0, 256, kf.color.shape[:2], dtype=np.uint8)
    hip = gpu_util.Hip(scene.to_device("cuda:0"))     # the texture mode travels with the host scene
    hip.assign_colors()
    got = hip.d.surfels_np()[5, :n].view(np.uint32)
    scene.assign_colors()
    want = scene.surfels[5, :n].view(np.uint32)
    assert np.array_equal(got, want)
    changed = want != sentinel
    assert changed.sum() > 0.9 * (n - 50) and not changed[n - 50:].any()
    assert len(np.unique(want[changed] & 0xff)) > 50 and len(np.unique((want[changed] >> 16) & 0xff)) > 50


def test_keyframe_without_any_association_keeps_its_pose(oracle):
    """A keyframe that sees none of the surfels: H = 0, b = 0.  Eigen's pivoted LDLT of the zero matrix solves to x = 0
    (BS/direct_ba_alternating.cc:206), so the pose must come back unchanged and finite, flagged converged after one
    iteration -- in the oracle and in the batched HIP loop alike, next to keyframes that do move."""
    from tests import gpu_util
    scene = scenes.synthetic_scene(3, seed=8, use_depth_residuals=True, use_descriptor_residuals=False)
    far = scene.keyframes[2]
    away = bso.se3_exp(np.array([50.0, 0, 0, 0, 0, 0], np.float32))
    far.global_T_frame = bso.se3_mul(far.global_T_frame, away)
    inits = [bso.se3_mul(kf.global_T_frame, bso.se3_exp(np.array([0.003, -0.002, 0.001, 0.0005, 0, -0.0005], np.float32))) for kf in scene.keyframes]
    inits[2] = far.global_T_frame
    hip = gpu_util.Hip(scene.to_device("cuda:0"))
    assert (hip.association(2) == 0xffffffff).all() or hip.accumulate_pose(2)["count"] == 0
    poses, iters, conv = hip.estimate_poses_batched(inits)
    got = bso.se3_to_np(poses[2])
    assert np.isfinite(got).all() and np.array_equal(got, bso.se3_to_np(inits[2]))
    assert conv[2] == 1 and iters[2] == 1
    ref, ref_it, ref_conv = scene.estimate_frame_pose(far, inits[2])
    assert np.array_equal(bso.se3_to_np(ref), bso.se3_to_np(inits[2])) and ref_conv and ref_it == 1
    for k in (0, 1):   # the others are optimised as usual
        assert conv[k] == 1 and iters[k] > 1


def test_color_camera_with_its_own_resolution(oracle):
    """Colour camera at twice the depth camera's resolution (the reference supports separate depth / colour intrinsics
    and sizes, BS/surfel_projection.cuh:184-207): the depth-to-colour transform, the luma-quad table of the larger image and
    the colour bounds test all have to use the colour camera's own dimensions."""
    from tests import gpu_util
    cam = bso.make_camera(262.5, 262.5, 160.0, 120.0, 320, 240)
    scene = scenes.synthetic_scene(3, seed=12, width=320, height=240, camera=cam, use_depth_residuals=True, use_descriptor_residuals=True, color_scale=2)
    assert scene.keyframes[0].color.shape == (480, 640, 4) and scene.keyframes[0].depth.shape == (240, 320)
    hip = gpu_util.Hip(scene.to_device("cuda:0"))
    for k, kf in enumerate(scene.keyframes):
        assert np.array_equal(scene.association(kf), hip.association(k))
        ref = scene.accumulate_pose(kf)
        got = hip.accumulate_pose(k)
        assert got["count"] == ref["count"] and ref["count"] > 5000
        assert np.abs(got["H"] - ref["H64"]).max() <= 1e-4 * np.abs(ref["H64"]).max()
        assert np.abs(got["b"] - ref["b64"]).max() <= 1e-4 * max(np.abs(ref["b64"]).max(), 1e-3 * np.abs(ref["H64"]).max())
    # colours and the joint position + descriptor step
    hip.assign_colors()
    scene.assign_colors()
    n = scene.surfels_size
    assert np.array_equal(hip.d.surfels_np()[5, :n].view(np.uint32), scene.surfels[5, :n].view(np.uint32))
    scene.update_activation()
    hip.update_activation()
    scene.optimize_geometry_iteration()
    hip.optimize_geometry_iteration()
    got = hip.d.surfels_np()[:8, :n]
    assert np.abs(got[:3] - scene.surfels[:3, :n]).max() < 1e-5
    assert np.abs(got[6:8] - scene.surfels[6:8, :n]).max() < 1e-3      # descriptors live on a scale of 180


def test_many_small_keyframes(oracle):
    """K = 70 keyframes of 160 x 120: nothing in the batched kernels may assume K <= 64 (keyframe chunks per block, the
    per-keyframe reduce blocks, the convergence counter) or a 640 x 480 image."""
    from tests import gpu_util
    cam = bso.make_camera(131.25, 131.25, 80.0, 60.0, 160, 120)
    scene = scenes.synthetic_scene(70, seed=13, width=160, height=120, camera=cam, use_depth_residuals=True, use_descriptor_residuals=False)
    K = len(scene.keyframes)
    hip = gpu_util.Hip(scene.to_device("cuda:0"))
    Hb, counts = hip.accumulate_pose_batched()
    for k in (0, 1, 33, 63, 64, 65, 69):
        ref = scene.accumulate_pose(scene.keyframes[k])
        assert counts[k] == ref["count"]
        assert np.abs(Hb[k, :21] - ref["H64"]).max() <= 1e-4 * np.abs(ref["H64"]).max()
    rng = np.random.default_rng(1)
    inits = [bso.se3_mul(kf.global_T_frame, bso.se3_exp(np.concatenate([rng.choice([-1, 1], 3) * 0.003, rng.choice([-1, 1], 3) * 0.0007]).astype(np.float32)))
             for kf in scene.keyframes]
    poses, iters, conv = hip.estimate_poses_batched(inits)
    assert all(conv) and max(iters) < 30
    for k in (0, 63, 64, 69):
        ref, _, _ = scene.estimate_frame_pose(scene.keyframes[k], inits[k])
        assert np.abs(bso.se3_to_np(poses[k]) - bso.se3_to_np(ref)).max() < 1e-5
    scene.update_activation()
    hip.update_activation()
    assert np.array_equal(hip.d.active_np()[0, :scene.surfels_size], scene.active[0, :scene.surfels_size])
    # the geometry iteration twice from the same state: in one launch over all 70 keyframes, and in launches of 16
    # keyframes with the per-surfel sums carried between them -- bit-identical to each other, and equal to the oracle
    n = scene.surfels_size
    start = hip.d.surfels.clone()
    badslam_amd.check(hip.L.bslam_set_geometry_keyframe_chunk(hip.ctx.handle, 0))
    hip.optimize_geometry_iteration()
    single = hip.d.surfels_np()[:8, :n].copy()
    hip.d.surfels.copy_(start)
    badslam_amd.check(hip.L.bslam_set_geometry_keyframe_chunk(hip.ctx.handle, 16))
    hip.optimize_geometry_iteration()
    chunked = hip.d.surfels_np()[:8, :n]
    assert np.array_equal(single.view(np.uint32), chunked.view(np.uint32))
    scene.optimize_geometry_iteration()
    assert np.array_equal(chunked[3].view(np.uint32), scene.surfels[3, :n].view(np.uint32))      # packed normals: bit-exact
    assert np.abs(chunked[:3] - scene.surfels[:3, :n]).max() < 1e-5


def test_photometric_geometry_chunked_equals_single_launch(oracle):
    """The photometric (position + descriptor) geometry iteration walked in keyframe chunks with the per-surfel sums carried in
    scratch is bit-identical to one launch over the whole list -- for chunk sizes that divide the keyframe list unevenly, with
    and without the block-level frustum culling -- and agrees with the oracle."""
    from tests import gpu_util
    cam = bso.make_camera(131.25, 131.25, 80.0, 60.0, 160, 120)
    scene = scenes.synthetic_scene(37, seed=17, width=160, height=120, camera=cam, use_depth_residuals=True, use_descriptor_residuals=True)
    hip = gpu_util.Hip(scene.to_device("cuda:0"))
    n = scene.surfels_size
    scene.update_activation()
    hip.update_activation()
    start = hip.d.surfels.clone()
    L, h = hip.L, hip.ctx.handle
    results = {}
    for name, culling, chunk in (("legacy", 0, 0), ("chunk16", 1, 16), ("chunk5", 0, 5), ("one chunk", 1, 0)):
        hip.d.surfels.copy_(start)
        badslam_amd.check(L.bslam_set_culling(h, culling))
        badslam_amd.check(L.bslam_set_geometry_keyframe_chunk(h, chunk))
        hip.optimize_geometry_iteration()
        results[name] = hip.d.surfels_np()[:8, :n].copy()
    badslam_amd.check(L.bslam_set_culling(h, 1))
    badslam_amd.check(L.bslam_set_geometry_keyframe_chunk(h, -1))
    for name in ("chunk16", "chunk5", "one chunk"):
        assert np.array_equal(results[name].view(np.uint32), results["legacy"].view(np.uint32)), name
    moved = np.abs(results["legacy"][:3] - start.cpu().numpy()[:3, :n]).max()
    assert moved > 1e-5                                              # the iteration did something
    scene.optimize_geometry_iteration()
    got = results["chunk16"]
    assert np.array_equal(got[3].view(np.uint32), scene.surfels[3, :n].view(np.uint32))      # packed normals: bit-exact
    assert np.abs(got[:3] - scene.surfels[:3, :n]).max() < 1e-5
    assert np.abs(got[6:8] - scene.surfels[6:8, :n]).max() < 2e-3    # descriptors (range +-180)


def test_per_surfel_work_order_is_transparent(multi):
    """Calls with >= 4 keyframes walk the surfels in the library's own order (Morton order of the positions, on a sorted copy of
    the rows, include/badslam_hip.h: bslam_set_xcd_schedule).  What is computed per surfel must not depend on it at all --
    activation flags, normals, positions, descriptors: identical bits with and without -- and the per-keyframe sums only in
    their summation order."""
    scene, hip = multi
    assert scene.surfels_size >= 64 * 256 and len(scene.keyframes) >= 4          # large enough for the order to be used
    start = hip.d.surfels.clone()
    out = {}
    for enable in (1, 0):
        badslam_amd.check(hip.L.bslam_set_xcd_schedule(hip.ctx.handle, enable))
        hip.d.surfels.copy_(start)
        hip.update_activation()
        hip.optimize_geometry_iteration()
        Hb, counts = hip.accumulate_pose_batched()
        out[enable] = (hip.d.active_np()[0, :scene.surfels_size].copy(), hip.d.surfels_np()[:8, :scene.surfels_size].copy(), Hb, counts)
    badslam_amd.check(hip.L.bslam_set_xcd_schedule(hip.ctx.handle, 1))
    hip.d.surfels.copy_(start)
    assert np.array_equal(out[1][0], out[0][0])
    assert np.array_equal(out[1][1].view(np.uint32), out[0][1].view(np.uint32))
    assert np.abs(out[1][1][:3] - start.cpu().numpy()[:3, :scene.surfels_size]).max() > 1e-6    # the iteration moved the surfels
    assert np.array_equal(out[1][3], out[0][3])
    for k in range(len(scene.keyframes)):
        assert rel_close(out[1][2][k, :21], out[0][2][k, :21], rel=1e-5), k


def test_work_order_with_deleted_surfels(multi):
    """Deleted surfels (x = NaN, BS/kernel_delete_surfels.cu) stay in the buffer until compaction: they sort to the end of the work
    order, are never associated, and are left untouched -- with the per-surfel order as without it."""
    import torch
    scene, hip = multi
    start = hip.d.surfels.clone()
    n = scene.surfels_size
    dead = torch.arange(0, n, 17, device=hip.d.surfels.device)
    out = {}
    try:
        for enable in (1, 0):
            badslam_amd.check(hip.L.bslam_set_xcd_schedule(hip.ctx.handle, enable))
            hip.d.surfels.copy_(start)
            hip.d.surfels[0, dead] = float("nan")
            hip.update_activation()
            hip.optimize_geometry_iteration()
            Hb, counts = hip.accumulate_pose_batched()
            out[enable] = (hip.d.active_np()[0, :n].copy(), hip.d.surfels_np()[:8, :n].copy(), Hb, counts)
    finally:
        badslam_amd.check(hip.L.bslam_set_xcd_schedule(hip.ctx.handle, 1))
        hip.d.surfels.copy_(start)
    dead = dead.cpu().numpy()
    assert np.array_equal(out[1][0], out[0][0]) and not out[1][0][dead].any()                       # never activated
    assert np.array_equal(out[1][1].view(np.uint32), out[0][1].view(np.uint32))
    assert np.isnan(out[1][1][0, dead]).all()
    assert np.array_equal(out[1][1][1:, dead].view(np.uint32), start.cpu().numpy()[1:8, dead].view(np.uint32))   # untouched
    assert np.array_equal(out[1][3], out[0][3]) and out[1][3].min() > 1000
    assert np.isfinite(out[1][2]).all()


def test_photometric_gauss_newton_converges_alike_in_both_texture_modes(oracle):
    """The batched photometric pose loop from 5 mm / 1 mrad perturbations: converged flags set well inside the cap of 30
    (BS/direct_ba_alternating.cc:130), the rendered poses recovered, and -- the 1/256 weight quantisation of the fixed-point
    texture mode being the one unpinned piece of the sampling arithmetic -- the same iteration counts and the same poses to 1e-6
    with exact float weights: the loop's convergence does not hinge on that guess."""
    from badslam_amd import abi
    from tests import gpu_util
    results = {}
    for name, mode in (("fixed", abi.TEX_FIXED_POINT_1_8), ("exact", abi.TEX_EXACT_FLOAT)):
        scene = scenes.synthetic_scene(4, seed=5, use_depth_residuals=True, use_descriptor_residuals=True, tex_mode=mode)
        hip = gpu_util.Hip(scene.to_device("cuda:0"))
        rng = np.random.default_rng(1)
        inits = [bso.se3_mul(kf.global_T_frame, bso.se3_exp(np.concatenate([rng.choice([-1, 1], 3) * 0.005, rng.choice([-1, 1], 3) * 0.001]).astype(np.float32)))
                 for kf in scene.keyframes]
        errs = []
        for cap in (1, 2, 4, 30):
            poses, iters, conv = hip.estimate_poses_batched(inits, max_iterations=cap)
            errs.append(max(np.abs(pose_error(p, kf.global_T_frame)).max() for p, kf in zip(poses, scene.keyframes)))
        assert all(conv) and max(iters) <= 10, (name, iters, conv)
        assert errs[0] > errs[1] > errs[2] > errs[3] and errs[3] < 1e-4, (name, errs)
        results[name] = (iters, np.stack([bso.se3_to_np(p) for p in poses]))
    assert results["fixed"][0] == results["exact"][0]
    assert np.abs(results["fixed"][1] - results["exact"][1]).max() < 1e-6


def test_work_order_caches_survive_interleaved_buffers_and_sizes(multi, oracle):
    """One context, two surfel buffers of different sizes, calls with < 4 keyframes (granule order) interleaved with calls with
    >= 4 keyframes (per-surfel order), and one buffer going S1 -> S2 -> S1: every call must give what a fresh context gives.
    (The two caches of make_schedule used to share scratch: building the per-surfel order of one buffer overwrote the cached
    granule order of another while leaving its key valid.)"""
    from tests import gpu_util
    sceneA, _ = multi                                                                    # 6 keyframes, 115 200 surfels
    sceneB = scenes.synthetic_scene(4, seed=11, use_depth_residuals=True, use_descriptor_residuals=True)
    assert sceneA.surfels_size >= 64 * 256 and sceneB.surfels_size >= 64 * 256 and sceneA.surfels_size != sceneB.surfels_size
    fresh = lambda scene: gpu_util.Hip(scene.to_device())
    refA1 = fresh(sceneA).accumulate_pose(0)
    refA_b = fresh(sceneA).accumulate_pose_batched()
    refB_b = fresh(sceneB).accumulate_pose_batched()
    refB1 = fresh(sceneB).accumulate_pose(1)
    ctx = badslam_amd.Context(0)
    A, B = gpu_util.Hip(sceneA.to_device(), ctx), gpu_util.Hip(sceneB.to_device(), ctx)

    def same(got, ref):
        if isinstance(ref, dict):
            return got["count"] == ref["count"] and np.array_equal(got["H"].view(np.uint32), ref["H"].view(np.uint32)) and np.array_equal(got["b"].view(np.uint32), ref["b"].view(np.uint32))
        return np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(got[1], ref[1])

    assert same(A.accumulate_pose(0), refA1)          # granule order of A cached
    assert same(B.accumulate_pose_batched(), refB_b)  # per-surfel order of B (other granule count) built
    assert same(A.accumulate_pose(0), refA1)          # A's cached granule order is still A's
    assert same(A.accumulate_pose_batched(), refA_b)
    assert same(B.accumulate_pose(1), refB1)
    assert same(A.accumulate_pose(0), refA1)
    assert same(B.accumulate_pose_batched(), refB_b)
    # one buffer, size S1 -> S2 -> S1
    full = A.d.surfels_size
    refA_half = None
    for size in (full, (full // 2) & ~255, full):
        A.d.surfels_size = size
        got1, gotb = A.accumulate_pose(0), A.accumulate_pose_batched()
        if size == full:
            assert same(got1, refA1) and same(gotb, refA_b)
        else:
            h = fresh(sceneA)
            h.d.surfels_size = size
            assert same(got1, h.accumulate_pose(0)) and same(gotb, h.accumulate_pose_batched())
    A.d.surfels_size = full

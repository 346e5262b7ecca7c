"""-m gpu: pairwise frame tracking (odometry) against the oracle's restatement of TrackFramePairwise
(BS/pairwise_frame_tracking.cc:256-678) and its kernels.  The reference has no unit test of the tracker; the known
answer here is the relative pose the two synthetic views were rendered with."""
import ctypes as C

import numpy as np
import pytest

import badslam_amd
from badslam_amd import abi
from tests import bso, scenes

pytestmark = pytest.mark.gpu
P = C.POINTER


def two_views(use_desc, seed=13, width=320, height=240):
    cam = bso.make_camera(262.5 * width / 320, 262.5 * width / 320, width / 2.0, height / 2.0, width, height)
    scene = scenes.synthetic_scene(2, seed=seed, width=width, height=height, cell=4, camera=cam, use_depth_residuals=True, use_descriptor_residuals=use_desc,
                                   translation_range=0.03, rotation_range=0.02)
    base, tracked = scene.keyframes
    truth = bso.se3_mul(bso.se3_inverse(base.global_T_frame), tracked.global_T_frame)
    return scene, base, tracked, truth


def make_ba(scene):
    from badslam_amd.direct_ba import DirectBA
    ba = DirectBA(1000, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, 0.8, 1, 1, 1, scene.color_camera, scene.depth_camera, 0,
                  scene.use_depth_residuals, scene.use_descriptor_residuals)
    ba.set_options(texture_mode=scene.tex_mode)
    for kf in scene.keyframes:
        ba.AddKeyframe(kf.id, max(kf.min_depth, 1e-3), max(kf.max_depth, 1e-2), kf.depth, kf.normals, kf.radius, kf.color, kf.global_T_frame)
    return ba


@pytest.mark.parametrize("use_desc", [False, True])
def test_track_frame_pairwise_recovers_the_rendered_motion(oracle, use_desc):
    scene, base, tracked, truth = two_views(use_desc)
    ba = make_ba(scene)
    est, its = ba.TrackKeyframePair(1, 0, bso.se3_identity(), num_scales=4)
    err = np.abs(bso.se3_log(bso.se3_mul(bso.se3_inverse(est), truth))).max()
    start = np.abs(bso.se3_log(truth)).max()
    assert start > 0.01 and err < (1e-4 if use_desc else 2e-5), (start, err, its)
    # same loop on the oracle: same number of Gauss-Newton iterations per scale (to within one) and the same pose
    ref, ref_its = scene.track_frame_pairwise(tracked, base, bso.se3_identity(), num_scales=4)
    assert np.abs(bso.se3_to_np(est) - bso.se3_to_np(ref)).max() < 2e-5, (bso.se3_to_np(est), bso.se3_to_np(ref))
    assert all(abs(a - b) <= 1 for a, b in zip(its, ref_its)), (its, ref_its)


def test_initial_estimate_selection(oracle):
    """test_different_initial_estimates: the candidate with (many) more / cheaper residuals at the coarsest scale wins
    (BS/pairwise_frame_tracking.cc:428-489); a wildly wrong first candidate must not hurt."""
    scene, base, tracked, truth = two_views(False)
    ba = make_ba(scene)
    bad = bso.se3_exp(np.array([0.6, -0.5, 0.4, 0.3, -0.3, 0.2], np.float32))
    est, _ = ba.TrackKeyframePair(1, 0, bad, bso.se3_identity(), num_scales=4, test_different_initial_estimates=True)
    assert np.abs(bso.se3_log(bso.se3_mul(bso.se3_inverse(est), truth))).max() < 2e-5


def dev_img(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def buf(t):
    return abi.Buffer2D(t.data_ptr(), t.shape[0], t.shape[1], t.stride(0) * t.element_size())


@pytest.mark.parametrize("use_desc", [False, True])
def test_pyramids_and_image_pair_kernels_match_oracle(oracle, use_desc):
    import torch
    scene, base, tracked, truth = two_views(use_desc, seed=17)
    O, L = bso.lib(), badslam_amd.lib()
    ctx = badslam_amd.Context(0)
    ctx.set_texture_mode(scene.tex_mode)
    num_scales = 3
    dp = scene.depth_params()
    tv, bv = tracked.view(), base.view()
    pyr = (abi.Buffer2D * (6 * num_scales))()
    O.bso_build_tracking_pyramids(num_scales, C.byref(scene.color_camera), C.byref(scene.depth_camera), C.byref(dp), C.byref(tv.depth), C.byref(tv.normals),
                                  C.byref(tv.color), C.byref(bv.depth), C.byref(bv.normals), C.byref(bv.color), scene.tex_mode, pyr)

    def as_np(b, dtype):
        n = b.height * b.pitch // np.dtype(dtype).itemsize
        return np.ctypeslib.as_array(C.cast(b.address, P(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,)).reshape(b.height, -1)[:, :b.width].copy()

    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # --- the device pyramid, level by level, from the same keyframe images
    cf = dev_img(torch, scene.cfactor)
    dp_dev = scene.depth_params(buf(cf))
    t_d16, b_d16 = dev_img(torch, tracked.depth.view(np.int16)), dev_img(torch, base.depth.view(np.int16))
    t_col, b_col = dev_img(torch, tracked.color), dev_img(torch, base.color)
    h, w = tracked.depth.shape
    col_buf = lambda t: abi.Buffer2D(t.data_ptr(), h, w, w * 4)
    gm_b = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    gm_t = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    x, y = col_buf(b_col), buf(gm_b)
    badslam_amd.check(L.bslam_compute_brightness_from_color(ctx.handle, stream, C.byref(x), C.byref(y)))
    x, y = col_buf(t_col), buf(gm_t)
    badslam_amd.check(L.bslam_compute_brightness_from_color(ctx.handle, stream, C.byref(x), C.byref(y)))
    levels = []
    bd = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    bc = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    a1, a2, a3, a4 = buf(b_d16), buf(gm_b), buf(bd), buf(bc)
    badslam_amd.check(L.bslam_calibrate_depth_and_transform_color_to_depth(ctx.handle, stream, C.byref(scene.color_camera), C.byref(scene.depth_camera),
                                                                          C.byref(dp_dev), C.byref(a1), C.byref(a2), C.byref(a3), C.byref(a4)))
    td = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    tc = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    a1, a3 = buf(t_d16), buf(td)
    badslam_amd.check(L.bslam_calibrate_depth(ctx.handle, stream, C.byref(dp_dev), C.byref(a1), C.byref(a3)))
    a1, a3 = buf(gm_t), buf(tc)
    badslam_amd.check(L.bslam_set_to_read_mode_normalized(ctx.handle, stream, C.byref(a1), C.byref(a3)))
    levels.append([bd, dev_img(torch, base.normals.view(np.int16)), bc, td, dev_img(torch, tracked.normals.view(np.int16)), tc])
    for s in range(1, num_scales):
        sh, sw = int(h / 2 ** s), int(w / 2 ** s)
        cur = []
        for side in range(2):
            d = torch.zeros((sh, sw), dtype=torch.float32, device="cuda")
            n = torch.zeros((sh, sw), dtype=torch.int16, device="cuda")
            c = torch.zeros((sh, sw), dtype=torch.uint8, device="cuda")
            pd, pn, pc = levels[s - 1][3 * side:3 * side + 3]
            i1, i2, i3, o1, o2, o3 = buf(pd), buf(pn), buf(pc), buf(d), buf(n), buf(c)
            badslam_amd.check(L.bslam_downsample_images(ctx.handle, stream, C.byref(i1), C.byref(i2), C.byref(i3), C.byref(o1), C.byref(o2), C.byref(o3)))
            cur += [d, n, c]
        levels.append(cur)
    torch.cuda.synchronize()
    for s in range(num_scales):
        for i, dtype in enumerate([np.float32, np.uint16, np.uint8, np.float32, np.uint16, np.uint8]):
            ref = as_np(pyr[6 * s + i], dtype)
            got = levels[s][i].cpu().numpy().view(dtype)
            if dtype == np.uint16:
                valid = as_np(pyr[6 * s + 3 * (i // 3)], np.float32) > 0     # normals are only defined where the level has depth
                assert np.array_equal(got[valid], ref[valid]), (s, i)
            else:
                assert np.array_equal(got.view(np.uint32 if dtype == np.float32 else dtype), ref.view(np.uint32 if dtype == np.float32 else dtype)), (s, i)

    # --- the two image-pair kernels at every scale, at a pose near the truth
    pose = bso.se3_mul(truth, bso.se3_exp(np.array([0.002, -0.001, 0.0015, 0.0005, -0.0004, 0.0006], np.float32)))
    M = bso.se3_matrix3x4(bso.se3_inverse(pose))
    for s in range(num_scales):
        f = 1.0 / 2 ** s
        sc = lambda c: bso.make_camera(c.fx * f, c.fy * f, c.cx * f, c.cy * f, int(f * c.width + 0.5), int(f * c.height + 0.5))
        tcc, tdc = sc(scene.color_camera), sc(scene.depth_camera)
        Pp = pyr[6 * s:6 * s + 6]
        H64, b64 = np.zeros(21), np.zeros(6)
        vis = C.c_uint32()
        O.bso_accumulate_pose_coeffs_from_images(1, int(use_desc), C.byref(tcc), C.byref(tdc), scene.baseline_fx, float(2 ** s), C.byref(Pp[3]), C.byref(Pp[4]),
                                                 C.byref(Pp[5]), C.byref(M), C.byref(Pp[0]), C.byref(Pp[1]), C.byref(Pp[2]), scene.tex_mode,
                                                 H64.ctypes.data_as(P(C.c_double)), b64.ctypes.data_as(P(C.c_double)), C.byref(vis))
        cnt_ref, cost_ref = C.c_uint32(), C.c_double()
        O.bso_compute_cost_and_residual_count_from_images(1, int(use_desc), C.byref(tcc), C.byref(tdc), scene.baseline_fx, float(2 ** s), C.byref(Pp[3]),
                                                          C.byref(Pp[4]), C.byref(Pp[5]), C.byref(M), C.byref(Pp[0]), C.byref(Pp[1]), C.byref(Pp[2]),
                                                          scene.tex_mode, C.byref(cnt_ref), C.byref(cost_ref))
        D = [buf(t) for t in levels[s]]
        H, b = np.zeros(21, np.float32), np.zeros(6, np.float32)
        v = C.c_uint32()
        badslam_amd.check(L.bslam_accumulate_pose_coeffs_from_images(ctx.handle, stream, 1, int(use_desc), C.byref(tcc), C.byref(tdc), scene.baseline_fx,
                                                                    float(2 ** s), C.byref(D[3]), C.byref(D[4]), C.byref(D[5]), C.byref(M), C.byref(D[0]),
                                                                    C.byref(D[1]), C.byref(D[2]), C.byref(v), H.ctypes.data_as(P(C.c_float)),
                                                                    b.ctypes.data_as(P(C.c_float))))
        cnt, cost = C.c_uint32(), C.c_float()
        badslam_amd.check(L.bslam_compute_cost_and_residual_count_from_images(ctx.handle, stream, 1, int(use_desc), C.byref(tcc), C.byref(tdc),
                                                                             scene.baseline_fx, float(2 ** s), C.byref(D[3]), C.byref(D[4]), C.byref(D[5]),
                                                                             C.byref(M), C.byref(D[0]), C.byref(D[1]), C.byref(D[2]), C.byref(cnt), C.byref(cost)))
        assert v.value == vis.value and cnt.value == cnt_ref.value and vis.value > 500, (s, v.value, vis.value)     # integer outputs: exact
        assert np.abs(H - H64).max() <= 1e-4 * np.abs(H64).max() and np.abs(b - b64).max() <= 1e-4 * np.abs(b64).max(), s
        assert abs(cost.value - cost_ref.value) <= 1e-4 * abs(cost_ref.value), s
    O.bso_free_tracking_pyramids(num_scales, pyr)


def test_sobel_and_calibrate_downsample_bit_exact(oracle):
    """ComputeSobelGradientMagnitudeCUDA (BS/kernel_pairwise_frame_tracking... via kernels_odometry: 3x3 Sobel of the luma, u8) and
    CalibrateAndDownsampleImagesCUDA (raw u16 depth -> calibrated, median-selected half-resolution depth + normals + colour) are
    integer / selection work: every output byte equals the oracle's."""
    import torch
    scene, base, tracked, truth = two_views(True, seed=23)
    O, L = bso.lib(), badslam_amd.lib()
    ctx = badslam_amd.Context(0)
    ctx.set_texture_mode(scene.tex_mode)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    h, w = tracked.depth.shape
    # --- Sobel
    ref_gm = np.zeros((h, w), np.uint8)
    tv = tracked.view()
    O.bso_compute_sobel_gradient_magnitude(C.byref(tv.color), C.byref(bso.np_buffer2d(ref_gm)))
    t_col = dev_img(torch, tracked.color)
    gm = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    x, y = abi.Buffer2D(t_col.data_ptr(), h, w, w * 4), buf(gm)
    badslam_amd.check(L.bslam_compute_sobel_gradient_magnitude(ctx.handle, stream, C.byref(x), C.byref(y)))
    torch.cuda.synchronize()
    assert np.array_equal(gm.cpu().numpy(), ref_gm) and ref_gm.max() >= 8 and (ref_gm > 0).mean() > 0.2
    # --- calibrate + downsample, with and without halving the colour image
    dp = scene.depth_params()
    cf = dev_img(torch, scene.cfactor)
    dp_dev = scene.depth_params(buf(cf))
    t_d16 = dev_img(torch, tracked.depth.view(np.int16))
    t_n = dev_img(torch, tracked.normals.view(np.int16))
    for downsample_color in (1, 0):
        ch, cw = (h // 2, w // 2) if downsample_color else (h, w)
        src_c = ref_gm if downsample_color else np.ascontiguousarray(ref_gm[:h // 2, :w // 2])
        rd, rn, rc = np.zeros((h // 2, w // 2), np.float32), np.zeros((h // 2, w // 2), np.uint16), np.zeros((h // 2, w // 2), np.uint8)
        O.bso_calibrate_and_downsample_images(downsample_color, C.byref(dp), C.byref(tv.depth), C.byref(tv.normals), C.byref(bso.np_buffer2d(src_c)),
                                              scene.tex_mode, C.byref(bso.np_buffer2d(rd)), C.byref(bso.np_buffer2d(rn)), C.byref(bso.np_buffer2d(rc)))
        d = torch.zeros((h // 2, w // 2), dtype=torch.float32, device="cuda")
        n = torch.zeros((h // 2, w // 2), dtype=torch.int16, device="cuda")
        c = torch.zeros((h // 2, w // 2), dtype=torch.uint8, device="cuda")
        sc = dev_img(torch, src_c)
        i1, i2, i3, o1, o2, o3 = buf(t_d16), buf(t_n), buf(sc), buf(d), buf(n), buf(c)
        badslam_amd.check(L.bslam_calibrate_and_downsample_images(ctx.handle, stream, downsample_color, C.byref(dp_dev), C.byref(i1), C.byref(i2), C.byref(i3),
                                                                 C.byref(o1), C.byref(o2), C.byref(o3)))
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy().view(np.uint32), rd.view(np.uint32)), downsample_color
        valid = rd > 0
        assert valid.mean() > 0.5
        assert np.array_equal(n.cpu().numpy().view(np.uint16)[valid], rn[valid]), downsample_color
        assert np.array_equal(c.cpu().numpy(), rc), downsample_color


def test_gradmag_image_pair_kernels_match_oracle(oracle):
    """The use_gradmag forms of the two image-pair kernels (one colour residual on the gradient-magnitude images,
    BS/kernel_opt_pose.cu:713-937 and 1173-1338) on the oracle's own pyramids, every scale."""
    import torch
    scene, base, tracked, truth = two_views(True, seed=17)
    O, L = bso.lib(), badslam_amd.lib()
    ctx = badslam_amd.Context(0)
    ctx.set_texture_mode(scene.tex_mode)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    num_scales = 3
    dp = scene.depth_params()
    tv, bv = tracked.view(), base.view()
    pyr = (abi.Buffer2D * (6 * num_scales))()
    O.bso_set_tracking_variant(1, 1)
    try:
        O.bso_build_tracking_pyramids(num_scales, C.byref(scene.color_camera), C.byref(scene.depth_camera), C.byref(dp), C.byref(tv.depth), C.byref(tv.normals),
                                      C.byref(tv.color), C.byref(bv.depth), C.byref(bv.normals), C.byref(bv.color), scene.tex_mode, pyr)

        def as_np(b, dtype):
            n = b.height * b.pitch // np.dtype(dtype).itemsize
            return np.ctypeslib.as_array(C.cast(b.address, P(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,)).reshape(b.height, -1)[:, :b.width].copy()

        pose = bso.se3_mul(truth, bso.se3_exp(np.array([0.002, -0.001, 0.0015, 0.0005, -0.0004, 0.0006], np.float32)))
        M = bso.se3_matrix3x4(bso.se3_inverse(pose))
        for s in range(num_scales):
            f = 1.0 / 2 ** s
            sc = lambda c: bso.make_camera(c.fx * f, c.fy * f, c.cx * f, c.cy * f, int(f * c.width + 0.5), int(f * c.height + 0.5))
            tcc, tdc = sc(scene.color_camera), sc(scene.depth_camera)
            Pp = pyr[6 * s:6 * s + 6]
            H64, b64 = np.zeros(21), np.zeros(6)
            vis = C.c_uint32()
            O.bso_accumulate_pose_coeffs_from_images(1, 1, C.byref(tcc), C.byref(tdc), scene.baseline_fx, float(2 ** s), C.byref(Pp[3]), C.byref(Pp[4]),
                                                     C.byref(Pp[5]), C.byref(M), C.byref(Pp[0]), C.byref(Pp[1]), C.byref(Pp[2]), scene.tex_mode,
                                                     H64.ctypes.data_as(P(C.c_double)), b64.ctypes.data_as(P(C.c_double)), C.byref(vis))
            cnt_ref, cost_ref = C.c_uint32(), C.c_double()
            O.bso_compute_cost_and_residual_count_from_images(1, 1, C.byref(tcc), C.byref(tdc), scene.baseline_fx, float(2 ** s), C.byref(Pp[3]),
                                                              C.byref(Pp[4]), C.byref(Pp[5]), C.byref(M), C.byref(Pp[0]), C.byref(Pp[1]), C.byref(Pp[2]),
                                                              scene.tex_mode, C.byref(cnt_ref), C.byref(cost_ref))
            dtypes = [np.float32, np.uint16, np.uint8, np.float32, np.uint16, np.uint8]
            T = [dev_img(torch, as_np(Pp[i], dt).view(np.int16 if dt == np.uint16 else dt)) for i, dt in enumerate(dtypes)]
            D = [buf(t) for t in T]
            H, b = np.zeros(21, np.float32), np.zeros(6, np.float32)
            v = C.c_uint32()
            badslam_amd.check(L.bslam_accumulate_pose_coeffs_from_images_gradmag(
                ctx.handle, stream, 1, 1, C.byref(tcc), C.byref(tdc), scene.baseline_fx, float(2 ** s), C.byref(D[3]), C.byref(D[4]), C.byref(D[5]), C.byref(M),
                C.byref(D[0]), C.byref(D[1]), C.byref(D[2]), C.byref(v), H.ctypes.data_as(P(C.c_float)), b.ctypes.data_as(P(C.c_float))))
            cnt, cost = C.c_uint32(), C.c_float()
            badslam_amd.check(L.bslam_compute_cost_and_residual_count_from_images_gradmag(
                ctx.handle, stream, 1, 1, C.byref(tcc), C.byref(tdc), scene.baseline_fx, float(2 ** s), C.byref(D[3]), C.byref(D[4]), C.byref(D[5]), C.byref(M),
                C.byref(D[0]), C.byref(D[1]), C.byref(D[2]), C.byref(cnt), C.byref(cost)))
            assert v.value == vis.value and cnt.value == cnt_ref.value and vis.value > 500, (s, v.value, vis.value, cnt.value, cnt_ref.value)
            assert np.abs(H - H64).max() <= 1e-4 * np.abs(H64).max() and np.abs(b - b64).max() <= 1e-4 * np.abs(b64).max(), s
            assert abs(cost.value - cost_ref.value) <= 1e-4 * abs(cost_ref.value), s
            # and it is a different problem from the two-residual descriptor form
            H2, b2 = np.zeros(21, np.float32), np.zeros(6, np.float32)
            badslam_amd.check(L.bslam_accumulate_pose_coeffs_from_images(
                ctx.handle, stream, 1, 1, C.byref(tcc), C.byref(tdc), scene.baseline_fx, float(2 ** s), C.byref(D[3]), C.byref(D[4]), C.byref(D[5]), C.byref(M),
                C.byref(D[0]), C.byref(D[1]), C.byref(D[2]), C.byref(v), H2.ctypes.data_as(P(C.c_float)), b2.ctypes.data_as(P(C.c_float))))
            assert np.abs(H2 - H).max() > 1e-3 * np.abs(H).max()
        O.bso_free_tracking_pyramids(num_scales, pyr)
    finally:
        O.bso_set_tracking_variant(0, 1)


@pytest.mark.parametrize("use_gradmag,use_pyramid_level_0", [(True, True), (False, False), (True, False)])
def test_tracker_variants_match_oracle(oracle, use_gradmag, use_pyramid_level_0):
    """TrackFramePairwise's two switches (BS/pairwise_frame_tracking.cc:164-166, 283-341, 367) against the oracle's loop."""
    scene, base, tracked, truth = two_views(True)
    ba = make_ba(scene)
    est, its = ba.TrackKeyframePair(1, 0, bso.se3_identity(), num_scales=4, use_pyramid_level_0=use_pyramid_level_0, use_gradmag=use_gradmag)
    err = np.abs(bso.se3_log(bso.se3_mul(bso.se3_inverse(est), truth))).max()
    assert err < 3e-4, (err, its)
    bso.lib().bso_set_tracking_variant(int(use_gradmag), int(use_pyramid_level_0))
    try:
        ref, ref_its = scene.track_frame_pairwise(tracked, base, bso.se3_identity(), num_scales=4)
    finally:
        bso.lib().bso_set_tracking_variant(0, 1)
    assert np.abs(bso.se3_to_np(est) - bso.se3_to_np(ref)).max() < 2e-5, (bso.se3_to_np(est), bso.se3_to_np(ref))
    assert all(abs(a - b) <= 1 for a, b in zip(its, ref_its)), (its, ref_its)
    if not use_pyramid_level_0:
        assert its[0] == 0 and its[1] > 0

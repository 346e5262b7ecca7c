"""-m gpu parity of the keyframe preprocessing producers against the oracle.  Every output is an
integer image (u8 luma, u16 depth, 2 x s8 normals, half-precision radius bits): bit-exact."""
import ctypes as C

import numpy as np
import pytest

import badslam_amd
from badslam_amd import abi
from tests import bso, scenes

pytestmark = pytest.mark.gpu
P = C.POINTER


def raw_depth_image(seed=0, width=640, height=480):
    """A raw sensor-like depth image: the synthetic plane scene + noise, holes (0) and far values."""
    rng = np.random.default_rng(seed)
    scene = scenes.synthetic_scene(1, seed=seed, width=width, height=height, cell=4)
    d = scene.keyframes[0].depth.astype(np.int64)
    raw = np.where(d >= 32768, 0, d + rng.integers(-30, 31, d.shape)).clip(0, 65535).astype(np.uint16)
    raw[rng.random(d.shape) < 0.03] = 0
    raw[100:140, 200:260] = 0
    raw[300:310, 50:400] = 40000          # beyond the cutoff
    return scene, raw


def dev(t, a):
    return t.from_numpy(np.ascontiguousarray(a)).cuda()


def buf(tensor, elems_w=None):
    w = tensor.shape[1] if elems_w is None else elems_w
    return abi.Buffer2D(tensor.data_ptr(), tensor.shape[0], w, tensor.stride(0) * tensor.element_size())


def stream_ptr(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_preprocessing_pipeline_matches_oracle(oracle):
    import torch
    L, O = badslam_amd.lib(), bso.lib()
    ctx = badslam_amd.Context(0)
    scene, raw = raw_depth_image()
    h, w = raw.shape
    cam, dp_host = scene.depth_camera, scene.depth_params()
    rng = np.random.default_rng(1)
    scene.cfactor[:] = rng.uniform(-0.003, 0.003, scene.cfactor.shape).astype(np.float32)   # exercise the depth deformation
    scene.a = 0.02
    dp_host = scene.depth_params()

    # --- brightness
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    col_ref = np.zeros((h, w, 4), np.uint8)
    cb = bso.np_buffer2d(col_ref)
    O.bso_compute_brightness(w, h, rgb.ctypes.data, C.byref(cb))
    rgb_d = dev(torch, rgb.reshape(h, w * 3))
    col_d = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    rb = abi.Buffer2D(rgb_d.data_ptr(), h, w, w * 3)
    ob = abi.Buffer2D(col_d.data_ptr(), h, w, w * 4)
    badslam_amd.check(L.bslam_compute_brightness(ctx.handle, stream_ptr(torch), C.byref(rb), C.byref(ob)))
    assert np.array_equal(col_d.cpu().numpy(), col_ref)

    # --- bilateral filter + cutoff (BS/bad_slam.cc:694-702 defaults: sigma_xy 3, sigma_inv_depth 0.005, radius factor 2, max depth 3 m)
    raw_to_float = scene.raw_to_float_depth
    max_depth = np.uint16(3.0 / raw_to_float)
    filt_ref = np.zeros((h, w), np.uint16)
    ib, fb = bso.np_buffer2d(raw), bso.np_buffer2d(filt_ref)
    O.bso_bilateral_filter_and_depth_cutoff(3.0, 0.005, 2.0, int(max_depth), raw_to_float, C.byref(ib), C.byref(fb))
    raw_d = dev(torch, raw.view(np.int16))
    filt_d = torch.zeros((h, w), dtype=torch.int16, device="cuda")
    a_, b_ = buf(raw_d), buf(filt_d)
    badslam_amd.check(L.bslam_bilateral_filter_and_depth_cutoff(ctx.handle, stream_ptr(torch), 3.0, 0.005, 2.0, int(max_depth), raw_to_float,
                                                               C.byref(a_), C.byref(b_)))
    filt = filt_d.cpu().numpy().view(np.uint16)
    assert np.array_equal(filt, filt_ref), int((filt != filt_ref).sum())
    assert ((filt_ref < 32768).sum() > 0.5 * h * w) and (filt_ref[300:310, 50:400] == 65535).all()

    # --- normals
    nd_ref, nn_ref = np.zeros((h, w), np.uint16), np.zeros((h, w), np.uint16)
    fb2, ndb, nnb = bso.np_buffer2d(filt_ref), bso.np_buffer2d(nd_ref), bso.np_buffer2d(nn_ref)
    O.bso_compute_normals(C.byref(cam), C.byref(dp_host), C.byref(fb2), C.byref(ndb), C.byref(nnb))
    cf_d = dev(torch, scene.cfactor)
    dp_dev = scene.depth_params(buf(cf_d))
    nd_d = torch.zeros((h, w), dtype=torch.int16, device="cuda")
    nn_d = torch.zeros((h, w), dtype=torch.int16, device="cuda")
    x_, y_, z_ = buf(filt_d), buf(nd_d), buf(nn_d)
    badslam_amd.check(L.bslam_compute_normals(ctx.handle, stream_ptr(torch), C.byref(cam), C.byref(dp_dev), C.byref(x_), C.byref(y_), C.byref(z_)))
    assert np.array_equal(nd_d.cpu().numpy().view(np.uint16), nd_ref)
    assert np.array_equal(nn_d.cpu().numpy().view(np.uint16), nn_ref)

    # --- radii + isolated pixel removal
    rad_ref, out_ref = np.zeros((h, w), np.uint16), np.zeros((h, w), np.uint16)
    ndb2, rb2, ob2 = bso.np_buffer2d(nd_ref), bso.np_buffer2d(rad_ref), bso.np_buffer2d(out_ref)
    O.bso_compute_point_radii_and_remove_isolated_pixels(C.byref(cam), raw_to_float, C.byref(ndb2), C.byref(rb2), C.byref(ob2))
    rad_d = torch.zeros((h, w), dtype=torch.int16, device="cuda")
    out_d = torch.zeros((h, w), dtype=torch.int16, device="cuda")
    p_, q_, r_ = buf(nd_d), buf(rad_d), buf(out_d)
    badslam_amd.check(L.bslam_compute_point_radii_and_remove_isolated_pixels(ctx.handle, stream_ptr(torch), C.byref(cam), raw_to_float,
                                                                            C.byref(p_), C.byref(q_), C.byref(r_)))
    assert np.array_equal(out_d.cpu().numpy().view(np.uint16), out_ref)
    assert np.array_equal(rad_d.cpu().numpy().view(np.uint16), rad_ref)
    assert 0 < (out_ref < 32768).sum() < (nd_ref < 32768).sum()          # isolated pixels were removed

    # --- min / max depth
    mn_ref, mx_ref = C.c_float(), C.c_float()
    O.bso_compute_min_max_depth(C.byref(ndb2), raw_to_float, C.byref(mn_ref), C.byref(mx_ref))
    mn, mx = C.c_float(), C.c_float()
    badslam_amd.check(L.bslam_compute_min_max_depth(ctx.handle, stream_ptr(torch), C.byref(p_), raw_to_float, C.byref(mn), C.byref(mx)))
    assert mn.value == mn_ref.value and mx.value == mx_ref.value and 0 < mn.value < mx.value


def test_min_max_of_an_all_invalid_image(oracle):
    import torch
    L = badslam_amd.lib()
    ctx = badslam_amd.Context(0)
    d = torch.full((48, 64), -1, dtype=torch.int16, device="cuda")   # 65535 everywhere
    b = buf(d)
    mn, mx = C.c_float(), C.c_float()
    badslam_amd.check(L.bslam_compute_min_max_depth(ctx.handle, stream_ptr(torch), C.byref(b), 0.0002, C.byref(mn), C.byref(mx)))
    assert np.isinf(mn.value) and mx.value == 0.0   # the init values (BS/cuda_depth_processing.cu:374-388)


def test_add_keyframe_from_raw_images_through_the_host_class(oracle):
    """new Keyframe(stream, frame_index, depth_params, depth_camera, depth_image, color_image, pose) + AddKeyframe
    (BS/keyframe.cc:82-161) on the device = the oracle's restatement of that constructor, bit for bit; and the known-answer
    pose test still holds when the keyframe is produced this way."""
    from badslam_amd.direct_ba import DirectBA
    scene, ref, depth, rgb = scenes.pose_geometric_scene(seed=0, return_raw=True)   # the oracle preprocessed `ref` from (depth, rgb)
    h, w = depth.shape
    ba = DirectBA(scene.max_surfels, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, 0.8, 1, 1, 1,
                  scene.color_camera, scene.depth_camera, 0, True, False)
    kid = ba.AddKeyframeFromImages(0, depth, rgb, bso.se3_identity())
    d, n, rad, col, mn, mx = ba.keyframe_images(kid, h, w)
    assert np.array_equal(d, ref.depth) and np.array_equal(n, ref.normals) and np.array_equal(rad, ref.radius) and np.array_equal(col, ref.color)
    assert mn == np.float32(ref.min_depth) and mx == np.float32(ref.max_depth)
    # surfels from this keyframe, then the known-answer pose estimation on it
    ba.CreateSurfelsForKeyframe(False, kid)
    assert ba.surfels_size() == scene.surfels_size == 163704
    gt = bso.se3_identity()
    worst = 0.0
    for off in scenes.offsets_13(0.005, 0.001):
        est = ba.EstimateFramePose(kid, bso.se3_mul(off, bso.se3_inverse(gt)))
        worst = max(worst, float(np.abs(bso.se3_log(bso.se3_mul(bso.se3_inverse(est), gt))).max()))
    assert worst < 1.1e-6, worst

"""Test-side binding of the ORACLE (oracle/libbslam_oracle.so) and host scene containers.

Test infrastructure only: the product package (badslam_amd) never imports this.
Host buffers are numpy arrays; `HostScene.to_device()` uploads them into torch CUDA
tensors and returns the same POD views pointing at device memory, so the HIP path
and the oracle see identical bytes.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from badslam_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libbslam_oracle.so")

P = C.POINTER
_CAM, _DP, _BUF, _KFS = P(abi.Camera4f), P(abi.DepthParams), P(abi.Buffer2D), P(abi.KeyframeView)


def build_oracle():
    """Compiles the oracle's C restatement with gcc (oracle/Makefile)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    L = C.CDLL(ORACLE_SO)
    f32p, u32p, f64p = P(C.c_float), P(C.c_uint32), P(C.c_double)
    sig = {
        "bso_se3_identity": (None, [P(abi.SE3f)]),
        "bso_se3_exp": (None, [f32p, P(abi.SE3f)]),
        "bso_se3_log": (None, [P(abi.SE3f), f32p]),
        "bso_se3_mul": (None, [P(abi.SE3f), P(abi.SE3f), P(abi.SE3f)]),
        "bso_se3_inverse": (None, [P(abi.SE3f), P(abi.SE3f)]),
        "bso_se3_matrix3x4": (None, [P(abi.SE3f), P(abi.Mat3x4)]),
        "bso_keyframe_set_pose": (None, [_KFS, P(abi.SE3f)]),
        "bso_is_scale1_pose_estimation_converged": (C.c_int, [f32p]),
        "bso_solve_ldlt_upper": (None, [C.c_int, f32p, f32p, f32p]),
        "bso_association": (None, [_CAM, _DP, _KFS, C.c_uint32, _BUF, u32p]),
        "bso_accumulate_pose_estimation_coeffs": (None, [
            C.c_int, C.c_int, _CAM, _CAM, _DP, _BUF, _BUF, _BUF, P(abi.Mat3x4), C.c_uint32, _BUF,
            C.c_int, u32p, f32p, f32p, f32p, f64p, f64p, f32p]),
        "bso_estimate_frame_pose": (None, [
            C.c_int, C.c_int, _CAM, _CAM, _DP, _BUF, _BUF, _BUF, P(abi.SE3f), C.c_uint32, _BUF,
            C.c_int, C.c_int, P(abi.SE3f), P(C.c_int), P(C.c_int)]),
        "bso_assign_colors": (None, [_CAM, _CAM, _DP, C.c_int, _KFS, C.c_int, C.c_uint32, _BUF]),
        "bso_update_surfel_activation": (None, [_CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, _BUF]),
        "bso_update_surfel_normals": (None, [_CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, _BUF]),
        "bso_optimize_geometry_iteration": (None, [
            C.c_int, C.c_int, _CAM, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, _BUF, C.c_int]),
        "bso_pcg_init": (None, [P(abi.PCGLayout), _CAM, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, P(abi.PCGVectors), C.c_int]),
        "bso_pcg_init2": (None, [P(abi.PCGLayout), C.c_float, P(abi.PCGVectors)]),
        "bso_pcg_step1": (None, [P(abi.PCGLayout), _CAM, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, P(abi.PCGVectors), C.c_int, C.c_int]),
        "bso_pcg_last_alpha_d64": (C.c_double, []),
        "bso_pcg_step2": (None, [P(abi.PCGLayout), P(abi.PCGVectors), f32p]),
        "bso_pcg_step3": (None, [P(abi.PCGLayout), P(abi.PCGVectors)]),
        "bso_update_surfels_from_pcg_delta": (None, [C.c_uint32, _BUF, C.c_int, C.c_uint32, f32p]),
        "bso_update_cfactors_from_pcg_delta": (None, [_BUF, C.c_uint32, f32p]),
        "bso_optimize_intrinsics": (None, [C.c_int, C.c_int, C.c_int, _KFS, _CAM, _CAM, _DP, C.c_uint32, _BUF, _CAM, _CAM, f32p, C.c_int]),
        "bso_compute_brightness": (None, [C.c_int, C.c_int, C.c_void_p, _BUF]),
        "bso_bilateral_filter_and_depth_cutoff": (None, [C.c_float, C.c_float, C.c_float, C.c_uint16, C.c_float, _BUF, _BUF]),
        "bso_compute_normals": (None, [_CAM, _DP, _BUF, _BUF, _BUF]),
        "bso_compute_point_radii_and_remove_isolated_pixels": (None, [_CAM, C.c_float, _BUF, _BUF, _BUF]),
        "bso_compute_min_max_depth": (None, [_BUF, C.c_float, f32p, f32p]),
        "bso_preprocess_depth": (None, [_CAM, _DP, _BUF, _BUF, _BUF, _BUF, f32p, f32p]),
        "bso_create_surfels_for_keyframe": (C.c_uint32, [_CAM, _CAM, _DP, _KFS, P(abi.SE3f), u32p, C.c_uint32, _BUF, C.c_int]),
        "bso_create_surfels_for_keyframe_ex": (C.c_uint32, [C.c_int, C.c_int, _CAM, _CAM, _DP, _KFS, P(abi.Mat3x4), C.c_int, _KFS, P(abi.Mat3x4),
                                                            u32p, C.c_uint32, _BUF, C.c_int]),
        "bso_determine_supporting_surfels": (None, [C.c_int, C.c_float, _CAM, _DP, _KFS, C.c_uint32, _BUF, u32p, u32p, u32p, u32p]),
        "bso_delete_surfels_and_update_radii": (None, [C.c_int, _CAM, _DP, C.c_int, _KFS, u32p, C.c_uint32, _BUF]),
        "bso_compact_surfels": (None, [C.c_uint32, u32p, _BUF, _BUF]),
        "bso_track_frame_pairwise": (None, [C.c_int, C.c_int, C.c_int, _CAM, _CAM, _DP, _BUF, _BUF, _BUF, _BUF, _BUF, _BUF, C.c_int, C.c_int,
                                            P(abi.SE3f), P(abi.SE3f), P(abi.SE3f), P(C.c_int)]),
        "bso_compute_sobel_gradient_magnitude": (None, [_BUF, _BUF]),
        "bso_calibrate_and_downsample_images": (None, [C.c_int, _DP, _BUF, _BUF, _BUF, C.c_int, _BUF, _BUF, _BUF]),
        "bso_set_tracking_variant": (None, [C.c_int, C.c_int]),
        "bso_build_tracking_pyramids": (None, [C.c_int, _CAM, _CAM, _DP, _BUF, _BUF, _BUF, _BUF, _BUF, _BUF, C.c_int, _BUF]),
        "bso_free_tracking_pyramids": (None, [C.c_int, _BUF]),
        "bso_accumulate_pose_coeffs_from_images": (None, [C.c_int, C.c_int, _CAM, _CAM, C.c_float, C.c_float, _BUF, _BUF, _BUF, P(abi.Mat3x4), _BUF, _BUF,
                                                          _BUF, C.c_int, f64p, f64p, u32p]),
        "bso_compute_cost_and_residual_count_from_images": (None, [C.c_int, C.c_int, _CAM, _CAM, C.c_float, C.c_float, _BUF, _BUF, _BUF, P(abi.Mat3x4),
                                                                   _BUF, _BUF, _BUF, C.c_int, u32p, f64p]),
        "bso_jacobian_probe": (C.c_int, [C.c_int, C.c_int, f32p, f32p]),
        "bso_set_intrinsics_sum64": (None, [C.c_int]),
        "bso_set_literal_mode": (None, [C.c_int]),
        "bso_get_literal_mode": (C.c_int, []),
        "bso_association_margins": (None, [_CAM, _DP, _KFS, C.c_uint32, _BUF, f64p]),
        "bso_bench_ba_iteration": (C.c_int, [C.c_int, C.c_int, _CAM, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, _BUF, C.c_int, C.c_int,
                                             P(C.c_uint64), f32p]),
        "bso_bench_pose_pass": (C.c_int, [C.c_int, C.c_int, _CAM, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, C.c_int, f32p, u32p, C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# ----------------------------------------------------------------------------- helpers

class literal_mode:
    """with bso.literal_mode(): the oracle evaluates the reference's expressions literally (oracle/bso_math.h)."""

    def __enter__(self):
        lib().bso_set_literal_mode(1)

    def __exit__(self, *exc):
        lib().bso_set_literal_mode(0)


def np_buffer2d(arr):
    """bslam_buffer2d over a C-contiguous 2-D (or [h, w, 4] uint8) numpy array."""
    assert arr.flags["C_CONTIGUOUS"]
    h, w = arr.shape[0], arr.shape[1]
    return abi.Buffer2D(arr.ctypes.data, h, w, arr.strides[0])


def fptr(arr):
    return arr.ctypes.data_as(P(C.c_float))


def make_camera(fx, fy, cx, cy, width, height):
    return abi.Camera4f(fx, fy, cx, cy, width, height)


def se3_identity():
    T = abi.SE3f()
    lib().bso_se3_identity(C.byref(T))
    return T


def se3_exp(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    T = abi.SE3f()
    lib().bso_se3_exp(fptr(x), C.byref(T))
    return T


def se3_log(T):
    out = np.zeros(6, np.float32)
    lib().bso_se3_log(C.byref(T), fptr(out))
    return out


def se3_mul(a, b):
    T = abi.SE3f()
    lib().bso_se3_mul(C.byref(a), C.byref(b), C.byref(T))
    return T


def se3_inverse(a):
    T = abi.SE3f()
    lib().bso_se3_inverse(C.byref(a), C.byref(T))
    return T


def se3_copy(a):
    T = abi.SE3f()
    C.memmove(C.byref(T), C.byref(a), C.sizeof(abi.SE3f))
    return T


def se3_matrix3x4(a):
    M = abi.Mat3x4()
    lib().bso_se3_matrix3x4(C.byref(a), C.byref(M))
    return M


def se3_to_np(a):
    return np.array(list(a.q) + list(a.t), np.float32)


class HostKeyframe:
    """Host mirror of vis::Keyframe's buffers (BS/keyframe.h:227-231)."""

    def __init__(self, depth, normals, radius, color, global_T_frame, kf_id=0, activation=abi.KF_ACTIVE,
                 min_depth=0.0, max_depth=0.0):
        self.depth, self.normals, self.radius, self.color = depth, normals, radius, color
        self.global_T_frame = se3_copy(global_T_frame)
        self.id = kf_id
        self.activation = activation
        self.min_depth, self.max_depth = min_depth, max_depth

    def view(self, bufs=None):
        """bslam_keyframe_view over host memory (or over `bufs`, 4 Buffer2D of device memory)."""
        v = abi.KeyframeView()
        if bufs is None:
            v.depth, v.normals = np_buffer2d(self.depth), np_buffer2d(self.normals)
            v.radius, v.color = np_buffer2d(self.radius), np_buffer2d(self.color)
        else:
            v.depth, v.normals, v.radius, v.color = bufs
        lib().bso_keyframe_set_pose(C.byref(v), C.byref(self.global_T_frame))
        v.activation = self.activation
        v.id = self.id
        return v


class HostScene:
    """Host mirror of the DirectBA scene state (BS/direct_ba.h: surfels_, active_surfels_,
    cfactor_buffer_, depth_params_, cameras, keyframes_)."""

    def __init__(self, color_camera, depth_camera, raw_to_float_depth, baseline_fx, cell, max_surfels,
                 use_depth_residuals=True, use_descriptor_residuals=False, tex_mode=abi.TEX_FIXED_POINT_1_8):
        self.color_camera, self.depth_camera = color_camera, depth_camera
        w, h = depth_camera.width, depth_camera.height
        self.cfactor = np.zeros(((h - 1) // cell + 1, (w - 1) // cell + 1), np.float32)
        self.a = 0.0
        self.raw_to_float_depth, self.baseline_fx, self.cell = raw_to_float_depth, baseline_fx, cell
        self.max_surfels = max_surfels
        self.surfels = np.zeros((abi.SURFEL_ATTRIBUTE_COUNT, max_surfels), np.float32)
        self.active = np.zeros((1, max_surfels), np.uint8)
        self.surfels_size = 0
        self.keyframes = []
        self.use_depth_residuals = use_depth_residuals
        self.use_descriptor_residuals = use_descriptor_residuals
        self.tex_mode = tex_mode

    # --- POD views over host memory
    def depth_params(self, cfactor_buf=None):
        dp = abi.DepthParams()
        dp.cfactor_buffer = np_buffer2d(self.cfactor) if cfactor_buf is None else cfactor_buf
        dp.a = self.a
        dp.raw_to_float_depth = self.raw_to_float_depth
        dp.baseline_fx = self.baseline_fx
        dp.sparse_surfel_cell_size = self.cell
        return dp

    def surfel_buf(self):
        return np_buffer2d(self.surfels)

    def active_buf(self):
        return np_buffer2d(self.active)

    def keyframe_views(self):
        arr = (abi.KeyframeView * max(1, len(self.keyframes)))()
        for i, kf in enumerate(self.keyframes):
            arr[i] = kf.view()
        return arr

    # --- scene construction through the oracle's restatement of the producers
    def add_keyframe_from_images(self, depth_u16, rgb_u8, global_T_frame):
        """Keyframe(stream, frame_index, depth_params, depth_camera, depth_image, color_image, pose)
        (BS/keyframe.cc:82-161) followed by DirectBA::AddKeyframe (BS/direct_ba.cc:196-204)."""
        L = lib()
        h, w = depth_u16.shape
        depth_in = np.ascontiguousarray(depth_u16, np.uint16)
        depth = np.zeros((h, w), np.uint16)
        normals = np.zeros((h, w), np.uint16)
        radius = np.zeros((h, w), np.uint16)
        rgb = np.ascontiguousarray(rgb_u8, np.uint8)
        ch, cw = rgb.shape[:2]                       # the colour camera may have its own resolution
        color = np.zeros((ch, cw, 4), np.uint8)
        cb = np_buffer2d(color)
        L.bso_compute_brightness(cw, ch, rgb.ctypes.data, C.byref(cb))
        dp = self.depth_params()
        mn, mx = C.c_float(), C.c_float()
        bi, bd, bn, br = np_buffer2d(depth_in), np_buffer2d(depth), np_buffer2d(normals), np_buffer2d(radius)
        L.bso_preprocess_depth(C.byref(self.depth_camera), C.byref(dp), C.byref(bi), C.byref(bd), C.byref(bn), C.byref(br),
                               C.byref(mn), C.byref(mx))
        kf = HostKeyframe(depth, normals, radius, color, global_T_frame, kf_id=len(self.keyframes),
                          min_depth=mn.value, max_depth=mx.value)
        self.keyframes.append(kf)
        return kf

    def create_surfels_for_keyframe(self, kf):
        """DirectBA::CreateSurfelsForKeyframe(filter_new_surfels=false) (BS/direct_ba.cc:340-405)."""
        L = lib()
        dp = self.depth_params()
        v = kf.view()
        size = C.c_uint32(self.surfels_size)
        sb = self.surfel_buf()
        n = L.bso_create_surfels_for_keyframe(C.byref(self.color_camera), C.byref(self.depth_camera), C.byref(dp), C.byref(v),
                                              C.byref(kf.global_T_frame), C.byref(size), self.max_surfels, C.byref(sb),
                                              self.tex_mode)
        self.active[0, self.surfels_size:size.value] = abi.BSLAM_SURFEL_ACTIVE_FLAG
        self.surfels_size = size.value
        return n

    # --- pairwise frame tracking (oracle)
    def track_frame_pairwise(self, tracked, base, init1, init2=None, num_scales=5, test_different_initial_estimates=False):
        """TrackFramePairwise (BS/pairwise_frame_tracking.cc:256-678): base_T_frame of `tracked` relative to `base`."""
        dp = self.depth_params()
        tv, bv = tracked.view(), base.view()
        out = abi.SE3f()
        its = (C.c_int * num_scales)()
        i2 = init1 if init2 is None else init2
        lib().bso_track_frame_pairwise(num_scales, int(self.use_depth_residuals), int(self.use_descriptor_residuals), C.byref(self.color_camera),
                                       C.byref(self.depth_camera), C.byref(dp), C.byref(tv.depth), C.byref(tv.normals), C.byref(tv.color),
                                       C.byref(bv.depth), C.byref(bv.normals), C.byref(bv.color), self.tex_mode, int(test_different_initial_estimates),
                                       C.byref(init1), C.byref(i2), C.byref(out), its)
        return out, list(its)

    # --- surfel lifecycle (oracle)
    def covis_args(self, kf, covis):
        """(count, KeyframeView[count], Mat3x4[count]) with covis_T_frame = covis frame_T_global * kf global_T_frame
        (BS/direct_ba.cc:365-370)."""
        n = len(covis)
        views = (abi.KeyframeView * max(1, n))()
        mats = (abi.Mat3x4 * max(1, n))()
        for i, ck in enumerate(covis):
            views[i] = ck.view()
            mats[i] = se3_matrix3x4(se3_mul(se3_inverse(ck.global_T_frame), kf.global_T_frame))
        return n, views, mats

    def create_surfels_for_keyframe_ex(self, kf, filter_new_surfels, min_observation_count, covis):
        """DirectBA::CreateSurfelsForKeyframe (BS/direct_ba.cc:340-405) with the observation-count filter."""
        dp, v, sb = self.depth_params(), kf.view(), self.surfel_buf()
        n, views, mats = self.covis_args(kf, covis)
        G = se3_matrix3x4(kf.global_T_frame)
        size = C.c_uint32(self.surfels_size)
        created = lib().bso_create_surfels_for_keyframe_ex(int(filter_new_surfels), min_observation_count, C.byref(self.color_camera),
                                                           C.byref(self.depth_camera), C.byref(dp), C.byref(v), C.byref(G), n, views, mats,
                                                           C.byref(size), self.max_surfels, C.byref(sb), self.tex_mode)
        self.active[0, self.surfels_size:size.value] = abi.BSLAM_SURFEL_ACTIVE_FLAG
        self.surfels_size = size.value
        return created

    def merge_surfels(self, kf, merge_dist_factor, surfel_count):
        dp, v, sb = self.depth_params(), kf.view(), self.surfel_buf()
        cells = self.cfactor.shape[0] * self.cfactor.shape[1]
        sup = [np.zeros(cells, np.uint32) for _ in range(3)]
        cnt = C.c_uint32(surfel_count)
        lib().bso_determine_supporting_surfels(1, merge_dist_factor, C.byref(self.depth_camera), C.byref(dp), C.byref(v), self.surfels_size, C.byref(sb),
                                               sup[0].ctypes.data_as(P(C.c_uint32)), sup[1].ctypes.data_as(P(C.c_uint32)),
                                               sup[2].ctypes.data_as(P(C.c_uint32)), C.byref(cnt))
        return cnt.value

    def delete_surfels_and_update_radii(self, min_observation_count, surfel_count):
        dp, sb, kfs = self.depth_params(), self.surfel_buf(), self.keyframe_views()
        cnt = C.c_uint32(surfel_count)
        lib().bso_delete_surfels_and_update_radii(min_observation_count, C.byref(self.depth_camera), C.byref(dp), len(self.keyframes), kfs,
                                                  C.byref(cnt), self.surfels_size, C.byref(sb))
        return cnt.value

    def compact_surfels(self, surfel_count, with_active=True):
        sb, ab = self.surfel_buf(), self.active_buf()
        size = C.c_uint32(self.surfels_size)
        lib().bso_compact_surfels(surfel_count, C.byref(size), C.byref(sb), C.byref(ab) if with_active else None)
        self.surfels_size = size.value

    # --- oracle calls on the host state
    def association(self, kf):
        out = np.zeros(max(1, self.surfels_size), np.uint32)
        dp, v, sb = self.depth_params(), kf.view(), self.surfel_buf()
        lib().bso_association(C.byref(self.depth_camera), C.byref(dp), C.byref(v), self.surfels_size, C.byref(sb),
                              out.ctypes.data_as(P(C.c_uint32)))
        return out[:self.surfels_size]

    def association_margins(self, kf):
        out = np.zeros((max(1, self.surfels_size), 5), np.float64)
        dp, v, sb = self.depth_params(), kf.view(), self.surfel_buf()
        lib().bso_association_margins(C.byref(self.depth_camera), C.byref(dp), C.byref(v), self.surfels_size, C.byref(sb),
                                      out.ctypes.data_as(P(C.c_double)))
        return out[:self.surfels_size]

    def accumulate_pose(self, kf, frame_T_global=None, per_surfel=False, use_depth=None, use_desc=None):
        use_depth = self.use_depth_residuals if use_depth is None else use_depth
        use_desc = self.use_descriptor_residuals if use_desc is None else use_desc
        H, b = np.zeros(21, np.float32), np.zeros(6, np.float32)
        H64, b64 = np.zeros(21, np.float64), np.zeros(6, np.float64)
        ps = np.zeros((max(1, self.surfels_size), 8), np.float32) if per_surfel else None
        cnt, cost = C.c_uint32(), C.c_float()
        dp, v, sb = self.depth_params(), kf.view(), self.surfel_buf()
        M = v.frame_T_global if frame_T_global is None else frame_T_global
        lib().bso_accumulate_pose_estimation_coeffs(
            int(use_depth), int(use_desc), C.byref(self.color_camera), C.byref(self.depth_camera), C.byref(dp),
            C.byref(v.depth), C.byref(v.normals), C.byref(v.color), C.byref(M), self.surfels_size, C.byref(sb),
            self.tex_mode, C.byref(cnt), C.byref(cost), fptr(H), fptr(b),
            H64.ctypes.data_as(P(C.c_double)), b64.ctypes.data_as(P(C.c_double)),
            fptr(ps) if ps is not None else None)
        return dict(H=H, b=b, H64=H64, b64=b64, count=cnt.value, cost=cost.value,
                    per_surfel=None if ps is None else ps[:self.surfels_size])

    def estimate_frame_pose(self, kf, initial, max_iterations=30):
        out = abi.SE3f()
        it, conv = C.c_int(), C.c_int()
        dp, v, sb = self.depth_params(), kf.view(), self.surfel_buf()
        lib().bso_estimate_frame_pose(
            int(self.use_depth_residuals), int(self.use_descriptor_residuals), C.byref(self.color_camera),
            C.byref(self.depth_camera), C.byref(dp), C.byref(v.depth), C.byref(v.normals), C.byref(v.color),
            C.byref(initial), self.surfels_size, C.byref(sb), self.tex_mode, max_iterations,
            C.byref(out), C.byref(it), C.byref(conv))
        return out, it.value, bool(conv.value)

    def update_activation(self):
        dp, sb, ab, kfs = self.depth_params(), self.surfel_buf(), self.active_buf(), self.keyframe_views()
        lib().bso_update_surfel_activation(C.byref(self.depth_camera), C.byref(dp), len(self.keyframes), kfs,
                                           self.surfels_size, C.byref(sb), C.byref(ab))

    def assign_colors(self):
        dp, sb, kfs = self.depth_params(), self.surfel_buf(), self.keyframe_views()
        lib().bso_assign_colors(C.byref(self.color_camera), C.byref(self.depth_camera), C.byref(dp), len(self.keyframes), kfs,
                                self.tex_mode, self.surfels_size, C.byref(sb))

    def update_normals(self):
        dp, sb, ab, kfs = self.depth_params(), self.surfel_buf(), self.active_buf(), self.keyframe_views()
        lib().bso_update_surfel_normals(C.byref(self.depth_camera), C.byref(dp), len(self.keyframes), kfs,
                                        self.surfels_size, C.byref(sb), C.byref(ab))

    def optimize_geometry_iteration(self):
        dp, sb, ab, kfs = self.depth_params(), self.surfel_buf(), self.active_buf(), self.keyframe_views()
        lib().bso_optimize_geometry_iteration(
            int(self.use_depth_residuals), int(self.use_descriptor_residuals), C.byref(self.color_camera),
            C.byref(self.depth_camera), C.byref(dp), len(self.keyframes), kfs, self.surfels_size, C.byref(sb), C.byref(ab),
            self.tex_mode)

    def optimize_intrinsics(self, optimize_depth, optimize_color):
        """One OptimizeIntrinsicsCUDA step on the host state (cameras, a, cfactor updated in place)."""
        dp, sb, kfs = self.depth_params(), self.surfel_buf(), self.keyframe_views()
        out_c, out_d = abi.Camera4f(), abi.Camera4f()
        a = C.c_float(self.a)
        lib().bso_optimize_intrinsics(int(optimize_depth), int(optimize_color), len(self.keyframes), kfs, C.byref(self.color_camera),
                                      C.byref(self.depth_camera), C.byref(dp), self.surfels_size, C.byref(sb), C.byref(out_c), C.byref(out_d),
                                      C.byref(a), self.tex_mode)
        if optimize_color:
            self.color_camera = out_c
        if optimize_depth:
            self.depth_camera = out_d
            self.a = a.value

    # --- upload
    def to_device(self, device="cuda:0"):
        return DeviceScene(self, device)


class DeviceScene:
    """The same scene in HBM: torch tensors own the memory, POD views point into them."""

    def __init__(self, host, device):
        import torch
        self.torch = torch
        self.host = host
        self.device = device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.surfels = t(host.surfels)
        self.active = t(host.active)
        self.cfactor = t(host.cfactor)
        self.kf_tensors = []
        for kf in host.keyframes:
            self.kf_tensors.append((t(kf.depth.view(np.int16)), t(kf.normals.view(np.int16)), t(kf.radius.view(np.int16)), t(kf.color)))
        self.surfels_size = host.surfels_size

    @staticmethod
    def tbuf(tensor, h=None, w=None):
        h = tensor.shape[0] if h is None else h
        w = tensor.shape[1] if w is None else w
        return abi.Buffer2D(tensor.data_ptr(), h, w, tensor.stride(0) * tensor.element_size())

    def depth_params(self):
        return self.host.depth_params(self.tbuf(self.cfactor))

    def surfel_buf(self):
        return self.tbuf(self.surfels)

    def active_buf(self):
        return self.tbuf(self.active)

    def keyframe_view(self, i):
        d, n, r, c = self.kf_tensors[i]
        return self.host.keyframes[i].view((self.tbuf(d), self.tbuf(n), self.tbuf(r), self.tbuf(c)))

    def keyframe_views(self):
        arr = (abi.KeyframeView * max(1, len(self.kf_tensors)))()
        for i in range(len(self.kf_tensors)):
            arr[i] = self.keyframe_view(i)
        return arr

    def surfels_np(self):
        return self.surfels.cpu().numpy()

    def active_np(self):
        return self.active.cpu().numpy()


# ----------------------------------------------------------------------------- PCG helpers

def pcg_layout(scene, optimize_poses=True, optimize_geometry=True, optimize_depth_intrinsics=False,
               optimize_color_intrinsics=False, gauge_keyframe_id=0):
    """Unknown ordering of BS/direct_ba_pcg.cc:270-306."""
    K = len(scene.keyframes)
    cur = 0
    if optimize_poses:
        cur += 6 * (K - 1)
    surfel_start = abi.INVALID_INDEX
    if optimize_geometry:
        surfel_start = cur
        cur += (3 if scene.use_descriptor_residuals else 1) * scene.surfels_size
    depth_start = a_index = abi.INVALID_INDEX
    if optimize_depth_intrinsics:
        depth_start = cur
        cur += 4 + 1 + scene.cfactor.shape[0] * scene.cfactor.shape[1]
        a_index = depth_start + 4
    color_start = abi.INVALID_INDEX
    if optimize_color_intrinsics:
        color_start = cur
        cur += 4
    return abi.PCGLayout(cur, surfel_start, depth_start, a_index, color_start, gauge_keyframe_id,
                         int(optimize_poses), int(optimize_geometry), int(optimize_depth_intrinsics),
                         int(optimize_color_intrinsics), int(scene.use_depth_residuals), int(scene.use_descriptor_residuals))


class HostPCG:
    """PCG vectors in host memory + the oracle's PCG steps (BS/kernel_pcg.cu restated)."""
    NAMES = ("r", "M", "delta", "g", "p")

    def __init__(self, scene, layout):
        self.scene, self.layout = scene, layout
        n = max(1, layout.unknown_count)
        for nm in self.NAMES:
            setattr(self, nm, np.zeros(n, np.float32))
        self.scalars = np.zeros(3, np.float32)   # alpha_n, alpha_d, beta_n
        self.an, self.bn = 0, 2

    def vectors(self):
        v = abi.PCGVectors()
        for nm in self.NAMES:
            setattr(v, nm, getattr(self, nm).ctypes.data)
        base = self.scalars.ctypes.data
        v.alpha_n, v.alpha_d, v.beta_n = base + 4 * self.an, base + 4, base + 4 * self.bn
        return v

    def swap_alpha_beta(self):   # std::swap(pcg_alpha_n_, pcg_beta_n_) BS/direct_ba_pcg.cc:390
        self.an, self.bn = self.bn, self.an

    def _args(self):
        s = self.scene
        return (C.byref(self.layout), C.byref(s.color_camera), C.byref(s.depth_camera))

    def init(self):
        s = self.scene
        dp, sb, kfs, v = s.depth_params(), s.surfel_buf(), s.keyframe_views(), self.vectors()
        lib().bso_pcg_init(*self._args(), C.byref(dp), len(s.keyframes), kfs, s.surfels_size, C.byref(sb), C.byref(v), s.tex_mode)

    def init2(self):
        v = self.vectors()
        lib().bso_pcg_init2(C.byref(self.layout), self.scene.a, C.byref(v))

    def step1(self, clear_g):
        s = self.scene
        dp, sb, kfs, v = s.depth_params(), s.surfel_buf(), s.keyframe_views(), self.vectors()
        lib().bso_pcg_step1(*self._args(), C.byref(dp), len(s.keyframes), kfs, s.surfels_size, C.byref(sb), C.byref(v), int(clear_g), s.tex_mode)

    def step2(self):
        v = self.vectors()
        out = C.c_float()
        lib().bso_pcg_step2(C.byref(self.layout), C.byref(v), C.byref(out))
        return out.value

    def step3(self):
        v = self.vectors()
        lib().bso_pcg_step3(C.byref(self.layout), C.byref(v))

    def apply_delta_to_surfels(self):
        s = self.scene
        sb = s.surfel_buf()
        lib().bso_update_surfels_from_pcg_delta(s.surfels_size, C.byref(sb), int(s.use_descriptor_residuals),
                                                self.layout.surfel_unknown_start_index, fptr(self.delta))

"""ctypes binding of the C++ host class bslam_host::DirectBA (badslam_amd/host/), the MI355X
counterpart of the reference's DirectBA (BS/direct_ba.h).  Method names follow the reference."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libbadslam_host.so")
_host = None


class DirectBAError(RuntimeError):
    pass


def host_lib():
    global _host
    if _host is not None:
        return _host
    from . import preload_torch_hip_runtime
    preload_torch_hip_runtime()
    if not os.path.exists(HOST_LIB_PATH):
        raise DirectBAError(f"{HOST_LIB_PATH} is missing: build it with `python -m badslam_amd.build`")
    L = C.CDLL(HOST_LIB_PATH)
    f32p, u16p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint16), C.POINTER(C.c_uint8)
    L.bsh_last_error.restype = C.c_char_p
    L.bsh_create.restype = C.c_void_p
    L.bsh_create.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                             f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.bsh_destroy.argtypes = [C.c_void_p]
    L.bsh_context.restype = C.c_void_p
    L.bsh_context.argtypes = [C.c_void_p]
    L.bsh_add_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float, u16p, u16p, u16p, u8p, f32p]
    L.bsh_add_keyframe_from_images.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, u16p, u8p, f32p]
    L.bsh_get_keyframe_images.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u16p, u16p, u16p, u8p, f32p]
    L.bsh_set_surfels.argtypes = [C.c_void_p, C.c_void_p, f32p, C.c_size_t, C.c_uint32]
    L.bsh_get_surfels.argtypes = [C.c_void_p, C.c_void_p, f32p, C.c_size_t, C.c_int]
    L.bsh_get_active_surfels.argtypes = [C.c_void_p, C.c_void_p, u8p]
    L.bsh_surfels_size.restype = C.c_uint32
    L.bsh_surfels_size.argtypes = [C.c_void_p]
    L.bsh_keyframe_count.argtypes = [C.c_void_p]
    L.bsh_get_keyframe_pose.argtypes = [C.c_void_p, C.c_int, f32p]
    L.bsh_set_keyframe_pose.argtypes = [C.c_void_p, C.c_int, f32p]
    L.bsh_get_keyframe_activation.argtypes = [C.c_void_p, C.c_int]
    L.bsh_set_keyframe_activation.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.bsh_keyframe_covisibility.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.bsh_upload_keyframe_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u16p]
    L.bsh_upload_keyframe_normals.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u16p]
    L.bsh_set_options.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.bsh_set_scheme_end_tasks.argtypes = [C.c_void_p, C.c_int]
    L.bsh_set_timings_file.argtypes = [C.c_void_p, C.c_char_p]
    L.bsh_keyframe_is_deleted.argtypes = [C.c_void_p, C.c_int]
    L.bsh_delete_keyframe.argtypes = [C.c_void_p, C.c_int]
    L.bsh_merge_keyframes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    L.bsh_update_keyframe_covisibility.argtypes = [C.c_void_p, C.c_int]
    L.bsh_assign_colors.argtypes = [C.c_void_p, C.c_void_p]
    L.bsh_export_point_cloud.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, f32p, u8p, f32p, C.POINTER(C.c_uint64)]
    L.bsh_create_surfels_for_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.bsh_set_allreduce.argtypes = [C.c_void_p, abi.ALLREDUCE_FN, C.c_void_p]
    L.bsh_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.bsh_comm_destroy.argtypes = [C.c_void_p]
    L.bsh_estimate_frame_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_int, f32p, f32p]
    L.bsh_bundle_adjustment.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 12 + [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bsh_get_intrinsics.argtypes = [C.c_void_p, f32p, f32p, f32p]
    L.bsh_set_intrinsics.argtypes = [C.c_void_p, f32p, f32p, C.c_float]
    L.bsh_get_cfactor.argtypes = [C.c_void_p, C.c_void_p, f32p]
    _host = L
    return L


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def pose7(se3f):
    return np.array(list(se3f.q) + list(se3f.t), np.float32)


def se3f_from7(p):
    T = abi.SE3f()
    for i in range(4):
        T.q[i] = float(p[i])
    for i in range(3):
        T.t[i] = float(p[4 + i])
    return T


# ---- file formats (badslam_amd/host/io.hpp) -----------------------------------------------------------------
def _io_lib():
    L = host_lib()
    if getattr(L, "_io_ready", False):
        return L
    L.bsh_png_info.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    L.bsh_read_png_gray16.argtypes = [C.c_char_p, C.POINTER(C.c_uint16), C.c_size_t]
    L.bsh_read_png_rgb8.argtypes = [C.c_char_p, C.POINTER(C.c_uint8), C.c_size_t]
    L.bsh_tum_open.restype = C.c_void_p
    L.bsh_tum_open.argtypes = [C.c_char_p, C.c_char_p]
    L.bsh_tum_close.argtypes = [C.c_void_p]
    L.bsh_tum_frame_count.argtypes = [C.c_void_p]
    L.bsh_tum_camera.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bsh_tum_frame.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.bsh_save_poses.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int, C.c_char_p]
    L.bsh_save_calibration_arrays.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.bsh_load_calibration_arrays.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.bsh_save_calibration.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
    L.bsh_load_calibration.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
    L._io_ready = True
    return L


def _io_check(L, rc):
    if rc != 0:
        raise DirectBAError(L.bsh_last_error().decode())


def read_png(path):
    """16-bit gray -> (h, w) uint16; 8-bit colour / gray -> (h, w, 3) uint8."""
    L = _io_lib()
    info = (C.c_int * 4)()
    _io_check(L, L.bsh_png_info(str(path).encode(), info))
    w, h, depth, channels = info[0], info[1], info[2], info[3]
    if channels == 1 and depth == 16:
        out = np.zeros((h, w), np.uint16)
        _io_check(L, L.bsh_read_png_gray16(str(path).encode(), out.ctypes.data_as(C.POINTER(C.c_uint16)), out.size))
        return out
    out = np.zeros((h, w, 3), np.uint8)
    _io_check(L, L.bsh_read_png_rgb8(str(path).encode(), out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size))
    return out


def read_tum_dataset(folder, trajectory_filename=None):
    """ReadTUMRGBDDatasetAssociatedAndCalibrated: dict(width, height, camera (corner convention), frames=[dict])."""
    L = _io_lib()
    h = L.bsh_tum_open(str(folder).encode(), trajectory_filename.encode() if trajectory_filename else None)
    if not h:
        raise DirectBAError(L.bsh_last_error().decode())
    try:
        cam = (C.c_float * 4)()
        w, ht = C.c_int(), C.c_int()
        L.bsh_tum_camera(h, cam, C.byref(w), C.byref(ht))
        frames = []
        for i in range(L.bsh_tum_frame_count(h)):
            bufs = [C.create_string_buffer(512) for _ in range(4)]
            p_rgb, p_depth = np.zeros(7, np.float32), np.zeros(7, np.float32)
            _io_check(L, L.bsh_tum_frame(h, i, bufs[0], bufs[1], bufs[2], bufs[3], 512, _f(p_rgb), _f(p_depth)))
            frames.append(dict(rgb_path=bufs[0].value.decode(), depth_path=bufs[1].value.decode(), rgb_timestamp=bufs[2].value.decode(),
                               depth_timestamp=bufs[3].value.decode(), rgb_global_T_frame=p_rgb, depth_global_T_frame=p_depth))
        return dict(width=w.value, height=ht.value, camera=np.array(list(cam), np.float32), frames=frames)
    finally:
        L.bsh_tum_close(h)


def save_poses(timestamp_strings, poses7, start_frame, path):
    """SavePoses (BS/io.cc:537-568); poses7 = (n, 7) rows qx qy qz qw tx ty tz of global_T_frame."""
    L = _io_lib()
    n = len(timestamp_strings)
    arr = (C.c_char_p * n)(*[t.encode() for t in timestamp_strings])
    p = np.ascontiguousarray(poses7, np.float32)
    _io_check(L, L.bsh_save_poses(n, arr, _f(p), start_frame, str(path).encode()))


def save_calibration_arrays(base, depth4, color4, a, cfactor):
    L = _io_lib()
    cf = np.ascontiguousarray(cfactor, np.float32)
    _io_check(L, L.bsh_save_calibration_arrays(str(base).encode(), _f(np.ascontiguousarray(depth4, np.float32)), _f(np.ascontiguousarray(color4, np.float32)),
                                               a, cf.shape[1], cf.shape[0], _f(cf)))


def load_calibration_arrays(base, cfactor_shape):
    L = _io_lib()
    d, c, a = np.zeros(4, np.float32), np.zeros(4, np.float32), C.c_float()
    cf = np.zeros(cfactor_shape, np.float32)
    _io_check(L, L.bsh_load_calibration_arrays(str(base).encode(), _f(d), _f(c), C.byref(a), cf.shape[1], cf.shape[0], _f(cf)))
    return d, c, a.value, cf


def state_file_round_trip(src, dst):
    """LoadState + SaveState of a version-1 state file; returns the summary (ints[8], floats[11])."""
    L = _io_lib()
    L.bsh_state_load.restype = C.c_void_p
    L.bsh_state_load.argtypes = [C.c_char_p]
    L.bsh_state_free.argtypes = [C.c_void_p]
    L.bsh_state_save.argtypes = [C.c_void_p, C.c_char_p]
    L.bsh_state_summary.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
    h = L.bsh_state_load(str(src).encode())
    if not h:
        raise DirectBAError(L.bsh_last_error().decode())
    try:
        ints, floats = np.zeros(8, np.int32), np.zeros(11, np.float32)
        L.bsh_state_summary(h, ints.ctypes.data_as(C.POINTER(C.c_int32)), _f(floats))
        _io_check(L, L.bsh_state_save(h, str(dst).encode()))
        return ints, floats
    finally:
        L.bsh_state_free(h)


class DirectBA:
    """Drives bslam_host::DirectBA.  Constructor arguments are those of BS/direct_ba.h:73-88."""

    def __init__(self, max_surfel_count, raw_to_float_depth, baseline_fx, sparse_surfel_cell_size, surfel_merge_dist_factor,
                 min_observation_count_while_bootstrapping_1, min_observation_count_while_bootstrapping_2, min_observation_count,
                 color_camera, depth_camera, pyramid_level_for_color, use_depth_residuals, use_descriptor_residuals, device=0, stream=None):
        self.L = host_lib()
        cc = np.array([color_camera.fx, color_camera.fy, color_camera.cx, color_camera.cy], np.float32)
        dc = np.array([depth_camera.fx, depth_camera.fy, depth_camera.cx, depth_camera.cy], np.float32)
        self._ba = self.L.bsh_create(max_surfel_count, raw_to_float_depth, baseline_fx, sparse_surfel_cell_size, surfel_merge_dist_factor,
                                     min_observation_count_while_bootstrapping_1, min_observation_count_while_bootstrapping_2,
                                     min_observation_count, _f(cc), color_camera.width, color_camera.height, _f(dc), depth_camera.width,
                                     depth_camera.height, pyramid_level_for_color, int(use_depth_residuals), int(use_descriptor_residuals), device)
        if not self._ba:
            raise DirectBAError(self.L.bsh_last_error().decode())
        self.stream = C.c_void_p(stream) if stream else C.c_void_p(None)

    def context_handle(self):
        """bslam_context* of the kernel library used by this DirectBA (for bslam_profile_*)."""
        return C.c_void_p(self.L.bsh_context(self._ba))

    def _check(self, rc):
        if rc < 0:
            raise DirectBAError(self.L.bsh_last_error().decode())
        return rc

    def close(self):
        if self._ba:
            self.L.bsh_destroy(self._ba)
            self._ba = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def AddKeyframe(self, frame_index, min_depth, max_depth, depth, normals, radius, color, global_T_frame):
        p = pose7(global_T_frame)
        u16 = lambda a: np.ascontiguousarray(a, np.uint16).ctypes.data_as(C.POINTER(C.c_uint16))
        col = np.ascontiguousarray(color, np.uint8)
        return self._check(self.L.bsh_add_keyframe(self._ba, self.stream, frame_index, min_depth, max_depth, u16(depth), u16(normals), u16(radius),
                                                   col.ctypes.data_as(C.POINTER(C.c_uint8)), _f(p)))

    def AddKeyframeFromImages(self, frame_index, depth_u16, rgb_u8, global_T_frame):
        """Keyframe(stream, frame_index, depth_params, depth_camera, depth_image, color_image, pose) + AddKeyframe: the raw
        images are preprocessed on the device (BS/keyframe.cc:82-161)."""
        d = np.ascontiguousarray(depth_u16, np.uint16)
        rgb = np.ascontiguousarray(rgb_u8, np.uint8)
        return self._check(self.L.bsh_add_keyframe_from_images(self._ba, self.stream, frame_index, d.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                               rgb.ctypes.data_as(C.POINTER(C.c_uint8)), _f(pose7(global_T_frame))))

    def keyframe_images(self, kf_id, height, width):
        u16 = lambda: np.zeros((height, width), np.uint16)
        depth, normals, radius, color, mm = u16(), u16(), u16(), np.zeros((height, width, 4), np.uint8), np.zeros(2, np.float32)
        p16 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint16))
        self._check(self.L.bsh_get_keyframe_images(self._ba, self.stream, kf_id, p16(depth), p16(normals), p16(radius),
                                                   color.ctypes.data_as(C.POINTER(C.c_uint8)), _f(mm)))
        return depth, normals, radius, color, float(mm[0]), float(mm[1])

    def TrackKeyframePair(self, tracked_id, base_id, init1, init2=None, num_scales=5, test_different_initial_estimates=False,
                          use_pyramid_level_0=True, use_gradmag=False):
        """TrackFramePairwise (BS/pairwise_frame_tracking.cc:256-678) of keyframe `tracked_id` against keyframe `base_id`:
        returns (base_T_tracked, iterations per scale)."""
        self.L.bsh_track_keyframe_pair_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                      C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int, C.c_int]
        out = np.zeros(7, np.float32)
        its = (C.c_int * num_scales)()
        p1 = pose7(init1)
        p2 = pose7(init2 if init2 is not None else init1)
        self._check(self.L.bsh_track_keyframe_pair_ex(self._ba, self.stream, tracked_id, base_id, num_scales, int(test_different_initial_estimates), _f(p1),
                                                      _f(p2), _f(out), its, int(use_pyramid_level_0), int(use_gradmag)))
        return se3f_from7(out), list(its)

    def SaveState(self, path, frame_count):
        """The DirectBA part of SaveState (BS/io.cc:38-178) as a version-1 state file."""
        L = _io_lib()
        L.bsh_state_save_from_ba.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p]
        self._check(L.bsh_state_save_from_ba(self._ba, self.stream, frame_count, str(path).encode()))

    def LoadState(self, path):
        L = _io_lib()
        L.bsh_state_load_into_ba.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        self._check(L.bsh_state_load_into_ba(self._ba, self.stream, str(path).encode()))

    def SaveCalibration(self, base_path):
        L = _io_lib()
        self._check(L.bsh_save_calibration(self._ba, self.stream, str(base_path).encode()))

    def LoadCalibration(self, base_path):
        L = _io_lib()
        self._check(L.bsh_load_calibration(self._ba, self.stream, str(base_path).encode()))

    def SetSurfels(self, rows, count):
        rows = np.ascontiguousarray(rows, np.float32)
        self._check(self.L.bsh_set_surfels(self._ba, self.stream, _f(rows), rows.strides[0], count))

    def GetSurfels(self, nrows=8):
        n = self.surfels_size()
        out = np.zeros((nrows, max(1, n)), np.float32)
        self._check(self.L.bsh_get_surfels(self._ba, self.stream, _f(out), out.strides[0], nrows))
        return out[:, :n]

    def GetActiveSurfels(self):
        n = self.surfels_size()
        out = np.zeros(max(1, n), np.uint8)
        self._check(self.L.bsh_get_active_surfels(self._ba, self.stream, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out[:n]

    def surfels_size(self):
        return self.L.bsh_surfels_size(self._ba)

    def keyframe_pose(self, kf_id):
        out = np.zeros(7, np.float32)
        self._check(self.L.bsh_get_keyframe_pose(self._ba, kf_id, _f(out)))
        return se3f_from7(out)

    def set_keyframe_pose(self, kf_id, T):
        self._check(self.L.bsh_set_keyframe_pose(self._ba, kf_id, _f(pose7(T))))

    def keyframe_activation(self, kf_id):
        return self.L.bsh_get_keyframe_activation(self._ba, kf_id)

    def set_keyframe_activation(self, kf_id, act):
        self._check(self.L.bsh_set_keyframe_activation(self._ba, kf_id, act))

    def keyframe_count(self):
        """keyframes().size(): includes deleted (null) entries."""
        return self.L.bsh_keyframe_count(self._ba)

    def keyframe_covisibility(self, kf_id):
        buf = (C.c_int * 4096)()
        n = self.L.bsh_keyframe_covisibility(self._ba, kf_id, buf, 4096)
        return list(buf[:n])

    # --- keyframe management and export (BS/direct_ba.h:95-126)
    def keyframe_is_deleted(self, kf_id):
        return bool(self.L.bsh_keyframe_is_deleted(self._ba, kf_id))

    def DeleteKeyframe(self, kf_id):
        self._check(self.L.bsh_delete_keyframe(self._ba, kf_id))

    def MergeKeyframes(self, approx_merge_count):
        """Returns the ids of the deleted keyframes."""
        ids = (C.c_int * 4096)()
        n = C.c_int()
        self._check(self.L.bsh_merge_keyframes(self._ba, self.stream, approx_merge_count, ids, 4096, C.byref(n)))
        return list(ids[:n.value])

    def UpdateKeyframeCoVisibility(self, kf_id):
        self._check(self.L.bsh_update_keyframe_covisibility(self._ba, kf_id))

    def AssignColors(self):
        self._check(self.L.bsh_assign_colors(self._ba, self.stream))

    def ExportToPointCloud(self):
        """(positions (n, 3) f32, colors (n, 3) u8, normals (n, 3) f32) of the valid surfels."""
        cap = max(1, self.surfels_size())
        pos, col, nrm = np.zeros((cap, 3), np.float32), np.zeros((cap, 3), np.uint8), np.zeros((cap, 3), np.float32)
        n = C.c_uint64()
        self._check(self.L.bsh_export_point_cloud(self._ba, self.stream, cap, _f(pos), col.ctypes.data_as(C.POINTER(C.c_uint8)), _f(nrm), C.byref(n)))
        return pos[:n.value], col[:n.value], nrm[:n.value]

    def upload_keyframe_depth(self, kf_id, depth):
        d = np.ascontiguousarray(depth, np.uint16)
        self._check(self.L.bsh_upload_keyframe_depth(self._ba, self.stream, kf_id, d.ctypes.data_as(C.POINTER(C.c_uint16))))

    def upload_keyframe_normals(self, kf_id, normals):
        d = np.ascontiguousarray(normals, np.uint16)
        self._check(self.L.bsh_upload_keyframe_normals(self._ba, self.stream, kf_id, d.ctypes.data_as(C.POINTER(C.c_uint16))))

    def set_options(self, batched_pose_optimization=True, pcg_gauge_keyframe=-1, texture_mode=abi.TEX_FIXED_POINT_1_8, scheme_end_tasks=True):
        self._check(self.L.bsh_set_options(self._ba, int(batched_pose_optimization), pcg_gauge_keyframe, texture_mode))
        self._check(self.L.bsh_set_scheme_end_tasks(self._ba, int(scheme_end_tasks)))

    def set_timings_file(self, path):
        """--save_timings: one block of BA_* lines per BA iteration (BS/direct_ba_alternating.cc:630-688); None stops."""
        self._check(self.L.bsh_set_timings_file(self._ba, path.encode() if path else None))

    def CreateSurfelsForKeyframe(self, filter_new_surfels, kf_id):
        self._check(self.L.bsh_create_surfels_for_keyframe(self._ba, self.stream, int(filter_new_surfels), kf_id))

    def set_allreduce(self, callback):
        self._check(self.L.bsh_set_allreduce(self._ba, callback, None))

    def InitComm(self, unique_id, rank, world_size):
        """RCCL communicator inside this DirectBA's kernel context (unique_id: the 128 bytes of badslam_amd.comm_unique_id() of rank 0)."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.L.bsh_comm_init(self._ba, buf, rank, world_size))

    def DestroyComm(self):
        self._check(self.L.bsh_comm_destroy(self._ba))

    def EstimateFramePose(self, kf_id, global_T_frame_initial_estimate):
        out = np.zeros(7, np.float32)
        self._check(self.L.bsh_estimate_frame_pose(self._ba, self.stream, kf_id, _f(pose7(global_T_frame_initial_estimate)), _f(out)))
        return se3f_from7(out)

    def BundleAdjustment(self, optimize_depth_intrinsics, optimize_color_intrinsics, do_surfel_updates, optimize_poses, optimize_geometry,
                         min_iterations, max_iterations, use_pcg, active_keyframe_window_start, active_keyframe_window_end,
                         increase_ba_iteration_count, pcg_max_inner_iterations=30):
        it, conv = C.c_int(), C.c_int()
        self._check(self.L.bsh_bundle_adjustment(self._ba, self.stream, int(optimize_depth_intrinsics), int(optimize_color_intrinsics),
                                                 int(do_surfel_updates), int(optimize_poses), int(optimize_geometry), min_iterations,
                                                 max_iterations, int(use_pcg), active_keyframe_window_start, active_keyframe_window_end,
                                                 int(increase_ba_iteration_count), pcg_max_inner_iterations, C.byref(it), C.byref(conv)))
        return it.value, bool(conv.value)

    def set_intrinsics(self, color4=None, depth4=None, a=0.0):
        c = None if color4 is None else _f(np.ascontiguousarray(color4, np.float32))
        d = None if depth4 is None else _f(np.ascontiguousarray(depth4, np.float32))
        self._check(self.L.bsh_set_intrinsics(self._ba, c, d, a))

    def cfactor(self, shape):
        out = np.zeros(shape, np.float32)
        self._check(self.L.bsh_get_cfactor(self._ba, self.stream, _f(out)))
        return out

    def intrinsics(self):
        c, d, a = np.zeros(4, np.float32), np.zeros(4, np.float32), C.c_float()
        self._check(self.L.bsh_get_intrinsics(self._ba, _f(c), _f(d), C.byref(a)))
        return c, d, a.value

"""ctypes binding of bslam_host::BadSlam (badslam_amd/host/bad_slam.hpp): the sequential front end of the reference's
vis::BadSlam (BS/bad_slam.cc) -- preprocessing, pairwise odometry with the constant-motion estimates, keyframe
scheduling and the planned bundle-adjustment iterations -- without threads, GUI and loop detection."""
import ctypes as C

import numpy as np

from . import direct_ba as dba


class BadSlam:
    def __init__(self, color_camera, depth_camera, keyframe_interval=10, max_num_ba_iterations_per_keyframe=10, num_scales=5,
                 max_surfel_count=25 * 1000 * 1000, sparse_surfel_cell_size=4, use_motion_model=True, use_geometric_residuals=True,
                 use_photometric_residuals=True, do_surfel_updates=True, use_pcg=False, optimize_intrinsics=False, disable_deactivation=False,
                 start_frame=0, raw_to_float_depth=1.0 / 5000, max_depth=3.0, baseline_fx=40.0, device=0):
        """Keyword defaults = BS/bad_slam_config.h."""
        self.L = dba.host_lib()
        L = self.L
        f32p = C.POINTER(C.c_float)
        L.bsh_slam_create.restype = C.c_void_p
        L.bsh_slam_create.argtypes = [C.POINTER(C.c_int), f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, f32p, C.c_int]
        L.bsh_slam_destroy.argtypes = [C.c_void_p]
        L.bsh_slam_direct_ba.restype = C.c_void_p
        L.bsh_slam_direct_ba.argtypes = [C.c_void_p]
        L.bsh_slam_process_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint16), C.POINTER(C.c_uint8), C.c_int]
        L.bsh_slam_run_bundle_adjustment.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.bsh_slam_frame_count.argtypes = [C.c_void_p]
        L.bsh_slam_get_frame_poses.argtypes = [C.c_void_p, f32p, C.c_int]
        L.bsh_slam_state.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        cfg = (C.c_int * 13)(keyframe_interval, max_num_ba_iterations_per_keyframe, num_scales, max_surfel_count, sparse_surfel_cell_size,
                             int(use_motion_model), int(use_geometric_residuals), int(use_photometric_residuals), int(do_surfel_updates), int(use_pcg),
                             int(optimize_intrinsics), int(disable_deactivation), start_frame)
        fcfg = (C.c_float * 3)(raw_to_float_depth, max_depth, baseline_fx)
        cc = np.array([color_camera.fx, color_camera.fy, color_camera.cx, color_camera.cy], np.float32)
        dc = np.array([depth_camera.fx, depth_camera.fy, depth_camera.cx, depth_camera.cy], np.float32)
        self._slam = L.bsh_slam_create(cfg, fcfg, color_camera.width, color_camera.height, dba._f(cc), depth_camera.width, depth_camera.height,
                                       dba._f(dc), device)
        if not self._slam:
            raise dba.DirectBAError(L.bsh_last_error().decode())
        # non-owning view of the DirectBA inside
        self.direct_ba = dba.DirectBA.__new__(dba.DirectBA)
        self.direct_ba.L = L
        self.direct_ba._ba = None                       # close() / __del__ of the view must not destroy it
        self.direct_ba.stream = C.c_void_p(None)
        self.direct_ba.close = lambda: None             # instance attribute: shadows DirectBA.close for this view only
        self._ba_ptr = L.bsh_slam_direct_ba(self._slam)

    def ba(self):
        """The DirectBA of this BadSlam (valid while the BadSlam object lives)."""
        view = self.direct_ba
        view._ba = self._ba_ptr
        return _BorrowedBA(view, self)

    def _check(self, rc):
        if rc < 0:
            raise dba.DirectBAError(self.L.bsh_last_error().decode())
        return rc

    def close(self):
        if self._slam:
            self.direct_ba._ba = None
            self.L.bsh_slam_destroy(self._slam)
            self._slam = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ProcessFrame(self, frame_index, depth_u16, rgb_u8, force_keyframe=False):
        d = np.ascontiguousarray(depth_u16, np.uint16)
        rgb = np.ascontiguousarray(rgb_u8, np.uint8)
        self._check(self.L.bsh_slam_process_frame(self._slam, frame_index, d.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                  rgb.ctypes.data_as(C.POINTER(C.c_uint8)), int(force_keyframe)))

    def RunBundleAdjustment(self, frame_index, optimize_depth_intrinsics, optimize_color_intrinsics, optimize_poses, optimize_geometry,
                            min_iterations, max_iterations, window_start=-1, window_end=-1, increase_ba_iteration_count=True):
        done, conv = C.c_int(), C.c_int()
        self._check(self.L.bsh_slam_run_bundle_adjustment(self._slam, frame_index, int(optimize_depth_intrinsics), int(optimize_color_intrinsics),
                                                          int(optimize_poses), int(optimize_geometry), min_iterations, max_iterations, window_start,
                                                          window_end, int(increase_ba_iteration_count), C.byref(done), C.byref(conv)))
        return done.value, bool(conv.value)

    def frame_poses(self):
        """(n, 7) rows qx qy qz qw tx ty tz: global_T_frame of every processed frame."""
        n = self.L.bsh_slam_frame_count(self._slam)
        out = np.zeros((max(1, n), 7), np.float32)
        self._check(self.L.bsh_slam_get_frame_poses(self._slam, dba._f(out), n))
        return out[:n]

    def state(self):
        s = (C.c_int * 5)()
        self._check(self.L.bsh_slam_state(self._slam, s))
        return dict(keyframe_created=bool(s[0]), pose_estimated=bool(s[1]), num_planned_ba_iterations=s[2], base_kf_id=s[3], motion_model_length=s[4])


class _BorrowedBA:
    """Forwards to the DirectBA wrapper; keeps the owning BadSlam alive and never destroys the native object."""

    def __init__(self, view, owner):
        object.__setattr__(self, "_view", view)
        object.__setattr__(self, "_owner", owner)

    def __getattr__(self, name):
        if name in ("close", "__del__"):
            raise AttributeError(name)
        return getattr(self._view, name)

"""Helpers for the -m gpu parity tests: calls into the HIP library through the C ABI."""
import ctypes as C

import numpy as np

import badslam_amd
from badslam_amd import abi

P = C.POINTER


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Hip:
    """Thin call layer over include/badslam_hip.h for a tests.bso.DeviceScene."""

    def __init__(self, dev_scene, ctx=None):
        import torch
        self.torch = torch
        self.d = dev_scene
        self.h = dev_scene.host
        self.L = badslam_amd.lib()
        self.ctx = ctx or badslam_amd.Context(0)
        self.ctx.set_texture_mode(self.h.tex_mode)

    def _common(self):
        return self.d.depth_params(), self.d.surfel_buf()

    def association(self, kf_index):
        dp, sb = self._common()
        v = self.d.keyframe_view(kf_index)
        out = self.torch.empty(max(1, self.d.surfels_size), dtype=self.torch.int32, device=self.d.device)
        badslam_amd.check(self.L.bslam_debug_association(self.ctx.handle, stream_ptr(), C.byref(self.h.depth_camera), C.byref(dp),
                                                         C.byref(v), self.d.surfels_size, C.byref(sb), C.c_void_p(out.data_ptr())))
        self.torch.cuda.synchronize()
        return out.cpu().numpy().view(np.uint32)[:self.d.surfels_size]

    def residual_probe(self, kf_index, use_depth=None, use_desc=None):
        use_depth = self.h.use_depth_residuals if use_depth is None else use_depth
        use_desc = self.h.use_descriptor_residuals if use_desc is None else use_desc
        dp, sb = self._common()
        v = self.d.keyframe_view(kf_index)
        out = self.torch.zeros((max(1, self.d.surfels_size), 8), dtype=self.torch.float32, device=self.d.device)
        badslam_amd.check(self.L.bslam_debug_pose_residuals(
            self.ctx.handle, stream_ptr(), int(use_depth), int(use_desc), C.byref(self.h.color_camera), C.byref(self.h.depth_camera),
            C.byref(dp), C.byref(v), self.d.surfels_size, C.byref(sb), C.c_void_p(out.data_ptr())))
        self.torch.cuda.synchronize()
        return out.cpu().numpy()[:self.d.surfels_size]

    def accumulate_pose(self, kf_index, frame_T_global=None, use_depth=None, use_desc=None):
        use_depth = self.h.use_depth_residuals if use_depth is None else use_depth
        use_desc = self.h.use_descriptor_residuals if use_desc is None else use_desc
        dp, sb = self._common()
        v = self.d.keyframe_view(kf_index)
        M = v.frame_T_global if frame_T_global is None else frame_T_global
        H, b = np.zeros(21, np.float32), np.zeros(6, np.float32)
        cnt, cost = C.c_uint32(), C.c_float()
        badslam_amd.check(self.L.bslam_accumulate_pose_estimation_coeffs(
            self.ctx.handle, stream_ptr(), int(use_depth), int(use_desc), C.byref(self.h.color_camera), C.byref(self.h.depth_camera),
            C.byref(dp), C.byref(v.depth), C.byref(v.normals), C.byref(v.color), C.byref(M), self.d.surfels_size, C.byref(sb),
            1, C.byref(cnt), C.byref(cost), H.ctypes.data_as(P(C.c_float)), b.ctypes.data_as(P(C.c_float))))
        return dict(H=H, b=b, count=cnt.value, cost=cost.value)

    def accumulate_pose_batched(self, use_depth=None, use_desc=None, download=True):
        use_depth = self.h.use_depth_residuals if use_depth is None else use_depth
        use_desc = self.h.use_descriptor_residuals if use_desc is None else use_desc
        dp, sb = self._common()
        kfs = self.d.keyframe_views()
        K = len(self.h.keyframes)
        Hb = np.zeros((K, 27), np.float32)
        counts = np.zeros(K, np.uint32)
        badslam_amd.check(self.L.bslam_accumulate_pose_coeffs_batched(
            self.ctx.handle, stream_ptr(), int(use_depth), int(use_desc), C.byref(self.h.color_camera), C.byref(self.h.depth_camera),
            C.byref(dp), K, kfs, self.d.surfels_size, C.byref(sb),
            Hb.ctypes.data_as(P(C.c_float)) if download else None, counts.ctypes.data_as(P(C.c_uint32)) if download else None))
        return Hb, counts

    def estimate_poses_batched(self, initial_poses, max_iterations=30, activations=None, allreduce=None):
        dp, sb = self._common()
        kfs = self.d.keyframe_views()
        K = len(self.h.keyframes)
        if activations is not None:
            for k in range(K):
                kfs[k].activation = activations[k]
        poses = (abi.SE3f * K)()
        for k in range(K):
            C.memmove(C.byref(poses[k]), C.byref(initial_poses[k]), C.sizeof(abi.SE3f))
        iters = (C.c_int32 * K)()
        conv = (C.c_int32 * K)()
        cb = allreduce if allreduce is not None else C.cast(None, abi.ALLREDUCE_FN)
        badslam_amd.check(self.L.bslam_estimate_frame_poses_batched(
            self.ctx.handle, stream_ptr(), int(self.h.use_depth_residuals), int(self.h.use_descriptor_residuals),
            C.byref(self.h.color_camera), C.byref(self.h.depth_camera), C.byref(dp), K, kfs, self.d.surfels_size, C.byref(sb),
            max_iterations, poses, iters, conv, cb, None))
        return [poses[k] for k in range(K)], list(iters), list(conv)

    def optimize_intrinsics(self, optimize_depth, optimize_color):
        """bslam_optimize_intrinsics on the device scene; returns (color_camera, depth_camera, a)."""
        dp, sb = self._common()
        kfs = self.d.keyframe_views()
        out_c, out_d = abi.Camera4f(), abi.Camera4f()
        a = C.c_float(self.h.a)
        badslam_amd.check(self.L.bslam_optimize_intrinsics(
            self.ctx.handle, stream_ptr(), int(optimize_depth), int(optimize_color), len(self.h.keyframes), kfs, C.byref(self.h.color_camera),
            C.byref(self.h.depth_camera), C.byref(dp), self.d.surfels_size, C.byref(sb), C.byref(out_c), C.byref(out_d), C.byref(a)))
        self.torch.cuda.synchronize()
        return out_c, out_d, a.value

    def update_activation(self):
        dp, sb = self._common()
        ab, kfs = self.d.active_buf(), self.d.keyframe_views()
        badslam_amd.check(self.L.bslam_update_surfel_activation(self.ctx.handle, stream_ptr(), C.byref(self.h.depth_camera), C.byref(dp),
                                                                len(self.h.keyframes), kfs, self.d.surfels_size, C.byref(sb), C.byref(ab)))
        self.torch.cuda.synchronize()

    def update_normals(self):
        dp, sb = self._common()
        ab, kfs = self.d.active_buf(), self.d.keyframe_views()
        badslam_amd.check(self.L.bslam_update_surfel_normals(self.ctx.handle, stream_ptr(), C.byref(self.h.depth_camera), C.byref(dp),
                                                             len(self.h.keyframes), kfs, self.d.surfels_size, C.byref(sb), C.byref(ab)))
        self.torch.cuda.synchronize()

    def optimize_geometry_iteration(self):
        dp, sb = self._common()
        ab, kfs = self.d.active_buf(), self.d.keyframe_views()
        badslam_amd.check(self.L.bslam_optimize_geometry_iteration(
            self.ctx.handle, stream_ptr(), int(self.h.use_depth_residuals), int(self.h.use_descriptor_residuals),
            C.byref(self.h.color_camera), C.byref(self.h.depth_camera), C.byref(dp), len(self.h.keyframes), kfs,
            self.d.surfels_size, C.byref(sb), C.byref(ab)))
        self.torch.cuda.synchronize()


    def assign_colors(self):
        dp, sb = self._common()
        kfs = self.d.keyframe_views()
        badslam_amd.check(self.L.bslam_assign_colors(
            self.ctx.handle, stream_ptr(), C.byref(self.h.color_camera), C.byref(self.h.depth_camera), C.byref(dp),
            len(self.h.keyframes), kfs, self.d.surfels_size, C.byref(sb)))
        self.torch.cuda.synchronize()

    # --- surfel lifecycle
    def create_surfels_for_keyframe(self, kf_index, filter_new_surfels, min_observation_count, covis_indices):
        dp, sb = self._common()
        kf = self.h.keyframes[kf_index]
        v = self.d.keyframe_view(kf_index)
        n = len(covis_indices)
        views = (abi.KeyframeView * max(1, n))()
        for i, ci in enumerate(covis_indices):
            views[i] = self.d.keyframe_view(ci)
        from tests import bso
        _, _, mats = self.h.covis_args(kf, [self.h.keyframes[ci] for ci in covis_indices])
        G = bso.se3_matrix3x4(kf.global_T_frame)
        created = C.c_uint32()
        badslam_amd.check(self.L.bslam_create_surfels_for_keyframe(
            self.ctx.handle, stream_ptr(), int(filter_new_surfels), min_observation_count, C.byref(self.h.color_camera), C.byref(self.h.depth_camera),
            C.byref(dp), C.byref(v), C.byref(G), n, views, mats, self.d.surfels_size, C.byref(sb), C.byref(created)))
        self.d.surfels_size += created.value
        return created.value

    def merge_surfels(self, kf_index, merge_dist_factor, surfel_count):
        dp, sb = self._common()
        v = self.d.keyframe_view(kf_index)
        cnt = C.c_uint32(surfel_count)
        badslam_amd.check(self.L.bslam_determine_supporting_surfels_and_merge(
            self.ctx.handle, stream_ptr(), merge_dist_factor, C.byref(self.h.depth_camera), C.byref(dp), C.byref(v), self.d.surfels_size, C.byref(sb),
            C.byref(cnt)))
        return cnt.value

    def delete_surfels_and_update_radii(self, min_observation_count, surfel_count):
        dp, sb = self._common()
        kfs = self.d.keyframe_views()
        cnt = C.c_uint32(surfel_count)
        badslam_amd.check(self.L.bslam_delete_surfels_and_update_radii(
            self.ctx.handle, stream_ptr(), min_observation_count, C.byref(self.h.depth_camera), C.byref(dp), len(self.h.keyframes), kfs,
            C.byref(cnt), self.d.surfels_size, C.byref(sb)))
        return cnt.value

    def compact_surfels(self, surfel_count, with_active=True):
        sb, ab = self.d.surfel_buf(), self.d.active_buf()
        size = C.c_uint32(self.d.surfels_size)
        badslam_amd.check(self.L.bslam_compact_surfels(self.ctx.handle, stream_ptr(), surfel_count, C.byref(size), C.byref(sb),
                                                       C.byref(ab) if with_active else None))
        self.torch.cuda.synchronize()
        self.d.surfels_size = size.value


class HipPCG:
    """PCG vectors in HBM + the HIP PCG entry points."""
    NAMES = ("r", "M", "delta", "g", "p")

    def __init__(self, hip, layout):
        import torch
        self.hip, self.layout = hip, layout
        n = max(1, layout.unknown_count)
        dev = hip.d.device
        # garbage-fill: the entry points must not rely on zero-initialised vectors
        for nm in self.NAMES:
            setattr(self, nm, torch.full((n,), 123.0, dtype=torch.float32, device=dev))
        self.scalars = torch.full((3,), 7.0, dtype=torch.float32, device=dev)
        self.an, self.bn = 0, 2

    def vectors(self):
        v = abi.PCGVectors()
        for nm in self.NAMES:
            setattr(v, nm, getattr(self, nm).data_ptr())
        base = self.scalars.data_ptr()
        v.alpha_n, v.alpha_d, v.beta_n = base + 4 * self.an, base + 4, base + 4 * self.bn
        return v

    def swap_alpha_beta(self):
        self.an, self.bn = self.bn, self.an

    def get(self, name):
        return getattr(self, name).cpu().numpy()

    def load_from(self, host_pcg):
        """Copies the oracle-side vectors and scalars into HBM (so one kernel is tested at a time)."""
        torch = self.hip.torch
        for nm in self.NAMES:
            getattr(self, nm).copy_(torch.from_numpy(getattr(host_pcg, nm)))
        self.scalars.copy_(torch.from_numpy(host_pcg.scalars))
        self.an, self.bn = host_pcg.an, host_pcg.bn

    def scalar(self, which):
        s = self.scalars.cpu().numpy()
        return float(s[{"alpha_n": self.an, "alpha_d": 1, "beta_n": self.bn}[which]])

    def _surf_args(self):
        h, d = self.hip.h, self.hip.d
        return h, d.depth_params(), d.surfel_buf(), d.keyframe_views()

    def init(self):
        h, dp, sb, kfs = self._surf_args()
        v = self.vectors()
        badslam_amd.check(self.hip.L.bslam_pcg_init(self.hip.ctx.handle, stream_ptr(), C.byref(self.layout), C.byref(h.color_camera),
                                                    C.byref(h.depth_camera), C.byref(dp), len(h.keyframes), kfs, self.hip.d.surfels_size,
                                                    C.byref(sb), C.byref(v)))

    def init2(self):
        v = self.vectors()
        badslam_amd.check(self.hip.L.bslam_pcg_init2(self.hip.ctx.handle, stream_ptr(), C.byref(self.layout), self.hip.h.a, C.byref(v)))

    def step1(self, clear_g):
        h, dp, sb, kfs = self._surf_args()
        v = self.vectors()
        badslam_amd.check(self.hip.L.bslam_pcg_step1(self.hip.ctx.handle, stream_ptr(), C.byref(self.layout), C.byref(h.color_camera),
                                                     C.byref(h.depth_camera), C.byref(dp), len(h.keyframes), kfs, self.hip.d.surfels_size,
                                                     C.byref(sb), C.byref(v), int(clear_g)))

    def step2(self):
        v = self.vectors()
        out = C.c_float()
        badslam_amd.check(self.hip.L.bslam_pcg_step2(self.hip.ctx.handle, stream_ptr(), C.byref(self.layout), C.byref(v), C.byref(out)))
        return out.value

    def step3(self):
        v = self.vectors()
        badslam_amd.check(self.hip.L.bslam_pcg_step3(self.hip.ctx.handle, stream_ptr(), C.byref(self.layout), C.byref(v)))

    def apply_delta_to_surfels(self):
        sb = self.hip.d.surfel_buf()
        badslam_amd.check(self.hip.L.bslam_update_surfels_from_pcg_delta(
            self.hip.ctx.handle, stream_ptr(), self.hip.d.surfels_size, C.byref(sb), int(self.hip.h.use_descriptor_residuals),
            self.layout.surfel_unknown_start_index, C.c_void_p(self.delta.data_ptr())))
        self.hip.torch.cuda.synchronize()

"""Oracle-driven restatements of the reference's BA host loops (test infrastructure):
BS/direct_ba_alternating.cc:345-717 (geometry-only use) and BS/direct_ba_pcg.cc:173-471."""
import numpy as np

from tests import bso


def pcg_ba_iteration(scene, optimize_poses, optimize_geometry, max_inner_iterations=30, gauge=0):
    """One outer iteration of BundleAdjustmentPCG with the oracle's kernels. Returns inner steps done."""
    scene.active[0, :scene.surfels_size] = 1                      # :209
    if optimize_geometry:
        scene.update_normals()                                    # :218
    layout = bso.pcg_layout(scene, optimize_poses=optimize_poses, optimize_geometry=optimize_geometry, gauge_keyframe_id=gauge)
    pcg = bso.HostPCG(scene, layout)
    pcg.init()
    pcg.init2()
    prev, bad, steps = np.inf, 0, 0
    for step in range(max_inner_iterations):
        if step > 0:
            pcg.swap_alpha_beta()
        pcg.step1(step > 0)
        r_norm = float(np.sqrt(np.float32(pcg.step2())))
        steps += 1
        if r_norm < prev - 1e-3:
            bad = 0
        else:
            bad += 1
            if bad >= 3:
                break
        prev = r_norm
        if step < max_inner_iterations - 1:
            pcg.step3()
    if optimize_poses:
        K = len(scene.keyframes)
        for kf in scene.keyframes:
            if kf.id == gauge:
                continue
            idx = 6 * (kf.id if kf.id < gauge else kf.id - 1)
            kf.global_T_frame = bso.se3_mul(kf.global_T_frame, bso.se3_exp(pcg.delta[idx:idx + 6]))
    if optimize_geometry:
        pcg.apply_delta_to_surfels()
    return steps


class OracleAlternatingBA:
    """Host state of DirectBA::BundleAdjustmentAlternating for the !optimize_poses use of the reference's intrinsics /
    depth-deformation tests (BS/direct_ba_alternating.cc:285-738, BS/direct_ba.cc:340-405,566-653), every kernel being the
    oracle's.  With optimize_poses = false the loop ends after min_iterations iterations (:693-700) and no keyframe ever
    leaves the kActive state, so one call is `min_iterations` iteration bodies."""

    def __init__(self, scene, merge_dist_factor=0.8, min_observation_count=2, covisibility=None):
        self.scene = scene
        self.covisibility = covisibility           # {keyframe id: [ids]} or None = every other keyframe
        self.merge_dist_factor = merge_dist_factor
        self.min_obs = min_observation_count       # the tests pass 2 for all three bootstrapping levels
        self.ba_iteration_count = 0
        self.last_ba_iteration_count = -1
        self.last_active_in_ba_iteration = {kf.id: -1 for kf in scene.keyframes}
        self.surfel_count = scene.surfels_size

    def covis(self, kf):
        if self.covisibility is not None:
            return [self.scene.keyframes[j] for j in self.covisibility[kf.id]]
        return [o for o in self.scene.keyframes if o is not kf]

    def scheme_end_tasks(self, do_surfel_updates):                      # BS/direct_ba.cc:566-653
        s = self.scene
        if do_surfel_updates:
            for kf in s.keyframes:
                if self.last_active_in_ba_iteration[kf.id] == self.ba_iteration_count:
                    self.surfel_count = s.merge_surfels(kf, self.merge_dist_factor, self.surfel_count)
        self.surfel_count = s.delete_surfels_and_update_radii(self.min_obs, self.surfel_count)
        s.compact_surfels(self.surfel_count, with_active=False)

    def bundle_adjustment(self, optimize_depth_intrinsics, optimize_color_intrinsics, do_surfel_updates, optimize_geometry,
                          min_iterations, increase_ba_iteration_count):
        s = self.scene
        fixed = self.ba_iteration_count
        if not increase_ba_iteration_count and fixed != self.last_ba_iteration_count:      # :313-319
            self.last_ba_iteration_count = fixed
            self.scheme_end_tasks(do_surfel_updates)
        s.active[0, :s.surfels_size] = 0                                                   # :339
        for _ in range(min_iterations):
            new_kfs = []
            old_size = s.surfels_size
            if optimize_geometry and do_surfel_updates:                                    # :396-427
                for kf in s.keyframes:
                    if self.last_active_in_ba_iteration[kf.id] != fixed:
                        self.last_active_in_ba_iteration[kf.id] = fixed
                        new_kfs.append(kf)
                for kf in new_kfs:
                    self.surfel_count += s.create_surfels_for_keyframe_ex(kf, True, self.min_obs, self.covis(kf))
            if optimize_geometry and s.surfels_size > old_size:                            # :434-440
                s.active[0, old_size:s.surfels_size] = 1
            n_all = s.surfels_size                                                         # :446-454: activation of the old surfels only
            s.surfels_size = old_size
            if old_size > 0:
                s.update_activation()
            s.surfels_size = n_all
            if optimize_geometry:
                s.optimize_geometry_iteration()
            if do_surfel_updates:                                                          # :489-533
                for kf in new_kfs:
                    self.surfel_count = s.merge_surfels(kf, self.merge_dist_factor, self.surfel_count)
                if new_kfs:
                    s.compact_surfels(self.surfel_count, with_active=True)
            if optimize_depth_intrinsics or optimize_color_intrinsics:
                if s.surfels_size > 0:
                    s.optimize_intrinsics(optimize_depth_intrinsics, optimize_color_intrinsics)
        if increase_ba_iteration_count:                                                    # :723-733
            self.scheme_end_tasks(do_surfel_updates)
            self.ba_iteration_count += 1

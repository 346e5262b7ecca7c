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

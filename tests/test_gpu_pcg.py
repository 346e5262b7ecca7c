"""-m gpu parity of the PCG entry points (BS/kernel_pcg.cu) against the oracle."""
import numpy as np
import pytest

from tests import bso, scenes

pytestmark = pytest.mark.gpu


def close(a, b, rel, what):
    a, b = np.atleast_1d(np.asarray(a, np.float64)), np.atleast_1d(np.asarray(b, np.float64))
    if b.size == 0:
        return 0.0
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rel, f"{what}: max rel err {err:.3e} > {rel}"
    return err


def perturbed_scene(use_desc, seed, K=4):
    scene = scenes.synthetic_scene(K, seed=seed, use_depth_residuals=True, use_descriptor_residuals=use_desc)
    rng = np.random.default_rng(seed)
    n = scene.surfels_size
    scene.surfels[2, :n] += rng.uniform(-0.003, 0.003, n).astype(np.float32)
    for kf in scene.keyframes[1:]:
        x = np.concatenate([rng.uniform(-0.002, 0.002, 3), rng.uniform(-0.0005, 0.0005, 3)]).astype(np.float32)
        kf.global_T_frame = bso.se3_mul(kf.global_T_frame, bso.se3_exp(x))
    return scene


@pytest.mark.parametrize("use_desc,opt_poses,opt_geom", [(False, True, True), (True, True, True), (True, False, True), (False, True, False)])
def test_pcg_steps_match_oracle(oracle, use_desc, opt_poses, opt_geom):
    from tests import gpu_util
    scene = perturbed_scene(use_desc, seed=31 + int(use_desc))
    hip = gpu_util.Hip(scene.to_device())
    layout = bso.pcg_layout(scene, optimize_poses=opt_poses, optimize_geometry=opt_geom, gauge_keyframe_id=1)
    ref = bso.HostPCG(scene, layout)
    got = gpu_util.HipPCG(hip, layout)
    n = layout.unknown_count
    npose = 6 * (len(scene.keyframes) - 1) if opt_poses else 0

    ref.init(); got.init()
    # per-surfel entries are formed in the reference's order: expect (near) identical bits
    close(got.get("r")[npose:n], ref.r[npose:n], 1e-6, "r0 surfel entries")
    close(got.get("M")[npose:n], ref.M[npose:n], 1e-6, "M surfel entries")
    if npose:
        close(got.get("r")[:npose], ref.r[:npose], 1e-4, "r0 pose entries")
        close(got.get("M")[:npose], ref.M[:npose], 1e-4, "M pose entries")
    ref.init2(); got.init2()
    close(got.get("p")[:n], ref.p[:n], 1e-4, "p0")
    close(got.scalar("alpha_n"), ref.scalars[ref.an], 1e-4, "alpha_n")
    assert not got.get("delta")[:n].any() and not got.get("g")[:n].any()

    # each kernel on identical inputs: the HBM vectors are reloaded from the oracle before every step,
    # so CG's amplification of rounding differences does not mask (or fake) a kernel difference
    for step in range(3):
        if step > 0:
            ref.swap_alpha_beta()
        got.load_from(ref)
        ref.step1(step > 0); got.step1(step > 0)
        close(got.scalar("alpha_d"), bso.lib().bso_pcg_last_alpha_d64(), 1e-4, f"alpha_d step {step}")
        close(got.get("g")[npose:n], ref.g[npose:n], 1e-5, f"g surfel entries step {step}")
        close(got.get("g")[:n], ref.g[:n], 1e-4, f"g step {step}")
        got.load_from(ref)
        b_ref = ref.step2(); b_got = got.step2()
        close(b_got, b_ref, 1e-4, f"beta_n step {step}")
        close(got.get("delta")[:n], ref.delta[:n], 1e-5, f"delta step {step}")
        close(got.get("r")[:n], ref.r[:n], 1e-5, f"r step {step}")
        close(got.get("g")[:n], ref.g[:n], 1e-5, f"z step {step}")
        got.load_from(ref)
        ref.step3(); got.step3()
        close(got.get("p")[:n], ref.p[:n], 1e-5, f"p step {step}")
    got.load_from(ref)

    if opt_geom:
        ref.apply_delta_to_surfels(); got.apply_delta_to_surfels()
        g = hip.d.surfels_np()[:8, :scene.surfels_size]
        r = scene.surfels[:8, :scene.surfels_size]
        for row in (0, 1, 2, 6, 7):
            close(g[row], r[row], 1e-4, f"surfel row {row} after delta")

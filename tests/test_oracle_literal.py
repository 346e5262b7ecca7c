"""How far has the oracle's "twin" evaluation shape (the one the HIP kernels share bit for bit: explicit fmaf chains, one
rounded reciprocal per projection, `* fl(1 / baseline_fx)`, Cephes expf) moved from a LITERAL transcription of the reference's
source expressions (oracle/bso_math.h, bso_literal_mode)?

The reference is built with -use_fast_math, so neither shape is "the" reference to the last bit; what must hold is that
the two shapes decide every surfel identically except on exact threshold ties, and that their float outputs differ by rounding
only.  These tests measure both on the 640x480 scenes and list the ties with their margins."""
import numpy as np
import pytest

from tests import bso, scenes


def association_both_modes(scene, kf):
    twin = scene.association(kf)
    with bso.literal_mode():
        lit = scene.association(kf)
    return twin, lit


def min_abs_margin(m):
    """Smallest |margin| of the evaluated stages in tie units (NaN = stage not reached).  Margins 2..4 are relative / cosines:
    tie below 1e-5.  Margin 1 is the distance of the projected point to a pixel boundary in pixels, on coordinates up to 640
    whose fp32 spacing is 6.1e-5 px: tie below 1e-5 x 640 px (|margin| < 1e-5 x scale); it is divided by 640 here."""
    mm = np.abs(m[:, 1:]).copy()
    mm[:, 0] /= 640.0
    mm[np.isnan(mm)] = np.inf
    return mm.min(axis=1)


@pytest.mark.parametrize("seed", [0xBAD51A4, 11])
def test_association_literal_vs_twin_differs_only_on_ties(oracle, seed, capsys):
    """Integer outputs (associated pixel or none) of the two shapes on a synthetic 640x480 stack (4 keyframes, cell 2:
    ~300 k surfels x 4 keyframes): every disagreement must sit on a decision threshold -- |margin| < 1e-5 (pixels for the
    pixel-boundary test, relative for the depth / angle tests), the survey's tie definition -- and there must be few of them."""
    scene = scenes.synthetic_scene(4, seed=seed, cell=2, use_depth_residuals=True, use_descriptor_residuals=False)
    pairs = mismatches = 0
    worst = 0.0
    lines = []
    for kf in scene.keyframes:
        twin, lit = association_both_modes(scene, kf)
        bad = np.nonzero(twin != lit)[0]
        pairs += twin.size
        mismatches += bad.size
        if bad.size:
            marg = scene.association_margins(kf)[bad]
            mam = min_abs_margin(marg)
            worst = max(worst, float(mam.max()))
            for i, b in enumerate(bad[:20]):
                lines.append(f"kf {kf.id} surfel {b}: twin {int(twin[b]) if twin[b] != 0xffffffff else None} literal "
                             f"{int(lit[b]) if lit[b] != 0xffffffff else None} margins {marg[i].tolist()}")
    with capsys.disabled():
        print(f"\n[literal vs twin] seed {seed:#x}: {mismatches} of {pairs} pairs differ (worst |margin| {worst:.2e})")
        for ln in lines:
            print("   ", ln)
    assert mismatches <= 1e-4 * pairs, (mismatches, pairs)   # measured: 9 / 474 848, 7 / 536 492, 23 / 818 520
    assert worst < 1e-5, worst


def test_association_literal_vs_twin_known_answer_scene(oracle):
    """Same on the reference's own test scene (BS/test/test_pose_optimization_geometric_residual.cc:50-131, ~165 k surfels,
    cell 1) with the keyframe pose moved by the test's 5 mm / 1 mrad offsets."""
    scene, kf = scenes.pose_geometric_scene(seed=0)
    pairs = mismatches = 0
    for off in scenes.offsets_13(0.005, 0.001)[:5]:
        kf.global_T_frame = bso.se3_mul(off, bso.se3_identity())
        twin, lit = association_both_modes(scene, kf)
        bad = np.nonzero(twin != lit)[0]
        pairs += twin.size
        mismatches += bad.size
        if bad.size:
            assert min_abs_margin(scene.association_margins(kf)[bad]).max() < 1e-5
    assert mismatches <= 1e-4 * pairs, (mismatches, pairs)   # measured: 9 / 474 848, 7 / 536 492, 23 / 818 520


@pytest.mark.parametrize("photometric", [False, True])
def test_residuals_and_coefficients_literal_vs_twin(oracle, photometric):
    """Float outputs: per-surfel raw residuals / weights and the accumulated H, b of the two shapes agree to rounding
    (residuals 1e-4 absolute on O(1) values for surfels associated in both, sums 1e-5 relative)."""
    scene = scenes.synthetic_scene(3, seed=5, cell=4, use_depth_residuals=True, use_descriptor_residuals=photometric)
    for kf in scene.keyframes:
        # perturb the pose so that residuals are not ~0
        kf.global_T_frame = bso.se3_mul(kf.global_T_frame, bso.se3_exp(np.array([0.004, -0.003, 0.002, 0.001, -0.0008, 0.0006], np.float32)))
    for kf in scene.keyframes:
        twin = scene.accumulate_pose(kf, per_surfel=True)
        with bso.literal_mode():
            lit = scene.accumulate_pose(kf, per_surfel=True)
        both = (twin["per_surfel"][:, 6] == lit["per_surfel"][:, 6]) & (twin["per_surfel"][:, 6] > 0)
        assert both.sum() > 1000
        d = np.abs(twin["per_surfel"][both, :6] - lit["per_surfel"][both, :6])
        # depth residual in sigmas (range +-10, the Tukey parameter) = inv_stddev x a difference of nearly equal metres: one
        # ulp of the local position (2.4e-7 m at 2.5 m) times inv_stddev (~1e2 / m head-on, ~1e4 / m for a surfel seen at a
        # grazing angle, where sigma ~ |n . ray| d^2 shrinks) is 2e-5 ... 2e-3 sigma: the two shapes agree to a few ulps of
        # their INPUTS.  Measured: median 2e-6, 99.9 % below 2.5e-5, worst 3.7e-3 (a grazing surfel).
        assert np.percentile(d[:, 0], 99.9) < 1e-4 and d[:, 0].max() < 1e-2, (np.percentile(d[:, 0], 99.9), d[:, 0].max())
        assert d[:, 1].max() < 1e-4, d[:, 1].max()                     # Tukey weight
        if photometric:
            # descriptor residuals are 180 x an 8-bit intensity difference sampled with 1/256-quantised weights: a last-bit
            # move of a sample point can flip one quantised weight (2^-8 x local contrast x 180)
            # (measured: median 0, 99 % below 1e-3, 99.9 % below 8.3e-3, worst 1.4e-2 on residuals of O(1 ... 100))
            assert np.percentile(d[:, [2, 4]], 99) < 2e-3 and np.percentile(d[:, [2, 4]], 99.9) < 2e-2
            assert d[:, [2, 4]].max() < 0.1
        scale = np.abs(twin["H64"]).max()
        assert np.abs(twin["H64"] - lit["H64"]).max() <= (2e-4 if photometric else 1e-5) * scale
        assert abs(int(twin["count"]) - int(lit["count"])) <= 2

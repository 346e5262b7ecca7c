"""CPU: the oracle's AssignColorsCUDA restatement (oracle/bslam_oracle.c: bso_assign_colors) on cases with a closed-form
answer.  The reference holds no test for this function (it only feeds the visualisation and the point-cloud export)."""
import numpy as np

from tests import scenes


def test_constant_keyframe_colors_are_reproduced_exactly(oracle):
    scene = scenes.synthetic_scene(3, seed=4, width=320, height=240, use_depth_residuals=True, use_descriptor_residuals=True)
    n = scene.surfels_size
    for kf in scene.keyframes:
        kf.color[..., 0], kf.color[..., 1], kf.color[..., 2], kf.color[..., 3] = 200, 17, 96, 255
    scene.surfels[0:3, n - 20:n] += 100.0          # nobody observes these
    sentinel = np.uint32(0x01020304)
    scene.surfels[5, :n] = np.full(n, sentinel, np.uint32).view(np.float32)
    scene.assign_colors()
    got = scene.surfels[5, :n].view(np.uint32)
    want = np.uint32(200 | (17 << 8) | (96 << 16) | (255 << 24))
    assert (got[n - 20:] == sentinel).all()
    seen = got != sentinel
    assert seen.sum() > 0.9 * (n - 20)
    # the bilinear blend of a constant image is the constant (weights sum to 1 within an ulp); rounding 255 * mean + 0.5 restores it
    assert (got[seen] == want).all()


def test_mean_over_keyframes(oracle):
    scene = scenes.synthetic_scene(2, seed=4, width=320, height=240, use_depth_residuals=True, use_descriptor_residuals=True)
    n = scene.surfels_size
    scene.keyframes[0].color[...] = 100
    scene.keyframes[1].color[...] = 50
    scene.surfels[5, :n] = np.zeros(n, np.float32)
    scene.assign_colors()
    r = scene.surfels[5, :n].view(np.uint32) & 0xff
    # surfels seen by both keyframes get the mean (75), the others the colour of the one keyframe that sees them
    values, counts = np.unique(r, return_counts=True)
    assert set(values.tolist()) <= {0, 50, 75, 100}
    assert dict(zip(values.tolist(), counts.tolist())).get(75, 0) > 0.2 * n

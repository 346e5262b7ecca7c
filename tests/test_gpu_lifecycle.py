"""-m gpu parity of the surfel lifecycle entry points against the oracle: creation with and
without the observation-count filter, supporting surfels + merge, deletion + radius update, compaction.
All decisions are integer / bit-pattern outputs (which pixel creates a surfel, which surfel is merged away or
deleted, where compaction moves it): they must match exactly."""
import numpy as np
import pytest

from badslam_amd import abi
from tests import bso, scenes

pytestmark = pytest.mark.gpu

NAN_BITS = 0x7FFFFFFF


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def build_scene(K=4, seed=5, use_desc=True):
    """K keyframes preprocessed by the oracle; surfels created (unfiltered) from the first K - 1 only."""
    full = scenes.synthetic_scene(K, seed=seed, cell=4, use_depth_residuals=True, use_descriptor_residuals=use_desc,
                                  max_surfels=19200 * K * 2)
    scene = bso.HostScene(full.color_camera, full.depth_camera, full.raw_to_float_depth, full.baseline_fx, full.cell, full.max_surfels,
                          use_depth_residuals=True, use_descriptor_residuals=use_desc, tex_mode=full.tex_mode)
    scene.keyframes = full.keyframes
    for kf in scene.keyframes[:K - 1]:
        scene.create_surfels_for_keyframe(kf)
    return scene


def rows_equal(got, ref, n, what):
    for row in range(8):
        same = np.array_equal(bits(got[row, :n]), bits(ref[row, :n]))
        assert same, f"{what}: surfel row {row} differs in {(bits(got[row, :n]) != bits(ref[row, :n])).sum()} of {n} columns"


@pytest.mark.parametrize("filtered", [False, True])
def test_create_surfels_matches_oracle(oracle, filtered):
    from tests import gpu_util
    scene = build_scene()
    K = len(scene.keyframes)
    hip = gpu_util.Hip(scene.to_device())
    before = scene.surfels_size
    covis = list(range(K - 1))
    created_ref = scene.create_surfels_for_keyframe_ex(scene.keyframes[K - 1], filtered, 2, [scene.keyframes[i] for i in covis])
    created = hip.create_surfels_for_keyframe(K - 1, filtered, 2, covis)
    assert created == created_ref and created > 1000
    if filtered:
        unfiltered = build_scene()
        assert created < unfiltered.create_surfels_for_keyframe_ex(unfiltered.keyframes[K - 1], False, 2, [])
    got = hip.d.surfels_np()
    # same pixels chosen, same positions / normals / radii / colours / descriptors: identical arithmetic, identical bits
    rows_equal(got[:, before:before + created], scene.surfels[:, before:before + created], created, "created surfels")
    rows_equal(got, scene.surfels, before, "existing surfels untouched")


def overlapping_scene():
    """Surfels created from every keyframe independently (no occupancy test) -> plenty of duplicates to merge."""
    full = scenes.synthetic_scene(3, seed=9, cell=4, use_depth_residuals=True, use_descriptor_residuals=False, max_surfels=19200 * 8)
    scene = bso.HostScene(full.color_camera, full.depth_camera, full.raw_to_float_depth, full.baseline_fx, full.cell, full.max_surfels,
                          use_depth_residuals=True, use_descriptor_residuals=False, tex_mode=full.tex_mode)
    scene.keyframes = full.keyframes
    for kf in scene.keyframes:
        single = bso.HostScene(full.color_camera, full.depth_camera, full.raw_to_float_depth, full.baseline_fx, full.cell, 19200,
                               use_depth_residuals=True, use_descriptor_residuals=False, tex_mode=full.tex_mode)
        single.keyframes = [kf]
        single.create_surfels_for_keyframe(kf)
        n = single.surfels_size
        scene.surfels[:, scene.surfels_size:scene.surfels_size + n] = single.surfels[:, :n]
        scene.surfels_size += n
    scene.active[0, :scene.surfels_size] = np.arange(scene.surfels_size, dtype=np.uint32).astype(np.uint8) | 1   # distinguishable flags
    return scene


def test_merge_delete_compact_match_oracle(oracle):
    from tests import gpu_util
    scene = overlapping_scene()
    hip = gpu_util.Hip(scene.to_device())
    n = scene.surfels_size
    count_ref = count = n
    # merge against every keyframe in turn without compaction in between (BS/direct_ba.cc:578-597)
    for k, kf in enumerate(scene.keyframes):
        count_ref = scene.merge_surfels(kf, 0.8, count_ref)
        count = hip.merge_surfels(k, 0.8, count)
        assert count == count_ref, (k, count, count_ref)
        got = hip.d.surfels_np()
        assert np.array_equal(bits(got[0, :n]) == NAN_BITS, bits(scene.surfels[0, :n]) == NAN_BITS), f"merged set differs at keyframe {k}"
    assert count < 0.8 * n, (count, n)   # the duplicates really are merged
    # deletion + radius update (BS/direct_ba.cc:616)
    count_ref = scene.delete_surfels_and_update_radii(2, count_ref)
    count = hip.delete_surfels_and_update_radii(2, count)
    assert count == count_ref
    got = hip.d.surfels_np()
    rows_equal(got, scene.surfels, n, "after merge + delete")
    assert 0 < count < n
    # compaction (BS/direct_ba.cc:620)
    scene.compact_surfels(count_ref)
    hip.compact_surfels(count)
    assert hip.d.surfels_size == scene.surfels_size == count
    got = hip.d.surfels_np()
    rows_equal(got, scene.surfels, count, "after compaction")
    assert not (bits(got[0, :count]) == NAN_BITS).any()
    assert np.array_equal(hip.d.active_np()[0, :count], scene.active[0, :count])


def test_compaction_without_active_buffer_and_noop(oracle):
    from tests import gpu_util
    scene = build_scene(K=3, use_desc=False)
    hip = gpu_util.Hip(scene.to_device())
    n = scene.surfels_size
    hip.compact_surfels(n)                      # nothing deleted: no-op
    assert hip.d.surfels_size == n
    rng = np.random.default_rng(0)
    kill = rng.choice(n, n // 3, replace=False)
    scene.surfels[0, kill] = np.array([NAN_BITS], np.uint32).view(np.float32)[0]
    hip.d.surfels.copy_(hip.torch.from_numpy(scene.surfels))
    scene.compact_surfels(n - kill.size, with_active=False)
    hip.compact_surfels(n - kill.size, with_active=False)
    rows_equal(hip.d.surfels_np(), scene.surfels, n - kill.size, "compaction of a random third")

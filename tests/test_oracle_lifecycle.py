"""CPU: invariants of the oracle's surfel lifecycle restatement (oracle/bso_lifecycle.c), the checker of
tests/test_gpu_lifecycle.py.  The reference has no unit test of these kernels; what can be pinned on the CPU are
the properties their definitions imply."""
import numpy as np

from tests import bso, scenes

NAN_BITS = 0x7FFFFFFF


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def small_scene(K=3, seed=3):
    cam = bso.make_camera(262.5, 262.5, 160.0, 120.0, 320, 240)
    full = scenes.synthetic_scene(K, seed=seed, width=320, height=240, cell=4, camera=cam, max_surfels=4800 * K * 3)
    return full


def test_creation_fills_every_free_cell_once(oracle):
    scene = small_scene()
    kf = scene.keyframes[0]
    # a second creation pass on the same keyframe finds (almost) every cell occupied by its own surfels
    before = scene.surfels_size
    again = scene.create_surfels_for_keyframe_ex(kf, False, 1, [])
    assert again < 0.02 * before
    # filtered creation never creates more than unfiltered, and needs >= 2 observers when asked to
    a, b = small_scene(), small_scene()
    a.surfels_size = b.surfels_size = 0
    n_unf = a.create_surfels_for_keyframe_ex(a.keyframes[0], False, 2, a.keyframes[1:])
    n_fil = b.create_surfels_for_keyframe_ex(b.keyframes[0], True, 2, b.keyframes[1:])
    assert 0 < n_fil < n_unf
    n_alone = small_scene()
    n_alone.surfels_size = 0
    assert n_alone.create_surfels_for_keyframe_ex(n_alone.keyframes[0], True, 2, []) == 0   # nobody else observes


def test_merge_delete_compact_invariants(oracle):
    scene = small_scene()
    # duplicate every surfel: the copies must be merged away by the keyframe that sees them
    n0 = scene.surfels_size
    scene.surfels[:, n0:2 * n0] = scene.surfels[:, :n0]
    scene.surfels_size = 2 * n0
    scene.active[0, :2 * n0] = 1
    count = scene.surfels_size
    for kf in scene.keyframes:
        count = scene.merge_surfels(kf, 0.8, count)
    deleted = bits(scene.surfels[0, :scene.surfels_size]) == NAN_BITS
    assert deleted.sum() == scene.surfels_size - count
    assert count < 1.3 * n0                                   # most copies are gone ...
    assert deleted[n0:].sum() > 0.7 * n0                      # ... and it is the later (higher-index) copy that goes
    count2 = scene.delete_surfels_and_update_radii(1, count)
    assert count2 <= count
    valid = bits(scene.surfels[0, :scene.surfels_size]) != NAN_BITS
    assert valid.sum() == count2
    survivors = scene.surfels[:8, :scene.surfels_size][:, valid].copy()
    scene.compact_surfels(count2)
    assert scene.surfels_size == count2
    packed = scene.surfels[:8, :count2]
    assert not (bits(packed[0]) == NAN_BITS).any()
    # compaction permutes, it never changes a surfel: same multiset of columns
    key = lambda m: np.sort(bits(m).astype(np.uint64).T.dot(np.arange(1, 9, dtype=np.uint64) * 0x9E3779B97F4A7C15 % (1 << 61)))
    assert np.array_equal(key(packed), key(survivors))
    # and only the tail moved: surviving columns in front of the first hole keep their place
    first_hole = int(np.argmin(valid)) if not valid.all() else count2
    assert np.array_equal(bits(packed[:, :first_hole]), bits(survivors[:, :first_hole]))

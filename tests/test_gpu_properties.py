"""-m gpu: size-independent properties of the hot path at the full bench size (50 keyframes x 960 000 surfels, the
BASELINE.json configs[1] shape), where the oracle would take minutes: linearity over surfel shards, permutation
invariance, idempotence, and a delete/compact round trip.  No oracle involved."""
import ctypes as C

import numpy as np
import pytest

import badslam_amd
from badslam_amd import abi, synthetic

pytestmark = pytest.mark.gpu
P = C.POINTER
K = 50


@pytest.fixture(scope="module")
def stack():
    return synthetic.SyntheticStack(K, seed=0xBAD51A4)


class Run:
    def __init__(self, stack, surfel_range=None, order=None):
        import torch
        self.torch = torch
        self.stack = stack
        self.dev = synthetic.DeviceStack(stack, "cuda:0", surfel_range)
        if order is not None:
            self.dev.surfels = self.dev.surfels[:, torch.from_numpy(order).cuda()].contiguous()
        self.L = badslam_amd.lib()
        self.ctx = badslam_amd.Context(0)
        self.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.S = self.dev.surfels_size

    def coeffs(self):
        dp, sb, kfs = self.dev.depth_params(), self.dev.buf(self.dev.surfels), self.dev.keyframe_views()
        Hb = np.zeros((K, 27), np.float32)
        counts = np.zeros(K, np.uint32)
        cam = self.stack.camera
        badslam_amd.check(self.L.bslam_accumulate_pose_coeffs_batched(
            self.ctx.handle, self.stream, 1, 0, C.byref(cam), C.byref(cam), C.byref(dp), K, kfs, self.S, C.byref(sb),
            Hb.ctypes.data_as(P(C.c_float)), counts.ctypes.data_as(P(C.c_uint32))))
        return Hb.astype(np.float64), counts.astype(np.uint64)

    def activation(self, kfs=None):
        dp, sb, ab = self.dev.depth_params(), self.dev.buf(self.dev.surfels), self.dev.buf(self.dev.active)
        kfs = self.dev.keyframe_views() if kfs is None else kfs
        cam = self.stack.camera
        badslam_amd.check(self.L.bslam_update_surfel_activation(self.ctx.handle, self.stream, C.byref(cam), C.byref(dp), K, kfs, self.S, C.byref(sb), C.byref(ab)))
        self.torch.cuda.synchronize()
        return self.dev.active.cpu().numpy()[0, :self.S].copy()

    def geometry(self, kfs=None):
        dp, sb, ab = self.dev.depth_params(), self.dev.buf(self.dev.surfels), self.dev.buf(self.dev.active)
        kfs = self.dev.keyframe_views() if kfs is None else kfs
        cam = self.stack.camera
        badslam_amd.check(self.L.bslam_optimize_geometry_iteration(self.ctx.handle, self.stream, 1, 0, C.byref(cam), C.byref(cam), C.byref(dp), K, kfs,
                                                                   self.S, C.byref(sb), C.byref(ab)))
        self.torch.cuda.synchronize()
        return self.dev.surfels.cpu().numpy()


def test_coefficients_are_additive_over_surfel_shards(stack):
    S = stack.surfels_size
    assert S == 960000
    full_Hb, full_cnt = Run(stack).coeffs()
    a_Hb, a_cnt = Run(stack, (0, S // 3)).coeffs()
    b_Hb, b_cnt = Run(stack, (S // 3, S)).coeffs()
    assert np.array_equal(a_cnt + b_cnt, full_cnt) and full_cnt.min() > 100000      # integer counts: exact
    scale = np.abs(full_Hb).max(axis=1, keepdims=True)
    assert (np.abs(a_Hb + b_Hb - full_Hb) <= 2e-5 * scale).all()                     # fp32 sums in another order


def test_coefficients_do_not_depend_on_the_surfel_order(stack):
    S = stack.surfels_size
    full_Hb, full_cnt = Run(stack).coeffs()
    order = np.random.default_rng(0).permutation(S)
    p_Hb, p_cnt = Run(stack, order=order).coeffs()
    assert np.array_equal(p_cnt, full_cnt)
    scale = np.abs(full_Hb).max(axis=1, keepdims=True)
    assert (np.abs(p_Hb - full_Hb) <= 2e-5 * scale).all()


def test_activation_is_idempotent_and_geometry_respects_inactive_keyframes(stack):
    r = Run(stack)
    first = r.activation()
    assert first.all()                                    # every surfel is seen by an active keyframe here
    assert np.array_equal(r.activation(), first)
    inactive = r.dev.keyframe_views(activation=abi.KF_INACTIVE)
    assert not (r.activation(inactive) & 1).any()         # no active keyframe -> no active surfel
    r.dev.active.fill_(1)
    before = r.dev.surfels.cpu().numpy().copy()
    after = r.geometry(inactive)                          # all keyframes inactive: nothing is observed, nothing moves
    assert np.array_equal(after.view(np.uint32), before.view(np.uint32))
    moved = r.geometry()                                  # and with active keyframes the step is deterministic
    r2 = Run(stack)
    r2.dev.active.fill_(1)
    assert np.array_equal(r2.geometry().view(np.uint32), moved.view(np.uint32))
    assert not np.array_equal(moved[:3], before[:3])


def test_delete_and_compact_round_trip_at_full_size(stack):
    r = Run(stack)
    S = r.S
    dp, sb, ab, kfs = r.dev.depth_params(), r.dev.buf(r.dev.surfels), r.dev.buf(r.dev.active), r.dev.keyframe_views()
    cam = stack.camera
    count = C.c_uint32(S)
    badslam_amd.check(r.L.bslam_delete_surfels_and_update_radii(r.ctx.handle, r.stream, 30, C.byref(cam), C.byref(dp), K, kfs, C.byref(count), S, C.byref(sb)))
    surf = r.dev.surfels.cpu().numpy()
    deleted = surf[0, :S].view(np.uint32) == 0x7FFFFFFF
    assert deleted.sum() == S - count.value and 0 < count.value < S          # "seen by >= 30 keyframes" splits the set
    survivors = surf[:8, :S][:, ~deleted]
    size = C.c_uint32(S)
    badslam_amd.check(r.L.bslam_compact_surfels(r.ctx.handle, r.stream, count.value, C.byref(size), C.byref(sb), C.byref(ab)))
    r.torch.cuda.synchronize()
    assert size.value == count.value
    packed = r.dev.surfels.cpu().numpy()[:8, :size.value]
    assert not (packed[0].view(np.uint32) == 0x7FFFFFFF).any()
    key = lambda m: np.sort(m.view(np.uint32).astype(np.uint64).T.dot(np.arange(1, 9, dtype=np.uint64)))   # multiset of columns
    assert np.array_equal(key(packed), key(survivors))
    # a second deletion pass with the same bar removes nothing more, and compaction is then a no-op
    count2 = C.c_uint32(size.value)
    badslam_amd.check(r.L.bslam_delete_surfels_and_update_radii(r.ctx.handle, r.stream, 30, C.byref(cam), C.byref(dp), K, kfs, C.byref(count2), size.value, C.byref(sb)))
    assert count2.value == size.value

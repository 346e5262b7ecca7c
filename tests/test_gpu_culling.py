"""-m gpu: block-level frustum culling (csrc/device_math.hpp: box_outside_frustum) must be exactly conservative.

A workgroup skips a keyframe when no point of its surfels' bounding box can project into the image.  If the test ever skipped a
pair that passes the projection test, a residual would go missing; so every output -- integer ones and float sums alike -- has to
be bit-identical with culling on and off.  Checked on the three synthetic stacks of SURVEY.md 8(d) (dense: little to cull;
survey ranges and trajectory: most pairs are out of view), with keyframe images whose border pixels carry measurements (so
that pairs on the very edge of the image do associate) and at poses shifted by fractions of a pixel, which moves thousands of
surfels across the image bounds."""
import ctypes as C

import numpy as np
import pytest

import badslam_amd
from badslam_amd import abi, synthetic

pytestmark = pytest.mark.gpu
P = C.POINTER


class Runner:
    def __init__(self, dev, use_desc):
        import torch
        self.torch, self.dev, self.use_desc = torch, dev, use_desc
        self.L = badslam_amd.lib()
        self.ctx = badslam_amd.Context(0)
        self.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.K = dev.stack.K
        self.cam = dev.stack.camera

    def views(self, xis=None):
        kfs = self.dev.keyframe_views()
        if xis is not None:
            for k in range(self.K):
                _, M, Rg = self.dev.stack.pose(k, xis[k])
                kfs[k].frame_T_global, kfs[k].global_R_frame = M, Rg
        return kfs

    def culling(self, on):
        badslam_amd.check(self.L.bslam_set_culling(self.ctx.handle, int(on)))

    def coeffs(self, kfs):
        dp, sb = self.dev.depth_params(), self.dev.buf(self.dev.surfels)
        Hb = np.zeros((self.K, 27), np.float32)
        counts = np.zeros(self.K, np.uint32)
        badslam_amd.check(self.L.bslam_accumulate_pose_coeffs_batched(
            self.ctx.handle, self.stream, 1, int(self.use_desc), C.byref(self.cam), C.byref(self.cam), C.byref(dp), self.K, kfs, self.dev.surfels_size,
            C.byref(sb), Hb.ctypes.data_as(P(C.c_float)), counts.ctypes.data_as(P(C.c_uint32))))
        return Hb, counts

    def geometry(self, kfs):
        dp, sb, ab = self.dev.depth_params(), self.dev.buf(self.dev.surfels), self.dev.buf(self.dev.active)
        badslam_amd.check(self.L.bslam_optimize_geometry_iteration(self.ctx.handle, self.stream, 1, int(self.use_desc), C.byref(self.cam), C.byref(self.cam),
                                                                   C.byref(dp), self.K, kfs, self.dev.surfels_size, C.byref(sb), C.byref(ab)))
        self.torch.cuda.synchronize()

    def normals(self, kfs):
        dp, sb, ab = self.dev.depth_params(), self.dev.buf(self.dev.surfels), self.dev.buf(self.dev.active)
        badslam_amd.check(self.L.bslam_update_surfel_normals(self.ctx.handle, self.stream, C.byref(self.cam), C.byref(dp), self.K, kfs, self.dev.surfels_size,
                                                             C.byref(sb), C.byref(ab)))
        self.torch.cuda.synchronize()

    def poses(self, kfs, inits, iterations):
        dp, sb = self.dev.depth_params(), self.dev.buf(self.dev.surfels)
        poses = (abi.SE3f * self.K)()
        C.memmove(poses, inits, C.sizeof(poses))
        iters, conv = (C.c_int32 * self.K)(), (C.c_int32 * self.K)()
        badslam_amd.check(self.L.bslam_estimate_frame_poses_batched(self.ctx.handle, self.stream, 1, int(self.use_desc), C.byref(self.cam), C.byref(self.cam),
                                                                    C.byref(dp), self.K, kfs, self.dev.surfels_size, C.byref(sb), iterations, poses, iters, conv,
                                                                    C.cast(None, abi.ALLREDUCE_FN), None))
        return np.array([[*p.q, *p.t] for p in poses], np.float32), list(iters)

    def cull_stats(self):
        t, c = C.c_uint64(), C.c_uint64()
        badslam_amd.check(self.L.bslam_debug_cull_stats(self.ctx.handle, C.byref(t), C.byref(c)))
        return t.value, c.value


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# the pose kernel culls from 64 keyframes on (shorter lists: its blocks are too short to pay for the test), the geometry passes always
@pytest.mark.parametrize("kind,K,use_desc", [("dense", 8, True), ("dense", 64, False), ("survey", 12, False), ("survey", 66, True),
                                             ("trajectory", 40, False), ("trajectory", 72, False), ("trajectory", 72, True)])
def test_outputs_are_bit_identical_with_and_without_culling(kind, K, use_desc):
    import torch
    dev = synthetic.TorchStack(K, "cuda:0", kind=kind, border_valid=True)
    assert dev.surfels_size >= 64 * 256 and K >= 4          # the per-surfel work order (and with it the culling) is in use
    run = Runner(dev, use_desc)
    start = dev.surfels.clone()
    rng = np.random.default_rng(11)
    pix = 1.0 / 525.0                                        # one pixel, in radians
    pose_sets = [None]
    for scale in (0.3, 1.0, 2.5):                           # sub-pixel and few-pixel shifts: surfels cross the image bounds
        pose_sets.append([np.concatenate([rng.uniform(-1, 1, 3) * 0.002 * scale, rng.uniform(-1, 1, 3) * pix * scale]) for _ in range(K)])
    inits = (abi.SE3f * K)()
    for k in range(K):
        inits[k] = dev.stack.pose(k, np.concatenate([rng.choice([-1, 1], 3) * 0.005, rng.choice([-1, 1], 3) * 0.001]))[0]
    out = {}
    for on in (1, 0):
        run.culling(on)
        dev.surfels.copy_(start)
        res = []
        badslam_amd.check(run.L.bslam_profile_enable(run.ctx.handle, 1))
        run.cull_stats()
        for xis in pose_sets:
            Hb, counts = run.coeffs(run.views(xis))
            res += [bits(Hb), counts]
        tested, culled = run.cull_stats()
        badslam_amd.check(run.L.bslam_profile_enable(run.ctx.handle, 0))
        run.normals(run.views(pose_sets[1]))
        res.append(bits(dev.surfels[:8].cpu().numpy()))
        run.geometry(run.views(pose_sets[2]))                # with descriptors: two iterations, so that the second one sees fitted descriptors
        run.geometry(run.views())
        res.append(bits(dev.surfels[:8].cpu().numpy()))
        poses, iters = run.poses(run.views(), inits, 4)
        res += [bits(poses), np.array(iters)]
        out[on] = (res, tested, culled)
    dev.surfels.copy_(start)
    for a, b in zip(out[1][0], out[0][0]):
        assert np.array_equal(a, b)
    assert out[0][2] == 0 and out[0][1] > 0                  # culling off: every (slot, keyframe) pair visited
    assert out[1][1] == out[0][1]
    frac = out[1][2] / out[1][1]
    counts = out[1][0][1]
    assert counts.min() > 1000, counts.min()                 # the sums are not trivially empty
    if K < 64:
        assert frac == 0.0                                   # short keyframe lists: the pose kernel visits everything
    elif kind == "trajectory":
        assert frac > 0.6, frac                              # most of the room is out of view of a keyframe
    elif kind == "survey":
        assert frac > 0.2, frac


def test_deleted_and_far_surfels_do_not_break_the_boxes():
    """NaN positions (deleted surfels) are ignored by the boxes, infinite or huge ones make a box that is never culled."""
    import torch
    K = 16
    dev = synthetic.TorchStack(K, "cuda:0", kind="trajectory", border_valid=True)
    run = Runner(dev, False)
    n = dev.surfels_size
    dev.surfels[0, torch.arange(0, n, 13, device="cuda:0")] = float("nan")
    dev.surfels[1, torch.arange(5, n, 4099, device="cuda:0")] = float("inf")
    dev.surfels[2, torch.arange(7, n, 5003, device="cuda:0")] = 3.0e38
    out = {}
    for on in (1, 0):
        run.culling(on)
        out[on] = run.coeffs(run.views())
    assert np.array_equal(bits(out[1][0]), bits(out[0][0])) and np.array_equal(out[1][1], out[0][1])
    assert np.isfinite(out[1][0]).all() and out[1][1].min() > 1000


def _pcg_vectors(n, device):
    import torch
    vec = {nm: torch.full((max(1, n),), 123.0, dtype=torch.float32, device=device) for nm in ("r", "M", "delta", "g", "p")}
    scal = torch.full((3,), 7.0, dtype=torch.float32, device=device)
    v = abi.PCGVectors()
    for nm, t in vec.items():
        setattr(v, nm, t.data_ptr())
    v.alpha_n, v.alpha_d, v.beta_n = scal.data_ptr(), scal.data_ptr() + 4, scal.data_ptr() + 8
    return vec, scal, v


@pytest.mark.parametrize("use_desc,intr", [(False, False), (True, False), (True, True)])
def test_pcg_intrinsics_lifecycle_and_activation_with_and_without_culling(use_desc, intr):
    """The other kernels that walk (surfel, keyframe) pairs in the library's work order: PCGInit / PCGStep1 (per-surfel and
    per-keyframe entries, alpha_d), the alternating intrinsics step, the observation count of the surfel lifecycle and -- from
    64 keyframes on -- the activation pass: identical bits with culling on and off, on a stack where most pairs are out of view."""
    import torch
    K = 72
    dev = synthetic.TorchStack(K, "cuda:0", kind="trajectory", border_valid=True)
    run = Runner(dev, use_desc)
    L, h = run.L, run.ctx.handle
    S = dev.surfels_size
    kfs = run.views()
    cells = dev.cfactor.shape[0] * dev.cfactor.shape[1]
    per = 3 if use_desc else 1
    n_pose, n_surf = 6 * (K - 1), per * S
    depth_start = n_pose + n_surf if intr else abi.INVALID_INDEX
    color_start = (depth_start + 5 + cells) if (intr and use_desc) else abi.INVALID_INDEX
    total = n_pose + n_surf + ((5 + cells) if intr else 0) + (4 if (intr and use_desc) else 0)
    layout = abi.PCGLayout(total, n_pose, depth_start, depth_start + 4 if intr else abi.INVALID_INDEX, color_start, 3,
                           1, 1, int(intr), int(intr and use_desc), 1, int(use_desc))
    start = dev.surfels.clone()
    if use_desc:   # fitted descriptors (they start at 0)
        run.geometry(kfs)
        start = dev.surfels.clone()
    out = {}
    for on in (1, 0):
        run.culling(on)
        dev.surfels.copy_(start)
        dev.active.fill_(0)
        res = []
        dp, sb, ab = dev.depth_params(), dev.buf(dev.surfels), dev.buf(dev.active)
        # activation (K >= 64: per-surfel order + culling)
        badslam_amd.check(L.bslam_update_surfel_activation(h, run.stream, C.byref(run.cam), C.byref(dp), K, kfs, S, C.byref(sb), C.byref(ab)))
        res.append(dev.active[0, :S].cpu().numpy().copy())
        # PCG init + init2 + step1
        vec, scal, v = _pcg_vectors(total, "cuda:0")
        badslam_amd.check(L.bslam_pcg_init(h, run.stream, C.byref(layout), C.byref(run.cam), C.byref(run.cam), C.byref(dp), K, kfs, S, C.byref(sb), C.byref(v)))
        badslam_amd.check(L.bslam_pcg_init2(h, run.stream, C.byref(layout), 0.0, C.byref(v)))
        badslam_amd.check(L.bslam_pcg_step1(h, run.stream, C.byref(layout), C.byref(run.cam), C.byref(run.cam), C.byref(dp), K, kfs, S, C.byref(sb), C.byref(v), 1))
        torch.cuda.synchronize()
        res += [bits(vec[nm].cpu().numpy()) for nm in ("r", "M", "g", "p")] + [bits(scal.cpu().numpy())]
        # alternating intrinsics step
        if intr:
            out_c, out_d, a = abi.Camera4f(), abi.Camera4f(), C.c_float(0.0)
            cf0 = dev.cfactor.clone()
            badslam_amd.check(L.bslam_optimize_intrinsics(h, run.stream, 1, int(use_desc), K, kfs, C.byref(run.cam), C.byref(run.cam), C.byref(dp), S, C.byref(sb),
                                                          C.byref(out_c), C.byref(out_d), C.byref(a)))
            torch.cuda.synchronize()
            res += [np.array([out_c.fx, out_c.fy, out_c.cx, out_c.cy, out_d.fx, out_d.fy, out_d.cx, out_d.cy, a.value], np.float32).view(np.uint32),
                    bits(dev.cfactor.cpu().numpy())]
            dev.cfactor.copy_(cf0)
        # surfel lifecycle: observation counts, deletions, radii
        cnt = C.c_uint32(S)
        badslam_amd.check(L.bslam_delete_surfels_and_update_radii(h, run.stream, 2, C.byref(run.cam), C.byref(dp), K, kfs, C.byref(cnt), S, C.byref(sb)))
        torch.cuda.synchronize()
        res += [np.array([cnt.value]), bits(dev.surfels[:8].cpu().numpy())]
        out[on] = res
    dev.surfels.copy_(start)
    assert len(out[1]) == len(out[0])
    for i, (a_, b_) in enumerate(zip(out[1], out[0])):
        assert np.array_equal(a_, b_), i
    assert out[1][0].sum() > 0.9 * S                                    # activation did its job
    assert np.isfinite(out[1][1].view(np.float32)).all()


def test_mixed_keyframe_activations_with_and_without_culling():
    """Keyframe activation states enter the per-lane decision next to the frustum test (activation pass: ACTIVE only; geometry
    passes: not INACTIVE; pose loop: INACTIVE keyframes are not optimised): the same outputs with culling on and off."""
    import torch
    K = 72
    dev = synthetic.TorchStack(K, "cuda:0", kind="trajectory", border_valid=True)
    run = Runner(dev, True)
    L, h = run.L, run.ctx.handle
    S = dev.surfels_size
    kfs = run.views()
    for k in range(K):
        kfs[k].activation = abi.KF_INACTIVE if k % 3 == 1 else (abi.KF_COVISIBLE_ACTIVE if k % 5 == 2 else abi.KF_ACTIVE)
    rng = np.random.default_rng(3)
    inits = (abi.SE3f * K)()
    for k in range(K):
        inits[k] = dev.stack.pose(k, np.concatenate([rng.choice([-1, 1], 3) * 0.004, rng.choice([-1, 1], 3) * 0.001]))[0]
    start = dev.surfels.clone()
    out = {}
    for on in (1, 0):
        run.culling(on)
        dev.surfels.copy_(start)
        dev.active.fill_(0)
        dp, sb, ab = dev.depth_params(), dev.buf(dev.surfels), dev.buf(dev.active)
        badslam_amd.check(L.bslam_update_surfel_activation(h, run.stream, C.byref(run.cam), C.byref(dp), K, kfs, S, C.byref(sb), C.byref(ab)))
        act = dev.active[0, :S].cpu().numpy().copy()
        run.geometry(kfs)
        surf = bits(dev.surfels[:8].cpu().numpy())
        poses, iters = run.poses(kfs, inits, 3)
        out[on] = (act, surf, bits(poses), np.array(iters))
    dev.surfels.copy_(start)
    for a_, b_ in zip(out[1], out[0]):
        assert np.array_equal(a_, b_)
    assert 0 < out[1][0].sum() < S                                     # surfels only seen by inactive / covisible keyframes stay inactive
    inactive = np.array([k % 3 == 1 for k in range(K)])
    assert (out[1][3][inactive] == 0).all() and (out[1][3][~inactive] > 0).all()


@pytest.mark.parametrize("kind,K,use_desc", [("trajectory", 72, False), ("trajectory", 130, True), ("dense", 64, False)])
def test_batched_pose_loop_is_bit_identical_with_and_without_the_keyframe_list(kind, K, use_desc):
    """From 64 keyframes on the batched Gauss-Newton loop walks a device-side list of the keyframes that are still unconverged
    (bslam_set_pose_keyframe_list): chunks cut the list instead of the keyframe table, the grid shrinks with it.  Which keyframes
    share a chunk changes from iteration to iteration; poses, iteration counts and convergence flags must not.  The start poses
    are perturbed by very different amounts (some keyframes start at their rendered pose and stop after one step, some need many),
    so that the list shrinks unevenly, and one keyframe is INACTIVE (never in the list)."""
    dev = synthetic.TorchStack(K, "cuda:0", kind=kind, border_valid=True)
    run = Runner(dev, use_desc)
    rng = np.random.default_rng(23)
    inits = (abi.SE3f * K)()
    for k in range(K):
        scale = (0.0, 0.2, 1.0, 3.0)[k % 4]
        xi = np.concatenate([rng.choice([-1, 1], 3) * 0.004 * scale, rng.choice([-1, 1], 3) * 0.0008 * scale])
        inits[k] = dev.stack.pose(k, xi)[0]
    kfs = run.views()
    kfs[5].activation = abi.KF_INACTIVE
    out = {}
    noop = abi.ALLREDUCE_FN(lambda user, ptr, count, stream: 0)   # one rank: the sum across ranks is the identity
    # default, never, always; "hook": the kernel sequence of a surfel-sharded run (row sums, exchange, solve as separate kernels)
    for name, threshold, hook in (("default", 64, None), ("never", 0, None), ("always", 1, None), ("hook", 64, noop), ("hook, never", 0, noop)):
        badslam_amd.check(run.L.bslam_set_pose_keyframe_list(run.ctx.handle, threshold))
        dp, sb = dev.depth_params(), dev.buf(dev.surfels)
        poses = (abi.SE3f * K)()
        C.memmove(poses, inits, C.sizeof(poses))
        iters, conv = (C.c_int32 * K)(), (C.c_int32 * K)()
        badslam_amd.check(run.L.bslam_estimate_frame_poses_batched(run.ctx.handle, run.stream, 1, int(use_desc), C.byref(run.cam), C.byref(run.cam), C.byref(dp), K, kfs,
                                                                   dev.surfels_size, C.byref(sb), 12, poses, iters, conv, hook if hook else C.cast(None, abi.ALLREDUCE_FN), None))
        out[name] = (bits(np.array([[*p.q, *p.t] for p in poses], np.float32)), list(iters), list(conv))
    badslam_amd.check(run.L.bslam_set_pose_keyframe_list(run.ctx.handle, 64))
    for name in ("default", "always", "hook", "hook, never"):
        assert np.array_equal(out[name][0], out["never"][0]), name
        assert out[name][1] == out["never"][1] and out[name][2] == out["never"][2], name
    iters = out["never"][1]
    assert iters[5] == 0                                      # the inactive keyframe is never touched
    assert len(set(iters)) >= 3, sorted(set(iters))           # keyframes leave the list at different iterations

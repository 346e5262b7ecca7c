"""-m gpu: the sequential BadSlam front end (badslam_amd/host/bad_slam.*): a rendered RGB-D sequence goes frame by frame
through preprocessing, pairwise odometry with the constant-motion model, keyframe creation every 4th frame and the planned
BA iterations; the estimated trajectory must follow the rendered one (ATE) and the non-keyframe poses must
move with their keyframes when BA changes those (BS/trajectory_deformation.cc)."""
import numpy as np
import pytest

from badslam_amd import ate, bad_slam
from badslam_amd import direct_ba as dba
from tests import bso, scenes

pytestmark = pytest.mark.gpu

W, H = 320, 240


def render_sequence(n_frames, seed=3):
    rng = np.random.default_rng(seed)
    cam = bso.make_camera(262.5, 262.5, 160.0, 120.0, W, H)
    raw_to_float = np.float32(1.0 / 5000)
    planes = scenes.random_planes(rng, 20)
    step = np.array([0.010, -0.004, 0.006, 0.004, -0.006, 0.003], np.float32)     # per frame: ~1.2 cm, ~0.45 degrees
    frames, gt = [], []
    T = bso.se3_identity()
    for k in range(n_frames):
        if k:
            wobble = (0.15 * np.sin(0.9 * k + np.arange(6))).astype(np.float32)   # not exactly constant motion
            T = bso.se3_mul(T, bso.se3_exp(step * (1 + wobble)))
        M = np.array(list(bso.se3_matrix3x4(T).m), np.float64).reshape(3, 4)
        tt, pidx, dg, o = scenes.render_planes(cam, W, H, M[:, :3], M[:, 3], planes)
        valid = np.isfinite(tt) & (tt < 6.0)
        depth = np.where(valid, tt / float(raw_to_float) + 0.5, 0).astype(np.uint32)
        depth = np.where(depth >= 32768, 0, depth).astype(np.uint16)              # 0 = no measurement, as in the dataset PNGs
        pts = o[None, None, :] + dg * np.where(valid, tt, 0.0)[..., None]
        lum = scenes.texture_at(pts, pidx, 0.37)
        frames.append((depth, np.ascontiguousarray(np.repeat(lum[:, :, None], 3, axis=2))))
        gt.append(T)
    return cam, float(raw_to_float), frames, gt


def trajectory_dict(poses7):
    return {100.0 + 0.1 * i: poses7[i, 4:7].astype(np.float64) for i in range(len(poses7))}


def test_sequence_through_the_front_end(oracle):
    n = 13
    cam, raw_to_float, frames, gt = render_sequence(n)
    slam = bad_slam.BadSlam(cam, cam, keyframe_interval=4, max_num_ba_iterations_per_keyframe=5, num_scales=4, max_surfel_count=400000,
                            raw_to_float_depth=raw_to_float, max_depth=6.0, baseline_fx=40.0)
    planned = []
    for k, (depth, rgb) in enumerate(frames):
        slam.ProcessFrame(k, depth, rgb)
        st = slam.state()
        assert st["keyframe_created"] == (k % 4 == 0)
        assert st["pose_estimated"] == (k > 0)
        assert st["base_kf_id"] == k // 4
        assert 1 <= st["motion_model_length"] <= 3
        planned.append(st["num_planned_ba_iterations"])
    with pytest.raises(dba.DirectBAError):
        slam.ProcessFrame(n + 3, *frames[0])                   # frames must arrive in order
    ba = slam.ba()
    assert ba.keyframe_count() == 4
    assert ba.surfels_size() > 5000
    assert planned[0] == 0 and max(planned) <= 5      # the first keyframe plans no BA; later ones 5 iterations each, run at once

    est = slam.frame_poses()
    assert est.shape == (n, 7)
    gt7 = np.array([dba.pose7(T) for T in gt], np.float32)
    # the frames follow the rendered motion (the alternating BA fixes no gauge keyframe, so frame 0 may drift a little too)
    err_t = np.linalg.norm(est[:, 4:7] - gt7[:, 4:7], axis=1)
    moved = np.linalg.norm(gt7[-1, 4:7] - gt7[0, 4:7])
    assert moved > 0.1
    assert err_t.max() < 4e-3, err_t
    r = ate.ate(trajectory_dict(gt7), trajectory_dict(est))
    assert r["pairs"] == n and r["rmse"] < 2e-3, r["rmse"]

    # an extra BA round changes keyframe poses by a little; non-keyframes must be carried along, not left behind:
    # the relative pose between a non-keyframe and its previous keyframe stays what the odometry measured
    before = est.copy()
    done, _ = slam.RunBundleAdjustment(n - 1, False, False, True, True, 2, 3)
    assert done >= 2
    after = slam.frame_poses()
    assert np.isfinite(after).all()
    kf_shift = np.abs(after[::4] - before[::4]).max()
    # the last frame is keyframe 3: its pose is the keyframe's pose
    assert np.allclose(after[12], dba.pose7(ba.keyframe_pose(3)), atol=1e-6)
    # frame 5 lies between keyframes 1 (frame 4) and 2 (frame 8): its change is bounded by theirs
    assert np.abs(after[5] - before[5]).max() <= 2 * max(kf_shift, 1e-7) + 1e-6


def test_stationary_camera_stays_put(oracle):
    """The same frame seven times: odometry, keyframe creation every 3rd frame, BA and the pose interpolation of
    BS/trajectory_deformation.cc:45-146 must keep every frame at the anchor pose."""
    cam, raw_to_float, frames, gt = render_sequence(1)
    slam = bad_slam.BadSlam(cam, cam, keyframe_interval=3, max_num_ba_iterations_per_keyframe=2, num_scales=3, max_surfel_count=200000,
                            raw_to_float_depth=raw_to_float, max_depth=6.0)
    for k in range(7):
        slam.ProcessFrame(k, *frames[0])
    est = slam.frame_poses()
    assert np.abs(est[:, 4:7]).max() < 2e-4 and np.abs(est[:, :3]).max() < 1e-4


def test_tum_directory_through_the_front_end_to_ate(oracle, tmp_path):
    """The whole chain a user of the reference's `badslam <dataset>` needs: TUM directory (PNG depth / colour,
    associated.txt, calibration.txt, groundtruth.txt) -> reader -> BadSlam::ProcessFrame per frame -> SavePoses -> ATE."""
    from tests.test_io_cpu import write_png
    from tools import run_tum
    n = 9
    cam, raw_to_float, frames, gt = render_sequence(n, seed=5)
    (tmp_path / "rgb").mkdir()
    (tmp_path / "depth").mkdir()
    assoc, traj = [], ["# timestamp tx ty tz qx qy qz qw"]
    for k, (depth, rgb) in enumerate(frames):
        ts = f"{200.0 + 0.1 * k:.6f}"
        write_png(tmp_path / "rgb" / f"{ts}.png", rgb, filter_type=k % 5)
        write_png(tmp_path / "depth" / f"{ts}.png", depth, filter_type=(k + 1) % 5)
        assoc.append(f"{ts} rgb/{ts}.png {ts} depth/{ts}.png")
        q = bso.se3_to_np(gt[k])
        traj.append(f"{ts} {q[4]:.9g} {q[5]:.9g} {q[6]:.9g} {q[0]:.9g} {q[1]:.9g} {q[2]:.9g} {q[3]:.9g}")
    (tmp_path / "associated.txt").write_text("\n".join(assoc) + "\n")
    (tmp_path / "calibration.txt").write_text(f"{cam.fx} {cam.fy} {cam.cx - 0.5} {cam.cy - 0.5}\n")   # pixel-centre convention on disk
    (tmp_path / "groundtruth.txt").write_text("\n".join(traj) + "\n")
    r = run_tum.run(tmp_path, trajectory="groundtruth.txt", keyframe_interval=4, ba_iterations=5, max_depth=6.0, num_scales=4, max_surfel_count=400000)
    assert r["frames"] == n and r["keyframes"] == 3 and r["surfels"] > 5000
    assert r["ate"]["pairs"] == n and r["ate"]["rmse"] < 2e-3, r["ate"]["rmse"]

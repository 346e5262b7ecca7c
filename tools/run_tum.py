#!/usr/bin/env python3
"""Runs the sequential BadSlam front end over a TUM RGB-D directory (associated.txt + calibration.txt, as the reference's
dataset reader expects, LV/rgbd_video_io_tum_dataset.h:128-240), writes the trajectory in TUM format and, if the directory
has a ground truth, prints the ATE RMSE.

    python tools/run_tum.py <dataset_dir> [--trajectory groundtruth.txt] [--out poses.txt] [--keyframe-interval 10]
                            [--ba-iterations 10] [--max-depth 3.0] [--end-frame N]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from badslam_amd import abi, ate, bad_slam          # noqa: E402
from badslam_amd import direct_ba as dba            # noqa: E402


def camera_from(params, width, height):
    cam = abi.Camera4f()
    cam.fx, cam.fy, cam.cx, cam.cy = [float(v) for v in params]
    cam.width, cam.height = int(width), int(height)
    return cam


def run(dataset_dir, trajectory=None, out=None, keyframe_interval=10, ba_iterations=10, max_depth=3.0, end_frame=None, raw_to_float_depth=1.0 / 5000,
        num_scales=5, max_surfel_count=25 * 1000 * 1000):
    ds = dba.read_tum_dataset(dataset_dir, trajectory or "")
    frames = ds["frames"] if end_frame is None else ds["frames"][:end_frame]
    cam = camera_from(ds["camera"], ds["width"], ds["height"])
    slam = bad_slam.BadSlam(cam, cam, keyframe_interval=keyframe_interval, max_num_ba_iterations_per_keyframe=ba_iterations, num_scales=num_scales,
                            max_surfel_count=max_surfel_count, raw_to_float_depth=raw_to_float_depth, max_depth=max_depth)
    for k, fr in enumerate(frames):
        slam.ProcessFrame(k, dba.read_png(fr["depth_path"]), dba.read_png(fr["rgb_path"]))
    poses = slam.frame_poses()
    out = out or os.path.join(str(dataset_dir), "poses_badslam_amd.txt")
    dba.save_poses([f["depth_timestamp"] for f in frames], poses, 0, out)
    result = {"frames": len(frames), "keyframes": slam.ba().keyframe_count(), "surfels": slam.ba().surfels_size(), "poses_file": out}
    if trajectory:
        result["ate"] = ate.ate_files(os.path.join(str(dataset_dir), trajectory), out)
    return result


def main():
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("dataset_dir")
    ap.add_argument("--trajectory", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--keyframe-interval", type=int, default=10)
    ap.add_argument("--ba-iterations", type=int, default=10)
    ap.add_argument("--max-depth", type=float, default=3.0)
    ap.add_argument("--end-frame", type=int, default=None)
    a = ap.parse_args()
    r = run(a.dataset_dir, a.trajectory, a.out, a.keyframe_interval, a.ba_iterations, a.max_depth, a.end_frame)
    print(f"{r['frames']} frames, {r['keyframes']} keyframes, {r['surfels']} surfels -> {r['poses_file']}")
    if "ate" in r:
        print(f"ATE RMSE {r['ate']['rmse']:.6f} m over {r['ate']['pairs']} poses")


if __name__ == "__main__":
    main()

"""Absolute trajectory error (ATE RMSE) between a TUM-format ground truth and an exported trajectory.

SURVEY.md 8(d): "Accuracy = ATE RMSE of exported TUM-format trajectory (BS/io.cc:537) computed with our own
Horn-alignment tool against groundtruth.txt".  The reference itself ships no such tool (its README points to
the TUM benchmark scripts); this is an independent restatement of the published method: associate by
timestamp (nearest, |dt| < max_difference, one-to-one, greedy by |dt|), solve the rigid alignment
(rotation + translation, no scale) in closed form from the SVD of the cross-covariance (Horn 1987 / Arun 1987),
and report the RMSE of the remaining translational differences.

Evaluation helper only: it never touches the GPU and the BA path does not depend on it.

    python -m badslam_amd.ate groundtruth.txt poses.txt [--max-difference 0.02] [--offset 0]
"""
import sys

import numpy as np


def read_trajectory(path):
    """{timestamp(float): xyz(float64[3])} from lines `timestamp tx ty tz qx qy qz qw` ('#' lines skipped)."""
    out = {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            w = line.replace(",", " ").split()
            if len(w) < 4:
                continue
            v = np.array([float(x) for x in w[1:4]], np.float64)
            if np.isfinite(v).all():
                out[float(w[0])] = v
    return out


def associate(first_stamps, second_stamps, offset=0.0, max_difference=0.02):
    """One-to-one (a, b) timestamp pairs with |a - (b + offset)| < max_difference, best matches first."""
    a = np.asarray(sorted(first_stamps), np.float64)
    b = np.asarray(sorted(second_stamps), np.float64)
    if a.size == 0 or b.size == 0:
        return []
    cand = []
    bo = b + offset
    for i, t in enumerate(a):
        j = int(np.searchsorted(bo, t))
        for jj in (j - 1, j, j + 1):
            if 0 <= jj < b.size:
                d = abs(t - bo[jj])
                if d < max_difference:
                    cand.append((d, i, jj))
    cand.sort()
    used_a, used_b, pairs = set(), set(), []
    for d, i, j in cand:
        if i in used_a or j in used_b:
            continue
        used_a.add(i)
        used_b.add(j)
        pairs.append((float(a[i]), float(b[j])))
    pairs.sort()
    return pairs


def align_rigid(model, data):
    """R (3x3), t (3,) minimising sum |R model_i + t - data_i|^2; model, data are (N, 3)."""
    model = np.asarray(model, np.float64)
    data = np.asarray(data, np.float64)
    mc, dc = model.mean(axis=0), data.mean(axis=0)
    W = (model - mc).T @ (data - dc)
    U, _, Vt = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    t = dc - R @ mc
    return R, t


def ate(ground_truth, estimate, offset=0.0, max_difference=0.02):
    """ground_truth / estimate: {timestamp: xyz}.  Returns a dict with rmse, mean, median, max, pairs."""
    pairs = associate(ground_truth.keys(), estimate.keys(), offset, max_difference)
    if len(pairs) < 3:
        raise ValueError(f"only {len(pairs)} associated poses; need >= 3 for the alignment")
    gt = np.array([ground_truth[a] for a, _ in pairs])
    est = np.array([estimate[b] for _, b in pairs])
    R, t = align_rigid(est, gt)
    err = np.linalg.norm(est @ R.T + t - gt, axis=1)
    return {"rmse": float(np.sqrt((err ** 2).mean())), "mean": float(err.mean()), "median": float(np.median(err)),
            "max": float(err.max()), "pairs": len(pairs), "R": R, "t": t}


def ate_files(ground_truth_path, estimate_path, offset=0.0, max_difference=0.02):
    return ate(read_trajectory(ground_truth_path), read_trajectory(estimate_path), offset, max_difference)


def main(argv):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("ground_truth")
    ap.add_argument("estimate")
    ap.add_argument("--offset", type=float, default=0.0)
    ap.add_argument("--max-difference", type=float, default=0.02)
    a = ap.parse_args(argv)
    r = ate_files(a.ground_truth, a.estimate, a.offset, a.max_difference)
    print(f"compared_pose_pairs {r['pairs']}")
    for k in ("rmse", "mean", "median", "max"):
        print(f"absolute_translational_error.{k} {r[k]:.6f} m")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

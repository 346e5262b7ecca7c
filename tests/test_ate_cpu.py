"""CPU: the ATE evaluation helper on trajectories with a known answer."""
import numpy as np
import pytest

from badslam_amd import ate


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _write(path, stamps, xyz):
    with open(path, "w") as f:
        f.write("# timestamp tx ty tz qx qy qz qw\n")
        for t, p in zip(stamps, xyz):
            f.write(f"{t:.6f} {p[0]:.9f} {p[1]:.9f} {p[2]:.9f} 0 0 0 1\n")


def test_rigidly_moved_trajectory_has_zero_error(tmp_path):
    rng = np.random.default_rng(0)
    n = 200
    stamps = 1000.0 + 0.033 * np.arange(n)
    gt = np.cumsum(rng.normal(scale=0.02, size=(n, 3)), axis=0)
    R, t = _rot(rng), rng.normal(size=3)
    est = (gt - t) @ R                  # est = R^T (gt - t)  <=>  gt = R est + t
    _write(tmp_path / "gt.txt", stamps, gt)
    _write(tmp_path / "est.txt", stamps + 0.004, est)      # shifted stamps still associate (|dt| < 0.02)
    r = ate.ate_files(tmp_path / "gt.txt", tmp_path / "est.txt")
    assert r["pairs"] == n
    assert r["rmse"] < 1e-7
    assert np.allclose(r["R"], R, atol=1e-6) and np.allclose(r["t"], t, atol=1e-6)


def test_known_residual_and_partial_association(tmp_path):
    rng = np.random.default_rng(1)
    n = 400
    stamps = 50.0 + 0.1 * np.arange(n)
    gt = np.stack([np.cos(stamps * 0.3), np.sin(stamps * 0.3), 0.1 * stamps], axis=1)
    noise = rng.normal(scale=0.01, size=(n, 3))
    noise -= noise.mean(axis=0)
    R, t = _rot(rng), rng.normal(size=3)
    est = (gt + noise - t) @ R
    keep = np.arange(n) % 3 != 0         # the estimate holds only 2 of 3 poses ...
    est_stamps = stamps[keep].copy()
    est_stamps[::7] += 0.05              # ... and some of those are too far away in time to associate
    _write(tmp_path / "gt.txt", stamps, gt)
    _write(tmp_path / "est.txt", est_stamps, est[keep])
    r = ate.ate_files(tmp_path / "gt.txt", tmp_path / "est.txt")
    expected_pairs = keep.sum() - len(est_stamps[::7])
    assert r["pairs"] == expected_pairs
    # the optimal alignment can only lower the error below that of the generating transform
    used = np.ones(keep.sum(), bool)
    used[::7] = False
    bound = np.sqrt((np.linalg.norm(noise[keep][used], axis=1) ** 2).mean())
    assert 0.8 * bound < r["rmse"] <= bound * (1 + 1e-9)


def test_mirror_configuration_returns_a_proper_rotation():
    # near-planar data where the unconstrained least-squares solution is a reflection
    rng = np.random.default_rng(2)
    model = np.concatenate([rng.normal(size=(50, 2)), 1e-9 * rng.normal(size=(50, 1))], axis=1)
    data = model * np.array([1.0, 1.0, -1.0])
    R, t = ate.align_rigid(model, data)
    assert np.linalg.det(R) > 0.999


def test_too_few_pairs_is_an_error():
    with pytest.raises(ValueError):
        ate.ate({1.0: np.zeros(3), 2.0: np.ones(3)}, {1.0: np.zeros(3), 2.0: np.ones(3)})
    assert ate.associate([], [1.0]) == []
    assert ate.associate([1.0, 1.01], [1.004]) == [(1.0, 1.004)]     # one-to-one, closest first

"""Pins the oracle's residual / Jacobian formulas to the REFERENCE's own symbolic derivation: tests/golden/jacobian_golden.npz
holds float64 values of the residuals and Jacobians that applications/badslam/scripts/jacobians_derivation.py defines
(generated in the build container by tests/golden/make_jacobian_golden.py, which imports the reference's scripts; only
numbers travel).  The functions probed are the ones every loop of the oracle calls (oracle/bso_math.h "Jacobians")."""
import ctypes as C

import numpy as np

from tests import bso, jacobian_fixtures


def oracle_probe(kind, inputs):
    out = np.zeros((inputs.shape[0], jacobian_fixtures.OUT_WIDTH[kind]), np.float32)
    rc = bso.lib().bso_jacobian_probe(kind, inputs.shape[0], inputs.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0
    return out


def test_fixture_file_is_complete():
    K = jacobian_fixtures.load()
    assert set(K) == {"depth_pose", "depth_position", "depth_intrinsics", "depth_deformation", "desc_pose", "desc_position", "desc_color_intrinsics"}
    for k in K.values():
        assert k.n >= 1000 and np.isfinite(k.jacobian).all() and np.isfinite(k.residual).all()
    assert K["depth_pose"].jacobian.shape[1] == 6 and K["desc_pose"].jacobian.shape[1] == 6
    assert K["depth_intrinsics"].jacobian.shape[1] == 4 and K["depth_deformation"].jacobian.shape[1] == 2


def test_oracle_jacobians_match_the_reference_derivation(oracle):
    """fp32 oracle formulas vs the float64 symbolic values at 1 000 random points per residual type: 1e-5 of the sample's
    largest entry (the inputs themselves are rounded to fp32: 6e-8 relative each)."""
    report = jacobian_fixtures.check(oracle_probe, 1e-5)
    assert len(report) == 10, report


def test_oracle_jacobians_literal_mode_matches_too(oracle):
    """The literal-transcription shape of the oracle (unfused sums, libm expf) against the same fixtures."""
    with bso.literal_mode():
        jacobian_fixtures.check(oracle_probe, 1e-5)

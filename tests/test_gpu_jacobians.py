"""-m gpu: the device residual / Jacobian functions (csrc/device_math.hpp, the ones every kernel calls) against the values of
the REFERENCE's own symbolic derivation (tests/golden/jacobian_golden.npz, see tests/test_jacobian_golden.py), through the
C-ABI probe bslam_debug_jacobians."""
import ctypes as C

import numpy as np
import pytest

import badslam_amd
from tests import jacobian_fixtures

pytestmark = pytest.mark.gpu


def hip_probe_factory(remap=None):
    ctx = badslam_amd.Context(0)
    L = badslam_amd.lib()

    def probe(kind, inputs):
        k = (remap or {}).get(kind, kind)
        out = np.zeros((inputs.shape[0], jacobian_fixtures.OUT_WIDTH[kind]), np.float32)
        badslam_amd.check(L.bslam_debug_jacobians(ctx.handle, None, k, inputs.shape[0], inputs.ctypes.data_as(C.POINTER(C.c_float)),
                                                  out.ctypes.data_as(C.POINTER(C.c_float))))
        return out
    probe.ctx = ctx
    return probe


def test_device_jacobians_match_the_reference_derivation():
    """fp32 device formulas vs the float64 symbolic values at 1 000 random points per residual type, 1e-4 of the sample's largest
    entry (the kernels' residual math uses the 1-ulp hardware reciprocal; the oracle holds the same fixtures at 1e-5)."""
    report = jacobian_fixtures.check(hip_probe_factory(), 1e-4)
    assert len(report) == 10, report
    assert max(report.values()) < 1e-4


def test_pose_kernel_fused_form_matches_too():
    """The pose kernel evaluates the depth residual and its Jacobian with fused multiply-adds (probe kind 6)."""
    jacobian_fixtures.check(hip_probe_factory({0: 6}), 1e-4)


def test_device_and_oracle_probes_agree(oracle):
    """Same points through the oracle's formulas: the two fp32 implementations agree to 2e-6 of the largest entry."""
    from tests.test_jacobian_golden import oracle_probe
    hip = hip_probe_factory()
    for name, kind, inp, _ in jacobian_fixtures.cases():
        x = np.ascontiguousarray(inp, np.float32)
        a, b = hip(kind, x).astype(np.float64), oracle_probe(kind, x).astype(np.float64)
        scale = np.maximum(np.abs(b).max(axis=1, keepdims=True), 1e-30)
        assert (np.abs(a - b) / scale).max() < 2e-6, (name, (np.abs(a - b) / scale).max())


def test_midrange_reciprocal_is_correctly_rounded():
    """rcp_rn_midrange (csrc/device_math.hpp: v_rcp + the fma chain of the compiler's own IEEE-division expansion, 7 instructions
    instead of 11) feeds the association predicates, whose integer outputs are compared bit for bit with the CPU: it must equal
    1.0f / x exactly.  Swept over EVERY fp32 mantissa at three exponents (the rounding of a reciprocal does not depend on the
    exponent inside the normal range) and 4 M random finite values in 2^-90 .. 2^90, both signs."""
    ctx = badslam_amd.Context(0)
    L = badslam_amd.lib()

    def run(x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros((x.size, 2), np.float32)
        badslam_amd.check(L.bslam_debug_jacobians(ctx.handle, None, 7, x.size, x.ctypes.data_as(C.POINTER(C.c_float)),
                                                  out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    mant = np.arange(1 << 23, dtype=np.uint32)
    for exp in (127, 100, 140):
        x = (mant | np.uint32(exp << 23)).view(np.float32)
        out = run(x)
        assert np.array_equal(out[:, 0].view(np.uint32), out[:, 1].view(np.uint32)), exp
        assert np.array_equal(out[:, 1], (np.float32(1) / x).astype(np.float32))          # and both equal the IEEE quotient
    rng = np.random.default_rng(3)
    bits = (rng.integers(37, 217, 1 << 22, dtype=np.uint32) << 23) | rng.integers(0, 1 << 23, 1 << 22, dtype=np.uint32) | (rng.integers(0, 2, 1 << 22, dtype=np.uint32) << 31)
    out = run(bits.view(np.float32))
    assert np.array_equal(out[:, 0].view(np.uint32), out[:, 1].view(np.uint32))

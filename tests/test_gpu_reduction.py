"""-m gpu: the wave reduction behind every per-keyframe sum (wave_column_sums_lds, csrc/device_math.hpp) on its own.

H, b and the residual counts of the pose step, the r / M / g pose entries of the PCG kernels all leave a wave through this
function: 27 (28 with the robust cost) columns in rounds of 4 (photometric kernels) or 8 (geometry-only), the PCG kernels' 6 and
12 columns in rounds of 4, two stages through a wave-private LDS tile.  Through bslam_debug_wave_column_sums: exact column sums
on integer-valued inputs (any summation order gives the same fp32 value), zeros for the columns a configuration does not sum,
untouched outputs for the columns nobody owns, float inputs against float64 sums, and the same bits on every run."""
import ctypes as C

import numpy as np
import pytest

import badslam_amd

pytestmark = pytest.mark.gpu
P = C.POINTER
CONFIGS = [(27, 4), (28, 4), (27, 8), (28, 8), (6, 4), (12, 4)]


def probe(L, ctx, stream, live, cols, values, fill):
    v = np.ascontiguousarray(values, np.float32)
    assert v.shape == (64, 32)
    out = np.full(32, fill, np.float32)
    badslam_amd.check(L.bslam_debug_wave_column_sums(ctx.handle, stream, live, cols, v.ctypes.data_as(P(C.c_float)), out.ctypes.data_as(P(C.c_float))))
    return out


def owned_columns(live, cols):
    """Columns that have an owner lane: one batch of four rounds covers 4 * cols columns, two batches all 32."""
    rounds = -(-live // cols)
    return 32 if rounds > 4 else min(32, 4 * cols)


@pytest.mark.parametrize("live,cols", CONFIGS)
def test_column_sums_are_exact_on_integers(live, cols):
    import torch
    L, ctx = badslam_amd.lib(), badslam_amd.Context(0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(1000 * live + cols)
    owned = owned_columns(live, cols)
    for trial in range(4):
        v = rng.integers(-9, 10, size=(64, 32)).astype(np.float32)
        if trial == 1:
            v[:] = 0
            v[37, :] = np.arange(32) + 1                     # one lane only: every column must come from lane 37
        if trial == 2:
            v = np.tile((np.arange(64, dtype=np.float32) - 31)[:, None], (1, 32)) * (np.arange(32, dtype=np.float32) + 1)[None, :]
        out = probe(L, ctx, stream, live, cols, v, fill=-777.0)
        want = v.astype(np.float64).sum(axis=0)
        assert np.array_equal(out[:live], want[:live].astype(np.float32)), (trial, out[:live], want[:live])
        assert np.all(out[live:owned] == 0.0)                # summed by nobody, owned by a lane: zero
        assert np.all(out[owned:] == -777.0)                 # no owner in this configuration: left alone


@pytest.mark.parametrize("live,cols", CONFIGS)
def test_column_sums_on_floats_are_deterministic_and_accurate(live, cols):
    import torch
    L, ctx = badslam_amd.lib(), badslam_amd.Context(0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(7 + live)
    v = (rng.standard_normal((64, 32)) * np.exp(rng.uniform(-3, 3, (1, 32)))).astype(np.float32)
    a = probe(L, ctx, stream, live, cols, v, fill=0.0)
    b = probe(L, ctx, stream, live, cols, v, fill=0.0)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    want = v.astype(np.float64).sum(axis=0)
    scale = np.abs(v.astype(np.float64)).sum(axis=0)
    assert np.all(np.abs(a[:live] - want[:live]) <= 4e-7 * scale[:live])

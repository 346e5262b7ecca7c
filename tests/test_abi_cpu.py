"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, exports every symbol
include/badslam_hip.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import badslam_amd
from badslam_amd import abi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hiplib():
    build.build()
    return badslam_amd.lib()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "badslam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(bslam_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if n not in ("bslam_allreduce_fn",)))


def test_every_declared_symbol_is_exported(hiplib):
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(hiplib, n), f"{n} declared in include/badslam_hip.h but not exported"
        assert n in abi.SIGNATURES, f"{n} has no ctypes signature in badslam_amd/abi.py"


def test_struct_layouts_match_the_reference_pods():
    # CUDABuffer_<T>: {T* address; int height; int width; size_t pitch} = 24 bytes on LP64
    assert C.sizeof(abi.Buffer2D) == 24
    assert abi.Buffer2D.height.offset == 8 and abi.Buffer2D.width.offset == 12 and abi.Buffer2D.pitch.offset == 16
    assert C.sizeof(abi.Mat3x4) == 48 and C.sizeof(abi.Mat3x3) == 36
    assert C.sizeof(abi.DepthParams) == 40   # CUDABuffer_ + 3 floats + int
    assert C.sizeof(abi.SE3f) == 28


def test_abi_version(hiplib):
    assert hiplib.bslam_abi_version() == 1


def test_no_cpu_fallback(hiplib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = hiplib.bslam_create(0, C.byref(ctx))
    assert rc == -3, "bslam_create must fail with BSLAM_ERR_NO_DEVICE when no GPU is visible"
    assert b"no CPU fallback" in hiplib.bslam_last_error()


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "badslam_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h", ".inc", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in text and "bslam_oracle" not in text and "bso_" not in text, f

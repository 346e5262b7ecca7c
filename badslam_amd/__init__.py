"""badslam_amd -- MI355X-native bundle-adjustment hot path of BAD SLAM behind a C ABI.

`lib()` loads the in-tree HIP library (libbadslam_hip.so, built by badslam_amd/build.py).
There is no CPU fallback: if the library is missing or no GPU is visible, calls fail loudly.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# BSLAM_HIP_LIB points tools/variants.py's tuning builds at an alternative kernel library
LIB_PATH = os.environ.get("BSLAM_HIP_LIB") or os.path.join(_HERE, "libbadslam_hip.so")

_lib = None


class BadSlamError(RuntimeError):
    pass


def preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (same soname as /opt/rocm's).  If our
    libraries are loaded first they pull in /opt/rocm's copy and torch then loads a second HIP
    runtime, which finds no GPU.  Importing torch first makes the bundled copy the one and only
    runtime of the process; without torch installed there is nothing to do."""
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib():
    """Loads libbadslam_hip.so and applies the signatures of include/badslam_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    preload_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise BadSlamError(
            f"{LIB_PATH} is missing: build it with `python -m badslam_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in abi.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise BadSlamError(f"bslam error {rc}: {lib().bslam_last_error().decode()}")


def comm_unique_id():
    """128-byte RCCL unique id (bslam_comm_get_unique_id): rank 0 creates it, every rank passes it to Context.comm_init."""
    buf = C.create_string_buffer(128)
    check(lib().bslam_comm_get_unique_id(buf, 128))
    return buf.raw


class Context:
    """RAII wrapper of bslam_context."""

    def __init__(self, device=0):
        self._ctx = C.c_void_p()
        check(lib().bslam_create(device, C.byref(self._ctx)))

    @property
    def handle(self):
        return self._ctx

    def set_texture_mode(self, mode):
        check(lib().bslam_set_texture_mode(self._ctx, mode))

    def comm_init(self, unique_id, rank, world_size):
        """The library's own RCCL communicator for surfel-sharded runs (bslam_comm_init)."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(lib().bslam_comm_init(self._ctx, buf, rank, world_size))

    def comm_query(self):
        """(rank, world size) as the RCCL communicator reports them; (0, 1) without one."""
        r, w = C.c_int(), C.c_int()
        check(lib().bslam_comm_query(self._ctx, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_destroy(self):
        check(lib().bslam_comm_destroy(self._ctx))

    def close(self):
        if self._ctx:
            lib().bslam_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""Multi-GPU plumbing: one process per GPU, surfels sharded, keyframes replicated
(SURVEY.md 8e).  The only exchange on the pose path is a sum of K x 32 floats of
Gauss-Newton coefficients per batched iteration; it goes through torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm, "gloo" in the CPU tests).  Every rank applies
identical updates from identical reduced sums, so poses stay bit-identical without a
broadcast.
"""
import ctypes as C

import numpy as np

from . import abi


def shard_range(surfels_size, rank, world_size):
    """Contiguous surfel index range [lo, hi) of `rank` (SURVEY.md 8e partition)."""
    lo = (surfels_size * rank) // world_size
    hi = (surfels_size * (rank + 1)) // world_size
    return lo, hi


class _CudaBlob:
    """Exposes a raw device pointer through __cuda_array_interface__ so that torch can alias it."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class AllReduceHook:
    """bslam_allreduce_fn backed by torch.distributed.all_reduce(SUM).

    device=True : the buffer is device memory of the current CUDA device (product path).
    device=False: the buffer is host memory (CPU/gloo tests of the same plumbing).
    """

    def __init__(self, device=True, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.device = device
        self.group = group
        self.calls = 0
        self._views = {}   # (ptr, count) -> torch view of the library's device buffer (the slabs are long-lived)
        self.callback = abi.ALLREDUCE_FN(self._call)

    def _call(self, user, ptr, count, stream):
        try:
            if self.device:
                t = self._views.get((ptr, count))
                if t is None:
                    if len(self._views) > 64:
                        self._views.clear()
                    t = self._views[(ptr, count)] = self.torch.as_tensor(_CudaBlob(ptr, count), device="cuda")
            else:
                arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(count,))
                t = self.torch.from_numpy(arr)
            if self.device and stream:
                # The library enqueued the producers on `stream`: make it torch's current stream for the call, so
                # the collective is ordered after them and the library's next kernels after the collective.
                with self.torch.cuda.stream(self.torch.cuda.ExternalStream(stream)):
                    self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            else:   # the NULL stream is torch's default stream
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            self.calls += 1
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("allreduce hook failed:", repr(e), flush=True)
            return 1

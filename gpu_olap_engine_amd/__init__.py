"""MI355X-native GpuHashJoin / GpuAggregate / GpuFilter execution path.

Layers (no CPU fallback anywhere):
  include/golap.h            C ABI (drop-in boundary)
  csrc/*.hip                 hand-written gfx950 kernels + the C-ABI implementation
  _native.py / engine.py     ctypes binding + device-level host API (torch tensors)
  operators.py               PhysicalPlan operator API over Arrow RecordBatches
  distributed.py             one-process-per-GPU sharded join (RCCL all-to-all over xGMI)
"""
from . import _native
from ._native import GolapError, LIB_PATH  # noqa: F401

__all__ = ["_native", "GolapError", "LIB_PATH"]

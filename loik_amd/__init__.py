"""loik_amd -- MI355X-native batched LoIK (constrained differential IK) solver.

Thin Python host layer over the C-ABI shared library ``loik_amd/lib/libloik_amd.so`` (hand-written HIP kernels for
gfx950, see ``loik_amd/csrc``).  There is NO CPU fallback: importing :mod:`loik_amd.capi` raises if the HIP
library has not been built (``python -c "import __graft_entry__ as g; g.build()"``).
"""
from .capi import (BatchedLoik, Model, builtin_model, device_count, lib, LoikError)  # noqa: F401

__all__ = ["BatchedLoik", "Model", "builtin_model", "device_count", "lib", "LoikError"]

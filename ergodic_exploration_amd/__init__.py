"""MI355X-native engine for the receding-horizon ergodic control path of
bostoncleek/ergodic_exploration (`ErgodicControl<ModelT>::control`).

csrc/   hand-written HIP kernels for gfx950 + the C ABI (include/ergodic_amd.h)
host/   C++ mirror of the reference's class surface on top of the C ABI
capi.py ctypes view of the C ABI used by tests/ and bench.py
"""
from . import capi  # noqa: F401

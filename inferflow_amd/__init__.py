"""inferflow_amd -- MI355X-native quantized transformer decode path behind
Inferflow's InferenceEngine / TensorOpr surface.

The compute lives in inferflow_amd/lib/libinferflow_amd.so (hand-written HIP for
gfx950 + a C ABI, include/inferflow_amd.h).  This package only binds it; there
is no CPU fallback: importing ops without the built library raises.
"""
from . import dtypes  # noqa: F401
from ._capi import lib, check, IfaError, library_path  # noqa: F401

__version__ = "0.1"

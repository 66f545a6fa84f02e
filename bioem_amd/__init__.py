"""bioem_amd -- MI355X-native BioEM likelihood engine (compare path only).

The product is the C-ABI shared library bioem_amd/lib/libbioem_hip.so (HIP kernels for gfx950,
include/bioem_hip.h) plus the C++ host layer in bioem_amd/host.  This Python package is a thin
ctypes binding used by tests/ and bench.py; there is no CPU fallback: loading fails loudly when
the HIP library is missing.
"""
from .engine import Engine, ParamDevice, load_library, lib_path  # noqa: F401

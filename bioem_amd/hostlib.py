"""ctypes binding of the C++ host layer's C entry points (bioem_amd/host/capi.cpp, libbioem_host.so):
the one-off precompute the reference does on the host before the hot loop."""
import ctypes as C
import os

import numpy as np

from .engine import POINT_DTYPE, ParamDevice, load_library

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def load_host_library():
    global _lib
    if _lib is None:
        load_library()  # libbioem_hip.so first (dependency)
        path = os.path.join(HERE, "lib", "libbioem_host.so")
        if not os.path.exists(path):
            raise RuntimeError("host layer not built: %s missing (make -C bioem_amd/csrc host)" % path)
        L = C.CDLL(path)
        ci, cf, vp = C.c_int, C.c_float, C.c_void_p
        L.bioem_host_ctf_kernels.argtypes = [ci, cf, cf, cf, ci, cf, cf, ci, cf, cf, ci, vp, vp, vp]
        L.bioem_host_ctf_kernels.restype = ci
        L.bioem_host_volume_element.argtypes = [cf, ci, ci, cf, ci, cf, cf, cf, cf, cf]
        L.bioem_host_volume_element.restype = cf
        L.bioem_host_center_model.argtypes = [vp, ci]
        L.bioem_host_center_model.restype = cf
        _lib = L
    return _lib


def ctf_kernels(N, pixelSize, amp, phase, env):
    """amp/phase/env = (start, end, n) in the reference's internal units (phase = defocus*2*pi*1e4*lambda).
    Returns refCTF [nCTF,N,H,2], ctfParam [nCTF,3], steps[3]."""
    L = load_host_library()
    n = amp[2] * phase[2] * env[2]
    H = N // 2 + 1
    ref = np.zeros((n, N, H, 2), dtype=np.float32)
    par = np.zeros((n, 3), dtype=np.float32)
    steps = np.zeros(3, dtype=np.float32)
    got = L.bioem_host_ctf_kernels(N, pixelSize, amp[0], amp[1], amp[2], phase[0], phase[1], phase[2], env[0], env[1],
                                   env[2], ref.ctypes.data, par.ctypes.data, steps.ctypes.data)
    assert got == n
    return ref, par, steps


def volume_element(voluang, gridSpace, maxD, pixelSize, nAmp, gridEnv, gridPhase, sigB, sigDef, sigAmp):
    return load_host_library().bioem_host_volume_element(voluang, gridSpace, maxD, pixelSize, nAmp, gridEnv,
                                                        gridPhase, sigB, sigDef, sigAmp)


def center_model(points):
    pts = np.ascontiguousarray(points.copy())
    nd = load_host_library().bioem_host_center_model(pts.ctypes.data, len(pts))
    return pts, np.float32(nd)

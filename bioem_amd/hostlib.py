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


class HostSetup(C.Structure):
    _fields_ = [("pd", ParamDevice), ("nAngles", C.c_int), ("nCTF", C.c_int), ("isQuat", C.c_int),
                ("usepsf", C.c_int), ("shiftX", C.c_int), ("shiftY", C.c_int), ("nocentermass", C.c_int),
                ("pixelSize", C.c_float), ("voluang", C.c_float), ("elecwavel", C.c_float)]


def setup_from_files(paramfile, anglefile=None):
    """readParameters + CalculateGridsParam (+ CalculateRefCTF in CTF mode) of the C++ host layer."""
    L = load_host_library()
    L.bioem_host_setup_from_files.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(HostSetup), C.c_void_p, C.c_void_p,
                                              C.c_void_p]
    S = HostSetup()
    af = anglefile.encode() if anglefile else None
    L.bioem_host_setup_from_files(paramfile.encode(), af, C.byref(S), None, None, None)
    N = S.pd.NumberPixels
    H = N // 2 + 1
    angles = np.zeros((S.nAngles, 4), dtype=np.float32)
    ref = np.zeros((S.nCTF, N, H, 2), dtype=np.float32)
    par = np.zeros((S.nCTF, 3), dtype=np.float32)
    L.bioem_host_setup_from_files(paramfile.encode(), af, C.byref(S), angles.ctypes.data, ref.ctypes.data,
                                  par.ctypes.data)
    return S, angles, ref, par


def read_model(path, isPDB=False, nocentermass=False, cap=1 << 20, isMRC=False, pixelSize=1.0):
    L = load_host_library()
    L.bioem_host_read_model.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_float)]
    L.bioem_host_read_model.restype = C.c_int
    pts = np.zeros(cap, dtype=POINT_DTYPE)
    nd = C.c_float()
    kind = 2 if isMRC else int(bool(isPDB))
    n = L.bioem_host_read_model(path.encode(), kind, int(nocentermass), float(pixelSize), pts.ctypes.data, cap,
                                C.byref(nd))
    return pts[:n].copy(), np.float32(nd.value)


def read_particles(path, N, mode=0, notnormmap=False, cap=4096):
    L = load_host_library()
    L.bioem_host_read_particles.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.bioem_host_read_particles.restype = C.c_int
    maps = np.zeros((cap, N, N), dtype=np.float32)
    n = L.bioem_host_read_particles(path.encode(), mode, N, int(notnormmap), maps.ctypes.data, cap)
    return maps[:n].copy()

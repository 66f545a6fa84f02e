"""ctypes binding of include/bioem_hip.h (libbioem_hip.so).  No fallback path: if the HIP library
cannot be loaded this module raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

PROB_MAP_DTYPE = np.dtype([("Total", "<f8"), ("Constoadd", "<f8"), ("cent_x", "<i4"), ("cent_y", "<i4"),
                           ("orient", "<i4"), ("conv", "<i4"), ("norm", "<f4"), ("mu", "<f4")])
PROB_ANGLE_DTYPE = np.dtype([("forAngles", "<f8"), ("ConstAngle", "<f8")])
PARAM5_DTYPE = np.dtype([("amp", "<f4"), ("pha", "<f4"), ("env", "<f4"), ("sumC", "<f4"), ("sumsquareC", "<f4")])
POINT_DTYPE = np.dtype([("pos", "<f4", (3,)), ("quat4", "<f4"), ("radius", "<f4"), ("density", "<f4")])
PHASE_RECORD_DTYPE = np.dtype([("phase", "<i4"), ("iOrientBegin", "<i4"), ("iOrientEnd", "<i4"), ("iConvBegin", "<i4"),
                               ("iConvEnd", "<i4"), ("pad", "<i4"), ("seconds", "<f8")])
CANDIDATE_DTYPE = np.dtype([("forAngles", "<f8"), ("ConstAngle", "<f8"), ("logp", "<f8"), ("orient", "<i4"),
                            ("pad", "<i4")])
MIN_PROB = -999999.0


class ParamDevice(C.Structure):
    """bioem_hip_param_device == bioem_param_device (reference include/param.h:26-47)."""
    _fields_ = [("maxDisplaceCenter", C.c_int), ("GridSpaceCenter", C.c_int), ("NumberPixels", C.c_int),
                ("NumberFFTPixels1D", C.c_int), ("NxDisp", C.c_int), ("NtotDisp", C.c_int),
                ("Ntotpi", C.c_float), ("volu", C.c_float), ("sigmaPriorbctf", C.c_float),
                ("sigmaPriordefo", C.c_float), ("Priordefcent", C.c_float), ("sigmaPrioramp", C.c_float),
                ("Priorampcent", C.c_float), ("writeAngles", C.c_int), ("tousepsf", C.c_int)]


def lib_path():
    # BIOEM_HIP_LIBRARY: an experiment build of the same library (scripts/slim_build.sh) for same-box A/B runs
    alt = os.environ.get("BIOEM_HIP_LIBRARY")
    if alt:
        return alt if os.path.isabs(alt) else os.path.join(os.path.dirname(HERE), alt)
    return os.path.join(HERE, "lib", "libbioem_hip.so")


_lib = None


def load_library():
    """Loads libbioem_hip.so and declares every entry point of include/bioem_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError("HIP engine not built: %s is missing (run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` or `make -C bioem_amd/csrc`); there is no CPU fallback" % path)
    L = C.CDLL(path)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.bioem_hip_device_count.restype = ci
    L.bioem_hip_create.argtypes = [C.POINTER(vp), ci, C.POINTER(ParamDevice), ci, ci, ci, ci]
    L.bioem_hip_create_shard.argtypes = [C.POINTER(vp), ci, C.POINTER(ParamDevice), ci, ci, ci, ci, ci, ci]
    L.bioem_hip_destroy.argtypes = [vp]
    L.bioem_hip_last_error.argtypes = [vp]
    L.bioem_hip_last_error.restype = C.c_char_p
    L.bioem_hip_upload_particles.argtypes = [vp, vp, vp, vp]
    L.bioem_hip_upload_particle_maps.argtypes = [vp, vp]
    L.bioem_hip_upload_ctf.argtypes = [vp, vp, vp]
    L.bioem_hip_upload_model.argtypes = [vp, vp, ci, cf, cf, ci, ci]
    L.bioem_hip_upload_orientations.argtypes = [vp, vp, ci, ci]
    L.bioem_hip_host_alloc.argtypes = [C.c_size_t]
    L.bioem_hip_host_alloc.restype = vp
    L.bioem_hip_host_free.argtypes = [vp]
    L.bioem_hip_prob_size.argtypes = [ci, ci, ci]
    L.bioem_hip_prob_size.restype = C.c_size_t
    L.bioem_hip_start_run.argtypes = [vp, vp]
    L.bioem_hip_compare.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp]
    L.bioem_hip_project_convolve_compare.argtypes = [vp, ci, ci]
    L.bioem_hip_project_convolve_compare_ctf.argtypes = [vp, ci, ci, ci, ci]
    L.bioem_hip_project.argtypes = [vp, ci, ci, ci]
    L.bioem_hip_convolve.argtypes = [vp, ci, ci, ci]
    L.bioem_hip_compare_device.argtypes = [vp, ci]
    L.bioem_hip_max_batch.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    L.bioem_hip_finish_run.argtypes = [vp, vp]
    L.bioem_hip_set_phase_timing.argtypes = [vp, ci]
    L.bioem_hip_phase_records.argtypes = [vp, vp, ci, C.POINTER(ci)]
    L.bioem_hip_topk_angles.argtypes = [vp, ci, C.c_double, vp]
    L.bioem_hip_merge_topk_host.argtypes = [ci, ci, ci, C.POINTER(vp), vp]
    L.bioem_hip_merge.argtypes = [C.POINTER(vp), ci, vp, ci, C.c_double, vp]
    L.bioem_hip_merge_host.argtypes = [ci, ci, ci, ci, C.POINTER(vp), vp]
    L.bioem_hip_debug_projection.argtypes = [vp, ci, vp]
    L.bioem_hip_debug_convolution.argtypes = [vp, ci, ci, vp, C.POINTER(cf), C.POINTER(cf)]
    L.bioem_hip_debug_particles.argtypes = [vp, vp, vp, vp]
    L.bioem_hip_kernel_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.bioem_hip_reset_kernel_stats.argtypes = [vp]
    L.bioem_hip_uses_fast_path.argtypes = [vp]
    L.bioem_hip_kernel_name.argtypes = [vp]
    L.bioem_hip_kernel_name.restype = C.c_char_p
    L.bioem_hip_kernel_signature.argtypes = [vp]
    L.bioem_hip_kernel_signature.restype = C.c_char_p
    L.bioem_hip_plan.argtypes = [ci, ci, ci, ci, C.c_char_p, ci]
    L.bioem_hip_synchronize.argtypes = [vp]
    L.bioem_hip_r2c.argtypes = [ci, ci, ci, vp, vp]
    _lib = L
    return L


EXPORTS = ["bioem_hip_device_count", "bioem_hip_create", "bioem_hip_create_shard", "bioem_hip_destroy",
           "bioem_hip_last_error", "bioem_hip_project_convolve_compare_ctf", "bioem_hip_topk_angles",
           "bioem_hip_merge_topk_host", "bioem_hip_merge",
           "bioem_hip_upload_particles", "bioem_hip_upload_particle_maps", "bioem_hip_upload_ctf",
           "bioem_hip_upload_model", "bioem_hip_upload_orientations", "bioem_hip_host_alloc",
           "bioem_hip_host_free", "bioem_hip_prob_size", "bioem_hip_start_run", "bioem_hip_compare",
           "bioem_hip_project_convolve_compare", "bioem_hip_finish_run", "bioem_hip_merge_host",
           "bioem_hip_debug_projection", "bioem_hip_debug_convolution", "bioem_hip_debug_particles",
           "bioem_hip_kernel_stats", "bioem_hip_reset_kernel_stats", "bioem_hip_uses_fast_path",
           "bioem_hip_kernel_name", "bioem_hip_kernel_signature", "bioem_hip_plan",
           "bioem_hip_synchronize", "bioem_hip_r2c", "bioem_hip_project", "bioem_hip_convolve",
           "bioem_hip_compare_device", "bioem_hip_max_batch", "bioem_hip_set_phase_timing", "bioem_hip_phase_records"]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def new_prob_block(nMaps, nAngles, writeAngles):
    """Initialised probability block as run() does (reference bioem.cpp:681-699): returns (raw uint8 array,
    pmap view, pang view or None)."""
    nbytes = nMaps * 40 + (nMaps * nAngles * 16 if writeAngles else 0)
    raw = np.zeros(nbytes, dtype=np.uint8)
    pmap = raw[:nMaps * 40].view(PROB_MAP_DTYPE)
    pmap["Total"] = 0.0
    pmap["Constoadd"] = MIN_PROB
    pang = None
    if writeAngles:
        pang = raw[nMaps * 40:].view(PROB_ANGLE_DTYPE).reshape(nAngles, nMaps)
        pang["forAngles"] = 0.0
        pang["ConstAngle"] = MIN_PROB
    return raw, pmap, pang


class Engine:
    """One GPU's comparison engine; method names mirror the plugin hooks of the reference
    (deviceInit / deviceStartRun / compareRefMaps / deviceFinishRun, include/bioem.h:52-79)."""

    def __init__(self, pd, nMaps, nAngles, nCTF, algo=1, device=0, shard=None):
        """shard = (iOrientBegin, iOrientEnd): bioem_hip_create_shard -- the handle owns that block of the nAngles
        global orientations, keeps its angle table on the device and moves only the map entries in start/finish_run."""
        self.L = load_library()
        self.h = C.c_void_p()
        self.pd = pd
        self.nMaps, self.nAngles, self.nCTF, self.algo = nMaps, nAngles, nCTF, algo
        self.N = pd.NumberPixels
        self.H = self.N // 2 + 1
        self.shard = shard
        if shard is None:
            rc = self.L.bioem_hip_create(C.byref(self.h), device, C.byref(pd), nMaps, nAngles, nCTF, algo)
        else:
            rc = self.L.bioem_hip_create_shard(C.byref(self.h), device, C.byref(pd), nMaps, nAngles, nCTF, algo,
                                               int(shard[0]), int(shard[1]))
        if rc:
            msg = self.L.bioem_hip_last_error(self.h).decode() if self.h else "create failed"
            raise RuntimeError("bioem_hip_create: " + msg)

    def _chk(self, rc, what):
        if rc:
            raise RuntimeError("%s: %s" % (what, self.L.bioem_hip_last_error(self.h).decode()))

    def close(self):
        if self.h:
            self.L.bioem_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def fast_path(self):
        return bool(self.L.bioem_hip_uses_fast_path(self.h))

    @property
    def kernel_name(self):
        return self.L.bioem_hip_kernel_name(self.h).decode()

    @property
    def kernel_signature(self):
        return self.L.bioem_hip_kernel_signature(self.h).decode()

    def upload_particles(self, refFFT, sumRef, sumsqRef):
        refFFT = np.ascontiguousarray(refFFT, dtype=np.float32)
        assert refFFT.shape == (self.nMaps, self.N, self.H, 2)
        s = np.ascontiguousarray(sumRef, dtype=np.float32)
        s2 = np.ascontiguousarray(sumsqRef, dtype=np.float32)
        self._chk(self.L.bioem_hip_upload_particles(self.h, _p(refFFT), _p(s), _p(s2)), "upload_particles")

    def upload_particle_maps(self, maps):
        maps = np.ascontiguousarray(maps, dtype=np.float32)
        assert maps.shape == (self.nMaps, self.N, self.N)
        self._chk(self.L.bioem_hip_upload_particle_maps(self.h, _p(maps)), "upload_particle_maps")

    def upload_ctf(self, refCTF, ctfParam):
        refCTF = np.ascontiguousarray(refCTF, dtype=np.float32)
        ctfParam = np.ascontiguousarray(ctfParam, dtype=np.float32)
        assert refCTF.shape == (self.nCTF, self.N, self.H, 2) and ctfParam.shape == (self.nCTF, 3)
        self._chk(self.L.bioem_hip_upload_ctf(self.h, _p(refCTF), _p(ctfParam)), "upload_ctf")

    def upload_model(self, points, NormDen, pixelSize, shiftX=0, shiftY=0):
        points = np.ascontiguousarray(points)
        assert points.dtype.itemsize == 24
        self._chk(self.L.bioem_hip_upload_model(self.h, _p(points), len(points), float(NormDen), float(pixelSize),
                                                int(shiftX), int(shiftY)), "upload_model")

    def upload_orientations(self, angles, isQuat):
        angles = np.ascontiguousarray(angles, dtype=np.float32)
        assert angles.ndim == 2 and angles.shape[1] == 4
        self._chk(self.L.bioem_hip_upload_orientations(self.h, _p(angles), len(angles), int(bool(isQuat))),
                  "upload_orientations")

    def prob_bytes(self):
        """bytes start_run / finish_run move: the whole block, or only the map entries for a shard handle"""
        if self.shard is not None:
            return self.L.bioem_hip_prob_size(self.nMaps, 0, 0)
        return self.L.bioem_hip_prob_size(self.nMaps, self.nAngles, self.pd.writeAngles)

    def start_run(self, raw):
        assert raw.nbytes == self.prob_bytes()
        self._chk(self.L.bioem_hip_start_run(self.h, _p(raw)), "start_run")

    def compare(self, iPipeline, iOrient, iConvStart, maxParallelConv, nTotParallelConv, conv_base, params_base):
        """== bioem::compareRefMaps; conv_base [2*nTotParallelConv, N, H, 2], params_base [2*nTotParallelConv]."""
        assert conv_base.dtype == np.float32 and conv_base.flags["C_CONTIGUOUS"]
        assert params_base.dtype == PARAM5_DTYPE and params_base.flags["C_CONTIGUOUS"]
        self._chk(self.L.bioem_hip_compare(self.h, iPipeline, iOrient, iConvStart, maxParallelConv, nTotParallelConv,
                                           _p(conv_base), _p(params_base)), "compare")

    def project_convolve_compare(self, o0, o1):
        self._chk(self.L.bioem_hip_project_convolve_compare(self.h, o0, o1), "project_convolve_compare")

    def project_convolve_compare_ctf(self, o0, o1, c0, c1):
        self._chk(self.L.bioem_hip_project_convolve_compare_ctf(self.h, o0, o1, c0, c1), "project_convolve_compare_ctf")

    def project(self, iPipeline, o0, o1):
        """== bioem::createProjection for [o0, o1), asynchronous; the spectra stay in buffer set iPipeline & 1"""
        self._chk(self.L.bioem_hip_project(self.h, iPipeline, o0, o1), "project")

    def convolve(self, iPipeline, c0, c1):
        """== bioem::createConvolutedProjectionMap for the projections of the set x CTFs [c0, c1), asynchronous"""
        self._chk(self.L.bioem_hip_convolve(self.h, iPipeline, c0, c1), "convolve")

    def compare_device(self, iPipeline):
        """== bioem::compareRefMaps for the conv spectra of the set, asynchronous"""
        self._chk(self.L.bioem_hip_compare_device(self.h, iPipeline), "compare_device")

    def max_batch(self):
        a, b = C.c_int(), C.c_int()
        self._chk(self.L.bioem_hip_max_batch(self.h, C.byref(a), C.byref(b)), "max_batch")
        return a.value, b.value

    def set_phase_timing(self, on):
        self._chk(self.L.bioem_hip_set_phase_timing(self.h, int(bool(on))), "set_phase_timing")

    def phase_records(self):
        """per-batch device time of projection (0) / convolution (1) / comparison (2) since set_phase_timing(True)"""
        n = C.c_int()
        self._chk(self.L.bioem_hip_phase_records(self.h, None, 0, C.byref(n)), "phase_records")
        out = np.zeros(n.value, dtype=PHASE_RECORD_DTYPE)
        if n.value:
            self._chk(self.L.bioem_hip_phase_records(self.h, _p(out), n.value, C.byref(n)), "phase_records")
        return out

    def finish_run(self, raw):
        assert raw.nbytes == self.prob_bytes()
        self._chk(self.L.bioem_hip_finish_run(self.h, _p(raw)), "finish_run")

    def topk_angles(self, K, numconst):
        """K best orientations per particle among the owned ones, selected on the device: [nMaps, K] CANDIDATE_DTYPE"""
        out = np.zeros((self.nMaps, K), dtype=CANDIDATE_DTYPE)
        self._chk(self.L.bioem_hip_topk_angles(self.h, K, float(numconst), _p(out)), "topk_angles")
        return out

    def synchronize(self):
        self._chk(self.L.bioem_hip_synchronize(self.h), "synchronize")

    def debug_projection(self, iOrient):
        out = np.empty((self.N, self.H, 2), dtype=np.float32)
        self._chk(self.L.bioem_hip_debug_projection(self.h, iOrient, _p(out)), "debug_projection")
        return out

    def debug_convolution(self, iOrient, iConv):
        out = np.empty((self.N, self.H, 2), dtype=np.float32)
        s, s2 = C.c_float(), C.c_float()
        self._chk(self.L.bioem_hip_debug_convolution(self.h, iOrient, iConv, _p(out), C.byref(s), C.byref(s2)),
                  "debug_convolution")
        return out, np.float32(s.value), np.float32(s2.value)

    def debug_particles(self):
        spec = np.empty((self.nMaps, self.N, self.H, 2), dtype=np.float32)
        s = np.empty(self.nMaps, dtype=np.float32)
        s2 = np.empty(self.nMaps, dtype=np.float32)
        self._chk(self.L.bioem_hip_debug_particles(self.h, _p(spec), _p(s), _p(s2)), "debug_particles")
        return spec, s, s2

    def kernel_stats(self):
        ms, n, c = C.c_double(), C.c_longlong(), C.c_longlong()
        self._chk(self.L.bioem_hip_kernel_stats(self.h, C.byref(ms), C.byref(n), C.byref(c)), "kernel_stats")
        return ms.value, n.value, c.value

    def reset_kernel_stats(self):
        self._chk(self.L.bioem_hip_reset_kernel_stats(self.h), "reset_kernel_stats")


def r2c(images, device=0):
    """fftwf_plan_dft_r2c_2d of a stack of square float images on the device (bioem_hip_r2c): [n, N, N] -> complex64
    [n, N, N // 2 + 1], the kernels the projections and the particle maps go through."""
    L = load_library()
    images = np.ascontiguousarray(images, dtype=np.float32)
    n, N, _ = images.shape
    out = np.empty((n, N, N // 2 + 1), dtype=np.complex64)
    if L.bioem_hip_r2c(device, N, n, _p(images), _p(out)):
        raise RuntimeError("bioem_hip_r2c failed (N = %d, %d images)" % (N, n))
    return out


def merge_topk_host(cands):
    """K-way merge of per-shard candidate lists ([nMaps, K] each, shards in ascending orientation-block order)."""
    L = load_library()
    nMaps, K = cands[0].shape
    cands = [np.ascontiguousarray(c, dtype=CANDIDATE_DTYPE) for c in cands]
    out = np.zeros((nMaps, K), dtype=CANDIDATE_DTYPE)
    arr = (C.c_void_p * len(cands))(*[c.ctypes.data for c in cands])
    if L.bioem_hip_merge_topk_host(len(cands), nMaps, K, arr, _p(out)):
        raise RuntimeError("bioem_hip_merge_topk_host failed")
    return out


def merge_rccl(engines, K=0, numconst=0.0):
    """bioem_hip_merge: RCCL all-gather + device fold of the engines' shards (one GPU per engine, this process).
    Returns (pmap [nMaps], candidates [nMaps, K] or None)."""
    L = load_library()
    nMaps = engines[0].nMaps
    pmap = np.zeros(nMaps, dtype=PROB_MAP_DTYPE)
    cand = np.zeros((nMaps, K), dtype=CANDIDATE_DTYPE) if K > 0 else None
    arr = (C.c_void_p * len(engines))(*[e.h.value for e in engines])
    rc = L.bioem_hip_merge(arr, len(engines), _p(pmap), K, float(numconst), _p(cand))
    if rc:
        raise RuntimeError("bioem_hip_merge: " + L.bioem_hip_last_error(engines[0].h).decode())
    return pmap, cand


def merge_host(blocks, nMaps, nAngles, writeAngles):
    """Host log-sum-exp merge of shard probability blocks (reference bioem.cpp:909-1044)."""
    L = load_library()
    out = np.zeros_like(blocks[0])
    arr = (C.c_void_p * len(blocks))(*[b.ctypes.data for b in blocks])
    rc = L.bioem_hip_merge_host(len(blocks), nMaps, nAngles, int(writeAngles), arr, _p(out))
    if rc:
        raise RuntimeError("bioem_hip_merge_host failed")
    return out

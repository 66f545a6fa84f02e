"""ctypes front-end of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (bioem_amd/) never does.

Besides binding bioem_oracle.c it restates, in numpy float32/float64 arithmetic, the host-side
set-up of the reference that feeds the hot path:
  * parameter-file defaults and unit conversions ........ /root/reference/param.cpp:64-627
  * orientation lists / Euler grid / quaternion grid ..... /root/reference/param.cpp:988-1334
  * model NormDen + centre of mass ....................... /root/reference/model.cpp:419-672
  * Output_Probabilities / ANG_PROB text ................. /root/reference/bioem.cpp:1047-1374
"""
import ctypes as C
import heapq
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


class ParamDevice(C.Structure):
    _fields_ = [("maxDisplaceCenter", C.c_int), ("GridSpaceCenter", C.c_int), ("NumberPixels", C.c_int),
                ("NumberFFTPixels1D", C.c_int), ("NxDisp", C.c_int), ("NtotDisp", C.c_int),
                ("Ntotpi", C.c_float), ("volu", C.c_float), ("sigmaPriorbctf", C.c_float),
                ("sigmaPriordefo", C.c_float), ("Priordefcent", C.c_float), ("sigmaPrioramp", C.c_float),
                ("Priorampcent", C.c_float), ("writeAngles", C.c_int), ("tousepsf", C.c_int)]


class CtfGrid(C.Structure):
    _fields_ = [("startAmp", C.c_float), ("endAmp", C.c_float), ("nAmp", C.c_int),
                ("startPhase", C.c_float), ("endPhase", C.c_float), ("nPhase", C.c_int),
                ("startEnv", C.c_float), ("endEnv", C.c_float), ("nEnv", C.c_int)]


PROB_MAP_DTYPE = np.dtype([("Total", "<f8"), ("Constoadd", "<f8"), ("cent_x", "<i4"), ("cent_y", "<i4"),
                           ("orient", "<i4"), ("conv", "<i4"), ("norm", "<f4"), ("mu", "<f4")])
PROB_ANGLE_DTYPE = np.dtype([("forAngles", "<f8"), ("ConstAngle", "<f8")])
PARAM5_DTYPE = np.dtype([("amp", "<f4"), ("pha", "<f4"), ("env", "<f4"), ("sumC", "<f4"), ("sumsquareC", "<f4")])
POINT_DTYPE = np.dtype([("pos", "<f4", (3,)), ("quat4", "<f4"), ("radius", "<f4"), ("density", "<f4")])

_lib = None


def cgroup_cpu_limit():
    """CPUs the container's cgroup grants (cpu.max quota / period), or None when unlimited / unknown.  On the GPU box
    the affinity mask shows every host core while the quota is the share of one GPU; running one thread per visible
    core then only makes the threads fight over the quota."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0:
            return max(1, quota // period)
    except (OSError, ValueError):
        pass
    return None


def usable_cpus(cap=16):
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    lim = cgroup_cpu_limit()
    if lim:
        n = min(n, lim)
    env = os.environ.get("OMP_NUM_THREADS")
    if env:
        try:
            n = min(n, int(env))
        except ValueError:
            pass
    return max(1, min(n, cap))


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        vp = C.c_void_p
        L.orc_fft2_r2c.argtypes = [C.c_int, vp, vp]
        L.orc_fft2_c2r.argtypes = [C.c_int, vp, vp]
        L.orc_ctf_kernels.argtypes = [C.c_int, C.c_float, C.c_int, C.POINTER(CtfGrid), vp, vp, vp]
        L.orc_ctf_kernels.restype = C.c_int
        L.orc_volu.argtypes = [C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_float]
        L.orc_volu.restype = C.c_float
        L.orc_center_model.argtypes = [vp, C.c_int, C.c_float]
        L.orc_map_sums.argtypes = [C.c_int, vp, fp, fp]
        L.orc_projection.argtypes = [vp, C.c_int, C.c_float, vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vp,
                                     vp]
        L.orc_projection.restype = C.c_int
        L.orc_convolve.argtypes = [C.c_int, vp, vp, vp, fp, fp]
        L.orc_calc_logpro.argtypes = [C.POINTER(ParamDevice)] + [C.c_float] * 8
        L.orc_calc_logpro.restype = C.c_double
        L.orc_cc_map.argtypes = [C.c_int, vp, vp, vp]
        L.orc_compare.argtypes = [C.POINTER(ParamDevice), C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int,
                                  C.c_int, vp, vp, vp, vp]
        L.orc_init_prob.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp]
        L.orc_run.argtypes = [C.POINTER(ParamDevice), C.c_int, vp, C.c_int, C.c_float, vp, C.c_int, C.c_int,
                              C.c_float, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int,
                              vp, vp]
        L.orc_merge.argtypes = [C.c_int, C.c_int, vp, vp]
        L.orc_final_logp.argtypes = [C.POINTER(ParamDevice), C.c_double, C.c_double]
        L.orc_final_logp.restype = C.c_double
        L.orc_sizeof_prob_map.restype = C.c_int
        assert L.orc_sizeof_prob_map() == PROB_MAP_DTYPE.itemsize == 40
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_get_max_threads.restype = C.c_int
        # never oversubscribe: the GPU box exposes more hardware threads than this job may use
        L.orc_set_num_threads(usable_cpus())
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ------------------------------------------------------------------------------------------------
# thin array-level wrappers
# ------------------------------------------------------------------------------------------------
def fft2_r2c(img):
    N = img.shape[0]
    img = np.ascontiguousarray(img, dtype=f32)
    out = np.empty((N, N // 2 + 1, 2), dtype=f32)
    lib().orc_fft2_r2c(N, _p(img), _p(out))
    return out


def fft2_c2r(spec):
    N = spec.shape[0]
    spec = np.ascontiguousarray(spec, dtype=f32)
    out = np.empty((N, N), dtype=f32)
    lib().orc_fft2_c2r(N, _p(spec), _p(out))
    return out


def map_sums(img):
    s = C.c_float()
    s2 = C.c_float()
    img = np.ascontiguousarray(img, dtype=f32)
    lib().orc_map_sums(img.shape[0], _p(img), C.byref(s), C.byref(s2))
    return f32(s.value), f32(s2.value)


def projection(points, NormDen, angle, isQuat, N, px, shiftX=0, shiftY=0, want_real=False):
    spec = np.empty((N, N // 2 + 1, 2), dtype=f32)
    real = np.empty((N, N), dtype=f32) if want_real else None
    angle = np.ascontiguousarray(angle, dtype=f32)
    lib().orc_projection(_p(points), len(points), f32(NormDen), _p(angle), int(isQuat), N, f32(px), shiftX, shiftY,
                         _p(real), _p(spec))
    return (spec, real) if want_real else spec


def convolve(proj, ctf):
    N = proj.shape[0]
    out = np.empty_like(proj)
    s = C.c_float()
    s2 = C.c_float()
    lib().orc_convolve(N, _p(np.ascontiguousarray(proj)), _p(np.ascontiguousarray(ctf)), _p(out), C.byref(s),
                       C.byref(s2))
    return out, f32(s.value), f32(s2.value)


def cc_map(conv, ref):
    N = conv.shape[0]
    out = np.empty((N, N), dtype=f32)
    lib().orc_cc_map(N, _p(np.ascontiguousarray(conv)), _p(np.ascontiguousarray(ref)), _p(out))
    return out


# ------------------------------------------------------------------------------------------------
# host set-up restated from the reference
# ------------------------------------------------------------------------------------------------
def parse_param_file(path):
    """param.cpp:64-627: defaults, keyword parsing, CTF unit conversions.  Returns a dict."""
    P = dict(usepsf=False, writeCTF=False, elecwavel=f32(0.019866), doquater=False, nocentermass=False,
             notnormmap=False, yespriorAngles=False, ignorePDB=False, priorMod=f32(1), shiftX=0, shiftY=0,
             sigmaPriorbctf=f32(100.), sigmaPriordefo=f32(2.0), Priordefcent=f32(3.0), sigmaPrioramp=f32(0.5),
             Priorampcent=f32(0.), writeAngles=0, GridPointsQuatern=None, angleGridPointsAlpha=None,
             angleGridPointsBeta=None)
    seen = set()
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            tok = [t for t in line.split(" ") if t != ""]
            if not tok or line.startswith("#"):
                continue
            k = tok[0]
            v = tok[1:]
            seen.add(k)
            if k == "PIXEL_SIZE":
                P["pixelSize"] = f32(float(v[0]))
            elif k == "NUMBER_PIXELS":
                P["N"] = int(v[0])
            elif k == "GRIDPOINTS_ALPHA":
                P["angleGridPointsAlpha"] = int(v[0])
            elif k == "GRIDPOINTS_BETA":
                P["angleGridPointsBeta"] = int(v[0])
            elif k == "USE_QUATERNIONS":
                P["doquater"] = True
            elif k == "GRIDPOINTS_QUATERNION":
                P["GridPointsQuatern"] = int(v[0])
                P["doquater"] = True
            elif k == "CTF_B_ENV":
                P["startBfactor"], P["endBfactor"], P["nEnv"] = f32(float(v[0])), f32(float(v[1])), int(v[2])
            elif k == "CTF_DEFOCUS":
                P["startDefocus"], P["endDefocus"], P["nPhase"] = f32(float(v[0])), f32(float(v[1])), int(v[2])
            elif k in ("CTF_AMPLITUDE", "PSF_AMPLITUDE"):
                P["startAmp"], P["endAmp"], P["nAmp"] = f32(float(v[0])), f32(float(v[1])), int(v[2])
            elif k == "ELECTRON_WAVELENGTH":
                P["elecwavel"] = f32(float(v[0]))
            elif k == "USE_PSF":
                P["usepsf"] = True
            elif k == "PSF_ENVELOPE":
                P["startEnv"], P["endEnv"], P["nEnv"] = f32(float(v[0])), f32(float(v[1])), int(v[2])
            elif k == "PSF_PHASE":
                P["startPhase"], P["endPhase"], P["nPhase"] = f32(float(v[0])), f32(float(v[1])), int(v[2])
            elif k == "DISPLACE_CENTER":
                P["maxD"], P["gridSpace"] = int(v[0]), int(v[1])
            elif k == "WRITE_PROB_ANGLES":
                P["writeAngles"] = int(v[0])
            elif k == "NO_CENTEROFMASS":
                P["nocentermass"] = True
            elif k == "NO_MAP_NORM":
                P["notnormmap"] = True
            elif k == "PRIOR_MODEL":
                P["priorMod"] = f32(float(v[0]))
            elif k == "PRIOR_ANGLES":
                P["yespriorAngles"] = True
            elif k == "SHIFT_X":
                P["shiftX"] = int(v[0])
            elif k == "SHIFT_Y":
                P["shiftY"] = int(v[0])
            elif k == "SIGMA_PRIOR_B_CTF":
                P["sigmaPriorbctf"] = f32(float(v[0]))
            elif k == "SIGMA_PRIOR_DEFOCUS":
                P["sigmaPriordefo"] = f32(float(v[0]))
            elif k == "PRIOR_DEFOCUS_CENTER":
                P["Priordefcent"] = f32(float(v[0]))
            elif k == "SIGMA_PRIOR_AMP_CTF":
                P["sigmaPrioramp"] = f32(float(v[0]))
            elif k == "PRIOR_AMP_CTF_CENTER":
                P["Priorampcent"] = f32(float(v[0]))
            elif k == "WRITE_CTF_PARAM":
                P["writeCTF"] = True
            elif k == "IGNORE_PDB":
                P["ignorePDB"] = True
    if not P["usepsf"]:
        # param.cpp:601-607 (double product, stored to float)
        fac = math.pi * float(f32(2.0)) * 10000 * float(P["elecwavel"])
        P["startPhase"] = f32(float(P["startDefocus"]) * math.pi * 2.0 * 10000 * float(P["elecwavel"]))
        P["endPhase"] = f32(float(P["endDefocus"]) * math.pi * 2.0 * 10000 * float(P["elecwavel"]))
        P["startEnv"] = P["startBfactor"]
        P["endEnv"] = P["endBfactor"]
        P["Priordefcent"] = f32(float(P["Priordefcent"]) * fac)
        P["sigmaPriordefo"] = f32(float(P["sigmaPriordefo"]) * fac)
    return P


_libm = C.CDLL("libm.so.6")
_libm.acosf.argtypes = [C.c_float]
_libm.acosf.restype = C.c_float


def _libm_acosf(x):
    return _libm.acosf(float(x))


def orientations(P, orient_lines=None):
    """param.cpp:988-1334.  Returns (angles[n,4] f32, isQuat, voluang f32)."""
    priorMod = P["priorMod"]
    P["angprior"] = None
    if orient_lines is not None and len(orient_lines):
        n = len(orient_lines)
        ang = np.zeros((n, 4), dtype=f32)
        ncol = 4 if P["doquater"] else 3
        pri = np.zeros(n, dtype=f32)
        for i, ln in enumerate(orient_lines):
            ln = str(ln)
            for c in range(ncol):
                ang[i, c] = f32(float(ln[12 * c:12 * c + 12]))
            if P["yespriorAngles"]:   # param.cpp:1098-1106 / 1296-1304: one more 12-char column
                pri[i] = f32(float(ln[12 * ncol:12 * ncol + 12]))
        if P["yespriorAngles"]:
            P["angprior"] = pri
        voluang = f32((1. / float(f32(n))) * float(priorMod))  # :1131,1324 (double expr -> float)
        return ang, P["doquater"], voluang
    if not P["doquater"]:
        na, nb = P["angleGridPointsAlpha"], P["angleGridPointsBeta"]
        grid_alpha = f32(float(f32(2.0)) * math.pi / float(f32(na)))  # :1015 (double expr -> float)
        cos_grid_beta = f32(f32(2.0) / f32(nb))                          # :1018
        ang = np.zeros((na * nb * na, 4), dtype=f32)
        n = 0
        for ia in range(na):
            for ib in range(nb):
                for ig in range(na):
                    # :1031-1039: float*float - double + float -> double -> float
                    ang[n, 0] = f32(float(f32(ia) * grid_alpha) - math.pi + float(grid_alpha * f32(0.5)))
                    carg = f32(f32(ib) * cos_grid_beta) - f32(1) + cos_grid_beta * f32(0.5)
                    ang[n, 1] = f32(_libm_acosf(f32(carg)))  # float overload of acos = libm acosf
                    ang[n, 2] = f32(float(f32(ig) * grid_alpha) - math.pi + float(grid_alpha * f32(0.5)))
                    n += 1
        # :1046-1047
        voluang = f32(float(grid_alpha * grid_alpha * cos_grid_beta) / (2.0 * math.pi) / (2.0 * math.pi) / 2.0 *
                      float(priorMod))
        return ang, False, voluang
    nq = P["GridPointsQuatern"]
    dg = f32(f32(2.0) / f32(nq + 1))
    rows = []
    for a in range(nq + 1):
        q1 = f32(float(f32(a) * dg - f32(1.0)) + 0.5 * float(dg))
        for b in range(nq + 1):
            q2 = f32(float(f32(b) * dg - f32(1.0)) + 0.5 * float(dg))
            for c in range(nq + 1):
                q3 = f32(float(f32(c) * dg - f32(1.0)) + 0.5 * float(dg))
                if f32(f32(q1 * q1 + q2 * q2) + q3 * q3) <= f32(1.0):
                    w = f32(np.sqrt(f32(f32(f32(f32(1.0) - q1 * q1) - q2 * q2) - q3 * q3)))
                    rows.append([q1, q2, q3, w])
                    rows.append([q1, q2, q3, -w])
    ang = np.array(rows, dtype=f32)
    voluang = f32(dg * dg * dg * priorMod)
    return ang, True, voluang


def model_from_array(arr, nocentermass=False):
    """model.cpp:419-601 (+604-672): float32 points, float NormDen (sequential sum), centre of mass removed."""
    arr = np.asarray(arr, dtype=np.float64)
    pts = np.zeros(len(arr), dtype=POINT_DTYPE)
    pts["pos"] = arr[:, :3].astype(f32)
    pts["radius"] = arr[:, 3].astype(f32)
    pts["density"] = arr[:, 4].astype(f32)
    nd = f32(0.0)
    for d in pts["density"]:
        nd = f32(nd + d)
    if not nocentermass:
        lib().orc_center_model(_p(pts), len(pts), nd)
    return pts, nd


class Setup:
    """Everything configure()/precalculate() prepares (bioem.cpp:438-622) for a given input set."""

    def __init__(self, P, model_arr, maps, orient_lines=None, debug_break=None):
        L = lib()
        self.P = P
        N = P["N"]
        self.N = N
        self.H = N // 2 + 1
        self.px = P["pixelSize"]
        self.points, self.NormDen = model_from_array(model_arr, P["nocentermass"])
        self.angles, self.isQuat, self.voluang = orientations(P, orient_lines)
        self.nAngles = len(self.angles)
        g = CtfGrid(P["startAmp"], P["endAmp"], P["nAmp"], P["startPhase"], P["endPhase"], P["nPhase"],
                    P["startEnv"], P["endEnv"], P["nEnv"])
        self.nCTF = P["nAmp"] * P["nPhase"] * P["nEnv"]
        self.refCTF = np.zeros((self.nCTF, N, self.H, 2), dtype=f32)
        self.ctfParam = np.zeros((self.nCTF, 3), dtype=f32)
        steps = np.zeros(3, dtype=f32)
        n = L.orc_ctf_kernels(N, self.px, int(P["usepsf"]), C.byref(g), _p(self.refCTF), _p(self.ctfParam),
                              _p(steps))
        assert n == self.nCTF
        self.steps = steps
        pd = ParamDevice()
        pd.maxDisplaceCenter = P["maxD"]
        pd.GridSpaceCenter = P["gridSpace"]
        pd.NumberPixels = N
        pd.NumberFFTPixels1D = self.H
        pd.NxDisp = 2 * (P["maxD"] // P["gridSpace"]) + 1
        pd.NtotDisp = pd.NxDisp * pd.NxDisp
        pd.Ntotpi = f32(N * N)
        pd.sigmaPriorbctf = P["sigmaPriorbctf"]
        pd.sigmaPriordefo = P["sigmaPriordefo"]
        pd.Priordefcent = P["Priordefcent"]
        pd.sigmaPrioramp = P["sigmaPrioramp"]
        pd.Priorampcent = P["Priorampcent"]
        pd.writeAngles = P["writeAngles"]
        pd.tousepsf = int(P["usepsf"])
        pd.volu = L.orc_volu(self.voluang, P["gridSpace"], P["maxD"], self.px, P["nAmp"], steps[2], steps[1],
                             pd.sigmaPriorbctf, pd.sigmaPriordefo, pd.sigmaPrioramp)
        self.pd = pd
        if debug_break is not None:
            # BIOEM_DEBUG_BREAK (bioem.cpp:518-525): counts truncated AFTER volu was computed with the full ones
            if self.nAngles > debug_break:
                self.nAngles = debug_break
                self.angles = np.ascontiguousarray(self.angles[:debug_break])
            if self.nCTF > debug_break:
                self.nCTF = debug_break
                self.refCTF = np.ascontiguousarray(self.refCTF[:debug_break])
                self.ctfParam = np.ascontiguousarray(self.ctfParam[:debug_break])
        maps = np.ascontiguousarray(maps, dtype=f32)
        self.nMaps = len(maps)
        self.maps = maps
        self.sumRef = np.zeros(self.nMaps, dtype=f32)
        self.sumsqRef = np.zeros(self.nMaps, dtype=f32)
        self.refFFT = np.zeros((self.nMaps, N, self.H, 2), dtype=f32)
        for i in range(self.nMaps):
            self.sumRef[i], self.sumsqRef[i] = map_sums(maps[i])
            self.refFFT[i] = fft2_r2c(maps[i])

    def new_prob(self):
        pmap = np.zeros(self.nMaps, dtype=PROB_MAP_DTYPE)
        pang = np.zeros((self.nAngles, self.nMaps), dtype=PROB_ANGLE_DTYPE) if self.pd.writeAngles else None
        lib().orc_init_prob(self.nMaps, self.nAngles, self.pd.writeAngles, _p(pmap), _p(pang))
        return pmap, pang

    def run(self, algo=1, o0=0, o1=None, pmap=None, pang=None):
        if o1 is None:
            o1 = self.nAngles
        if pmap is None:
            pmap, pang = self.new_prob()
        lib().orc_run(C.byref(self.pd), algo, _p(self.points), len(self.points), self.NormDen, _p(self.angles),
                      self.nAngles, int(self.isQuat), self.px, self.P["shiftX"], self.P["shiftY"], self.nCTF,
                      _p(self.refCTF), _p(self.ctfParam), self.nMaps, _p(self.refFFT), _p(self.sumRef),
                      _p(self.sumsqRef), o0, o1, _p(pmap), _p(pang))
        return pmap, pang

    def conv_spectra(self, iOrient):
        """Projection + all CTF convolutions of one orientation: (conv[nCTF,N,H,2], params[nCTF])."""
        proj = projection(self.points, self.NormDen, self.angles[iOrient], self.isQuat, self.N, self.px,
                          self.P["shiftX"], self.P["shiftY"])
        conv = np.empty((self.nCTF, self.N, self.H, 2), dtype=f32)
        p5 = np.zeros(self.nCTF, dtype=PARAM5_DTYPE)
        for c in range(self.nCTF):
            conv[c], s, s2 = convolve(proj, self.refCTF[c])
            p5[c] = (self.ctfParam[c, 0], self.ctfParam[c, 1], self.ctfParam[c, 2], s, s2)
        return conv, p5

    def compare(self, algo, iOrient, iConvStart, conv, p5, pmap, pang=None):
        nConv = len(conv)
        lib().orc_compare(C.byref(self.pd), algo, self.nMaps, self.nAngles, _p(self.refFFT), _p(self.sumRef),
                          _p(self.sumsqRef), iOrient, iConvStart, nConv, _p(np.ascontiguousarray(conv)),
                          _p(np.ascontiguousarray(p5)), _p(pmap), _p(pang))

    def final_logp(self, pm):
        return lib().orc_final_logp(C.byref(self.pd), float(pm["Total"]), float(pm["Constoadd"]))


def mrc_reader_maps(stack, notnormmap=False):
    """What the reference's MRC particle reader delivers for a mode-2 stack given in FILE order
    [section][row][column] (map.cpp:808-845): stored transposed, z-scored with float accumulators in file order."""
    stack = np.asarray(stack, dtype=f32)
    ns, nr, nc = stack.shape
    out = np.zeros((ns, nc, nr), dtype=f32)
    for s_ in range(ns):
        st = f32(0)
        st2 = f32(0)
        for v in stack[s_].ravel():
            st = f32(st + v)
            st2 = f32(st2 + f32(v * v))
        m = np.ascontiguousarray(stack[s_].T)
        if not notnormmap:
            st = f32(st / f32(nr * nc))
            sd = f32(np.sqrt(f32(f32(st2 / f32(nr * nc)) - f32(st * st))))
            m = (m / sd - f32(st / sd)).astype(f32)
        out[s_] = m
    return out


def merge(shards):
    shards = np.ascontiguousarray(np.stack(shards))
    out = np.zeros(shards.shape[1], dtype=PROB_MAP_DTYPE)
    lib().orc_merge(shards.shape[0], shards.shape[1], _p(shards), _p(out))
    return out


def logp_constant(pd):
    """0.5 log(pi) + (1 - Np/2)(log 2pi + 1) + log(volu)   (bioem.cpp:1146-1149)."""
    return 0.5 * math.log(math.pi) + (1 - float(f32(pd.Ntotpi)) * 0.5) * (math.log(2 * math.pi) + 1) + math.log(
        float(f32(pd.volu)))


def format_output_probabilities(S, pmap):
    """bioem.cpp:1077-1222 (fixed, 4 decimals).  Returns the file text."""
    P = S.P
    f4 = lambda x: "%.4f" % float(x)  # noqa: E731
    o = []
    bar = "************************* HEADER:: NOTATION *******************************************\n"
    o.append(bar)
    o.append("Notation= RefMap:  MapNumber ; LogProb natural logarithm of posterior Probability ; Constant: "
             "Numerical Const. for adding Probabilities \n")
    ang = "alpha[rad] - beta[rad] - gamma[rad]" if not S.isQuat else "q1 - q2 - q3 - q4"
    if P["usepsf"]:
        ker = "PSF amp - PSF phase - PSF envelope" if not S.isQuat else "PSF amp - PSF phase - PSF envelope"
        sep = " - " if not S.isQuat else " -"
    else:
        ker = "CTF amp - CTF defocus - CTF B-Env"
        sep = " - "
    o.append("Notation= RefMap:  MapNumber ; Maximizing Param: MaxLogProb - " + ang + sep + ker +
             " - center x - center y - normalization - offsett \n")
    if P["writeCTF"]:
        o.append(" RefMap:  MapNumber ; CTFMaxParm: defocus - b-Env (B ref. Penzeck 2010)\n")
    if P["yespriorAngles"]:
        o.append("**** Remark: Using Prior Proability in Angles ****\n")
    o.append(bar + "\n")
    for i, pm in enumerate(pmap):
        if pm["Total"] > 1.e-38:
            lp = S.final_logp(pm)
            o.append("RefMap: %d LogProb:  %s Constant: %s\n" % (i, f4(lp), f4(pm["Constoadd"])))
            o.append("RefMap: %d Maximizing Param: %s " % (i, f4(lp)))
        else:
            o.append("Warning - RefMap: %dNumerical Integrated Probability without constant = 0.0;\n" % i)
            o.append("Warning - RefMap: %dCheck that constant is finite: %s\n" % (i, f4(pm["Constoadd"])))
            o.append("Warning - RefMap: i) check model, ii) check refmap , iii) check GPU on/off command "
                     "inconsitency\n")
        a = S.angles[pm["orient"]]
        s = "%s [] %s [] %s [] " % (f4(a[0]), f4(a[1]), f4(a[2]))
        if S.isQuat:
            s += "%s [] " % f4(a[3])
        c = S.ctfParam[pm["conv"]]
        s += "%s [] " % f4(c[0])
        if not P["usepsf"]:
            # float / 2.f / double / float * double  (bioem.cpp:1199-1200)
            defo = float(f32(c[1]) / f32(2.0)) / math.pi / float(P["elecwavel"]) * 0.0001
            s += "%s [micro-m] %s [A²] " % (f4(defo), f4(c[2]))
        else:
            s += "%s [1/A²] %s [1/A²] " % (f4(c[1]), f4(c[2]))
        s += "%d [pix] %d [pix] %s [] %s [] \n" % (pm["cent_x"], pm["cent_y"], f4(pm["norm"]), f4(pm["mu"]))
        o.append(s)
        if P["writeCTF"] and P["usepsf"]:
            # bioem.cpp:1225-1242
            denomi = f32(f32(c[1] * c[1]) + f32(c[2] * c[2]))
            v1 = 2 * math.pi * float(c[1]) / float(denomi) / float(P["elecwavel"]) * 0.0001
            v2 = 4 * math.pi * math.pi * float(c[2]) / float(denomi)
            o.append("RefMap: %d CTFMaxParam: %s [micro-m] %s [A²] \n" % (i, f4(v1), f4(v2)))
    return "".join(o)


def ang_prob_rows(S, pmap, pang):
    """bioem.cpp:1245-1365: per map the K best orientations (min-heap on (logp, iOrient)), best first."""
    K = S.pd.writeAngles
    const = logp_constant(S.pd)
    out = {}
    for m in range(S.nMaps):
        q = []
        for io in range(S.nAngles):
            pa = pang[io, m]
            logp = (math.log(pa["forAngles"]) if pa["forAngles"] > 0 else -math.inf) + pa["ConstAngle"] + const
            if len(q) < K:
                heapq.heappush(q, (logp, io))
            elif q[0][0] < logp:
                heapq.heapreplace(q, (logp, io))
        rows = sorted(q, reverse=True)
        pri = S.P.get("angprior")
        out[m] = [dict(orient=io, logp=lp + (float(pri[io]) if pri is not None else 0.0),
                       logsum=math.log(pang[io, m]["forAngles"]), const=float(pang[io, m]["ConstAngle"]),
                       numconst=const, prior=(float(pri[io]) if pri is not None else None)) for lp, io in rows]
    return out

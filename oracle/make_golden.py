#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE ONLY -- never imported by the product).

The reference (bio-phys/BioEM, /root/reference) ships no tests or fixtures, so its
results for the hot path are pinned by running the REFERENCE ITSELF on small
synthetic inputs and committing inputs + outputs under tests/golden/.

The reference binary is oracle/_ref/bioEM_ref, built by oracle/Makefile from the
unmodified reference sources against the image's own FFTW3-API library (AMD hipFFTW,
/opt/rocm/lib/libhipfftw.so -- it needs a GPU, so the `run` stage executes on the
GPU box through gpurun).  Three stages:

  prepare  (this container)  write input files of every case into oracle/_ref/cases/<case>/
                             (+ inputs.npz).  Needs /root/reference only for the
                             576-orientation quaternion list (data).
  run      (GPU box)         run bioEM_ref for every case/algo, outputs -> gpurun_out/golden/<case>/
  collect  (this container)  copy inputs.npz + input text files that tests need + reference
                             outputs into tests/golden/<case>/

Usage: python oracle/make_golden.py prepare|run|collect
"""
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import io_formats as iof  # noqa: E402

CASES_DIR = os.path.join(HERE, "_ref", "cases")
OUT_DIR = os.path.join(ROOT, "gpurun_out", "golden")
GOLD_DIR = os.path.join(ROOT, "tests", "golden")
QUAT576 = "/root/reference/Quaternions/QUATERNION_LIST_576_Orient"


def quat_rotmat(q):
    """Rotation matrix convention of the reference projection (bioem.cpp:1638-1646)."""
    q0, q1, q2, q3 = [float(v) for v in q]
    R = np.empty((3, 3))
    R[0, 0] = 1 - 2 * q1 * q1 - 2 * q2 * q2
    R[1, 0] = 2 * (q0 * q1 - q2 * q3)
    R[2, 0] = 2 * (q0 * q2 + q1 * q3)
    R[0, 1] = 2 * (q0 * q1 + q2 * q3)
    R[1, 1] = 1 - 2 * q0 * q0 - 2 * q2 * q2
    R[2, 1] = 2 * (q1 * q2 - q0 * q3)
    R[0, 2] = 2 * (q0 * q2 - q1 * q3)
    R[1, 2] = 2 * (q1 * q2 + q0 * q3)
    R[2, 2] = 1 - 2 * q0 * q0 - 2 * q1 * q1
    return R


def synth_model(rng, n, extent, rmin, rmax):
    pts = rng.normal(0.0, extent / 2.5, size=(n, 3))
    r = np.linalg.norm(pts, axis=1)
    pts[r > extent] *= (extent / r[r > extent])[:, None] * 0.95
    rad = rng.uniform(rmin, rmax, size=n)
    den = rng.uniform(40.0, 108.0, size=n)
    m = np.concatenate([pts, rad[:, None], den[:, None]], axis=1)
    return np.round(m, 6)


def synth_pdb_model(rng, n, extent):
    """C-alpha trace: coordinates with the 3 decimals a PDB record holds, residue names drawn from all 20 types;
    radius / electrons from the residue tables (io_formats.RESIDUES, restating model.cpp:738-844)."""
    pts = rng.normal(0.0, extent / 2.5, size=(n, 3))
    r = np.linalg.norm(pts, axis=1)
    pts[r > extent] *= (extent / r[r > extent])[:, None] * 0.95
    pts = np.round(pts, 3)
    names = sorted(iof.RESIDUES)
    resnames = [names[int(k)] for k in rng.integers(0, len(names), size=n)]
    resnames[:len(names)] = names[:n]  # every residue type at least once
    rad = np.array([iof.RESIDUES[x][0] for x in resnames])
    den = np.array([iof.RESIDUES[x][1] for x in resnames])
    return np.concatenate([pts, rad[:, None], den[:, None]], axis=1), resnames


def synth_particles(rng, model, rots, N, px, nP, snr, maxshift):
    """Independent (non-reference) image synthesis: Gaussian-blob projection + noise, z-scored."""
    maps = np.zeros((nP, N, N), dtype=np.float64)
    fx = np.fft.fftfreq(N)
    g = np.exp(-2 * (np.pi ** 2) * (1.3 ** 2) * (fx[:, None] ** 2 + fx[None, :] ** 2))
    for p in range(nP):
        R = rots[(7 * p + 3) % len(rots)]
        xyz = model[:, :3] @ R.T
        i = np.floor(xyz[:, 0] / px + N / 2 + 0.5).astype(int)
        j = np.floor(xyz[:, 1] / px + N / 2 + 0.5).astype(int)
        ok = (i >= 0) & (i < N) & (j >= 0) & (j < N)
        img = np.zeros((N, N))
        np.add.at(img, (i[ok], j[ok]), model[ok, 4])
        img = np.real(np.fft.ifft2(np.fft.fft2(img) * g))
        sx, sy = rng.integers(-maxshift, maxshift + 1, size=2)
        img = np.roll(img, (int(sx), int(sy)), axis=(0, 1))
        img = (img - img.mean()) / img.std()
        img = img * np.sqrt(snr) + rng.normal(size=(N, N))
        img = (img - img.mean()) / img.std()
        maps[p] = img
    return np.round(maps, 8).astype(np.float32)


def euler_rotmat(a, b, g):
    """bioem.cpp:1664-1672."""
    ca, sa, cb, sb, cg, sg = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(g), np.sin(g)
    return np.array([[cg * ca - cb * sa * sg, cg * sa + cb * ca * sg, sg * sb],
                     [-sg * ca - cb * sa * cg, -sg * sa + cb * ca * cg, cg * sb],
                     [sb * sa, -sb * ca, cb]])


def load_quat576_lines():
    with open(QUAT576) as f:
        lines = f.read().split("\n")
    n = int(lines[0][:12])
    return lines[1:1 + n]


# ----------------------------------------------------------------------------------------------
# Case table.  Every case: N, pixel size, particles, model, orientation source, parameter keywords.
# ----------------------------------------------------------------------------------------------
def case_table():
    C = {}
    # G1: 48^2 (= 3 * 16), sphere-splat branch, both algos
    C["g1_n48"] = dict(N=48, px=1.77, nP=3, npts=30, extent=24.0, rad=(2.25, 3.4), orient=("list", 40),
                       kw=[("CTF_B_ENV", [50.0, 250.0, 2]), ("CTF_DEFOCUS", [1.0, 3.0, 2]),
                           ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [4, 2])],
                       algos=[1, 2], snr=0.2, maxshift=3, seed=101)
    # G2: config-1 shape: 128^2, 10 particles, all 576 orientations, 2x2x1 CTF grid, +-10 px grid 1
    C["g2_n128"] = dict(N=128, px=1.77, nP=10, npts=400, extent=60.0, rad=(2.25, 3.4), orient=("list", 576),
                        kw=[("CTF_B_ENV", [2.0, 300.0, 2]), ("CTF_DEFOCUS", [1.0, 4.0, 2]),
                            ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [10, 1])],
                        algos=[1, 2], snr=0.05, maxshift=8, seed=102)
    # G3: 32^2 with the per-displacement DEBUG_PROB trace (separate -DDEBUG_PROB binary)
    C["g3_n32_trace"] = dict(N=32, px=3.0, nP=2, npts=20, extent=25.0, rad=(3.2, 3.4), orient=("list", 4),
                             kw=[("CTF_B_ENV", [50.0, 150.0, 2]), ("CTF_DEFOCUS", [1.5, 1.5, 1]),
                                 ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [3, 1])],
                             algos=[1, 2], snr=0.3, maxshift=2, seed=103, trace=True)
    # G4: WRITE_PROB_ANGLES -> ANG_PROB
    C["g4_n32_angles"] = dict(N=32, px=3.0, nP=3, npts=25, extent=25.0, rad=(3.2, 3.4), orient=("list", 24),
                              kw=[("CTF_B_ENV", [50.0, 150.0, 2]), ("CTF_DEFOCUS", [1.0, 2.0, 2]),
                                  ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [4, 1]),
                                  ("WRITE_PROB_ANGLES", [5])],
                              algos=[1, 2], snr=0.3, maxshift=2, seed=104)
    # G5: PSF mode
    C["g5_n32_psf"] = dict(N=32, px=3.0, nP=2, npts=25, extent=25.0, rad=(3.2, 3.4), orient=("list", 12),
                           kw=[("USE_PSF", []), ("PSF_AMPLITUDE", [0.1, 0.1, 1]), ("PSF_ENVELOPE", [0.02, 0.08, 2]),
                               ("PSF_PHASE", [0.01, 0.05, 2]), ("DISPLACE_CENTER", [3, 1])],
                           algos=[1], snr=0.3, maxshift=2, seed=105)
    # G6: Euler-angle grid, point-splat branch (radius <= pixel size), model shift keywords ignored there
    C["g6_n32_euler"] = dict(N=32, px=3.0, nP=2, npts=25, extent=25.0, rad=(1.5, 2.9), orient=("eulergrid", 4, 3),
                             kw=[("GRIDPOINTS_ALPHA", [4]), ("GRIDPOINTS_BETA", [3]),
                                 ("CTF_B_ENV", [50.0, 150.0, 2]), ("CTF_DEFOCUS", [1.0, 2.0, 2]),
                                 ("CTF_AMPLITUDE", [0.1, 0.3, 2]), ("DISPLACE_CENTER", [3, 1])],
                             algos=[1], snr=0.3, maxshift=2, seed=106)
    # G7: 224^2 (radix-7 size = headline benchmark size), tiny counts
    C["g7_n224"] = dict(N=224, px=1.77, nP=4, npts=600, extent=60.0, rad=(2.25, 3.4), orient=("list", 8),
                        kw=[("CTF_B_ENV", [2.0, 300.0, 2]), ("CTF_DEFOCUS", [2.0, 2.0, 1]),
                            ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [10, 1])],
                        algos=[1], snr=0.05, maxshift=8, seed=107)
    # G8: maxD % grid != 0 -> ALGO 1 and ALGO 2 visit different displacement sets
    C["g8_n32_grid"] = dict(N=32, px=3.0, nP=2, npts=25, extent=25.0, rad=(3.2, 3.4), orient=("list", 10),
                            kw=[("CTF_B_ENV", [50.0, 150.0, 2]), ("CTF_DEFOCUS", [1.0, 2.0, 2]),
                                ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [5, 2])],
                            algos=[1, 2], snr=0.3, maxshift=2, seed=108)
    # G9: odd N, priors + shifts + prior model keywords, quaternion grid
    C["g9_n35_odd"] = dict(N=35, px=2.5, nP=2, npts=25, extent=22.0, rad=(2.6, 3.4), orient=("quatgrid", 3),
                           kw=[("GRIDPOINTS_QUATERNION", [3]),
                               ("CTF_B_ENV", [50.0, 150.0, 2]), ("CTF_DEFOCUS", [1.0, 2.0, 2]),
                               ("CTF_AMPLITUDE", [0.1, 0.2, 2]), ("DISPLACE_CENTER", [3, 1]),
                               ("SHIFT_X", [1]), ("SHIFT_Y", [-2]), ("PRIOR_MODEL", [0.5]),
                               ("SIGMA_PRIOR_B_CTF", [80.0]), ("SIGMA_PRIOR_DEFOCUS", [1.5]),
                               ("PRIOR_DEFOCUS_CENTER", [2.0]), ("SIGMA_PRIOR_AMP_CTF", [0.4]),
                               ("PRIOR_AMP_CTF_CENTER", [0.1])],
                           algos=[1], snr=0.3, maxshift=2, seed=109)
    # G10: 64^2 (multiple of 32 -> fast device path), 6 particles, 64 orientations
    C["g10_n64"] = dict(N=64, px=2.2, nP=6, npts=120, extent=40.0, rad=(2.25, 3.4), orient=("list", 64),
                        kw=[("CTF_B_ENV", [20.0, 200.0, 3]), ("CTF_DEFOCUS", [1.0, 3.0, 2]),
                            ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [6, 1])],
                        algos=[1, 2], snr=0.1, maxshift=4, seed=110)
    # G11: Euler-angle LIST with per-orientation priors (PRIOR_ANGLES) and WRITE_PROB_ANGLES
    C["g11_n32_eulerlist"] = dict(N=32, px=3.0, nP=2, npts=25, extent=25.0, rad=(3.2, 3.4), orient=("eulerlist", 14),
                                  kw=[("PRIOR_ANGLES", []), ("CTF_B_ENV", [50.0, 150.0, 2]),
                                      ("CTF_DEFOCUS", [1.0, 2.0, 2]), ("CTF_AMPLITUDE", [0.1, 0.1, 1]),
                                      ("DISPLACE_CENTER", [3, 1]), ("WRITE_PROB_ANGLES", [4])],
                                  algos=[1], snr=0.3, maxshift=2, seed=111)
    # G12: NO_CENTEROFMASS, ELECTRON_WAVELENGTH, 3-point amplitude grid, BIOEM_DEBUG_BREAK truncation
    C["g12_n32_misc"] = dict(N=32, px=3.0, nP=3, npts=25, extent=20.0, rad=(3.2, 3.4), orient=("list", 16),
                             kw=[("NO_CENTEROFMASS", []), ("ELECTRON_WAVELENGTH", [0.0251]),
                                 ("CTF_B_ENV", [50.0, 150.0, 2]), ("CTF_DEFOCUS", [1.0, 2.0, 2]),
                                 ("CTF_AMPLITUDE", [0.1, 0.4, 3]), ("DISPLACE_CENTER", [4, 2])],
                             algos=[1, 2], snr=0.3, maxshift=2, seed=112, env={"BIOEM_DEBUG_BREAK": "9"})
    # G13: PSF mode with WRITE_CTF_PARAM
    C["g13_n32_psf_writectf"] = dict(N=32, px=3.0, nP=2, npts=25, extent=25.0, rad=(3.2, 3.4), orient=("list", 8),
                                     kw=[("USE_PSF", []), ("WRITE_CTF_PARAM", []), ("PSF_AMPLITUDE", [0.1, 0.3, 2]),
                                         ("PSF_ENVELOPE", [0.02, 0.08, 2]), ("PSF_PHASE", [0.01, 0.05, 2]),
                                         ("DISPLACE_CENTER", [3, 1])],
                                     algos=[1], snr=0.3, maxshift=2, seed=113)
    # G14: particles from an MRC stack (--ReadMRC): transposed store + float z-score of the reference reader
    C["g14_n32_mrc"] = dict(N=32, px=3.0, nP=3, npts=25, extent=25.0, rad=(3.2, 3.4), orient=("list", 10),
                            kw=[("CTF_B_ENV", [50.0, 150.0, 2]), ("CTF_DEFOCUS", [1.0, 2.0, 2]),
                                ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [3, 1])],
                            algos=[1], snr=0.3, maxshift=2, seed=114, particles="mrc")
    # G15: MRC stack with NO_MAP_NORM
    C["g15_n32_mrc_nonorm"] = dict(N=32, px=3.0, nP=2, npts=25, extent=25.0, rad=(3.2, 3.4), orient=("list", 6),
                                   kw=[("NO_MAP_NORM", []), ("CTF_B_ENV", [50.0, 150.0, 2]),
                                       ("CTF_DEFOCUS", [1.5, 1.5, 1]), ("CTF_AMPLITUDE", [0.1, 0.1, 1]),
                                       ("DISPLACE_CENTER", [3, 1])],
                                   algos=[1], snr=0.3, maxshift=2, seed=115, particles="mrc")
    # G16-G20: image sizes that select the other register-FFT lengths of the device kernel (N = N1 * R) and the
    # Nyquist-column split: 40 = 5*8, 36 = 9*4, 50 = 25*2, 200 = 25*8, 256 = 8*32 with N/2 a multiple of 64
    C["g16_n40"] = dict(N=40, px=2.6, nP=3, npts=40, extent=28.0, rad=(2.25, 3.4), orient=("list", 12),
                        kw=[("CTF_B_ENV", [40.0, 200.0, 2]), ("CTF_DEFOCUS", [1.0, 3.0, 2]),
                            ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [6, 1])],
                        algos=[1, 2], snr=0.2, maxshift=4, seed=116)
    C["g17_n36"] = dict(N=36, px=2.8, nP=3, npts=40, extent=26.0, rad=(2.25, 3.4), orient=("list", 10),
                        kw=[("CTF_B_ENV", [40.0, 200.0, 2]), ("CTF_DEFOCUS", [1.0, 3.0, 2]),
                            ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [5, 1])],
                        algos=[1, 2], snr=0.2, maxshift=3, seed=117)
    C["g18_n50"] = dict(N=50, px=2.4, nP=3, npts=50, extent=30.0, rad=(2.25, 3.4), orient=("list", 10),
                        kw=[("CTF_B_ENV", [40.0, 200.0, 2]), ("CTF_DEFOCUS", [1.0, 3.0, 2]),
                            ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [7, 1])],
                        algos=[1, 2], snr=0.2, maxshift=4, seed=118)
    C["g19_n200"] = dict(N=200, px=1.77, nP=3, npts=500, extent=58.0, rad=(2.25, 3.4), orient=("list", 6),
                         kw=[("CTF_B_ENV", [2.0, 300.0, 2]), ("CTF_DEFOCUS", [2.0, 2.0, 1]),
                             ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [10, 1])],
                         algos=[1], snr=0.05, maxshift=8, seed=119)
    C["g20_n256"] = dict(N=256, px=1.77, nP=3, npts=600, extent=62.0, rad=(2.25, 3.4), orient=("list", 6),
                         kw=[("CTF_B_ENV", [2.0, 300.0, 2]), ("CTF_DEFOCUS", [2.0, 2.0, 1]),
                             ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [10, 1])],
                         algos=[1, 2], snr=0.05, maxshift=8, seed=120)
    # G21-G22: wide translation windows (the tutorial's production setting is DISPLACE_CENTER 40 1): more offsets per
    # axis than the device kernel's 31-row window -> tiled launches
    C["g21_n64_wide20"] = dict(N=64, px=2.2, nP=4, npts=100, extent=30.0, rad=(2.25, 3.4), orient=("list", 12),
                               kw=[("CTF_B_ENV", [20.0, 200.0, 2]), ("CTF_DEFOCUS", [1.0, 3.0, 2]),
                                   ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [20, 1])],
                               algos=[1, 2], snr=0.1, maxshift=15, seed=121)
    C["g22_n128_wide40"] = dict(N=128, px=1.77, nP=3, npts=300, extent=45.0, rad=(2.25, 3.4), orient=("list", 8),
                                kw=[("CTF_B_ENV", [2.0, 300.0, 2]), ("CTF_DEFOCUS", [2.0, 2.0, 1]),
                                    ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [40, 1])],
                                algos=[1, 2], snr=0.05, maxshift=30, seed=122)
    # G23: the parameter set the reference's tutorial suggests for production runs (doc/index.rst, Param_ProRun):
    # 4 x 8 CTF grid with Gaussian priors, DISPLACE_CENTER 40 1
    C["g23_n128_tutorial"] = dict(N=128, px=1.77, nP=2, npts=300, extent=45.0, rad=(2.25, 3.4), orient=("list", 6),
                                  kw=[("CTF_B_ENV", [2.0, 300.0, 4]), ("CTF_DEFOCUS", [0.5, 4.5, 8]),
                                      ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("SIGMA_PRIOR_B_CTF", [50.0]),
                                      ("SIGMA_PRIOR_DEFOCUS", [0.4]), ("PRIOR_DEFOCUS_CENTER", [2.8]),
                                      ("DISPLACE_CENTER", [40, 1])],
                                  algos=[1, 2], snr=0.05, maxshift=30, seed=123)
    # G24: model from a PDB file (--ReadPDB): C-alpha records only, residue radius / electron tables
    # (model.cpp:85-329, 738-844); the file also carries non-CA atoms, HETATM, TER and REMARK records to be skipped
    C["g24_n32_pdb"] = dict(N=32, px=3.0, nP=3, npts=40, extent=25.0, rad=None, orient=("list", 12),
                            kw=[("CTF_B_ENV", [50.0, 150.0, 2]), ("CTF_DEFOCUS", [1.0, 2.0, 2]),
                                ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [3, 1])],
                            algos=[1, 2], snr=0.3, maxshift=2, seed=124, model_format="pdb")
    # G25: model from an MRC density map (--ReadModelMRC): one point per voxel, radius 2 px (model.cpp:332-416)
    C["g25_n32_modelmrc"] = dict(N=32, px=3.0, nP=3, npts=0, extent=0.0, rad=None, orient=("list", 10),
                                 kw=[("CTF_B_ENV", [50.0, 150.0, 2]), ("CTF_DEFOCUS", [1.0, 2.0, 2]),
                                     ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [3, 1])],
                                 algos=[1], snr=0.3, maxshift=2, seed=125, model_format="mrc", voxels=(5, 6, 4))
    # G26: particles from several MRC stacks named in a list file (--ReadMRC --ReadMultipleMRC, map.cpp:81-265)
    C["g26_n32_multimrc"] = dict(N=32, px=3.0, nP=5, npts=25, extent=25.0, rad=(3.2, 3.4), orient=("list", 10),
                                 kw=[("CTF_B_ENV", [50.0, 150.0, 2]), ("CTF_DEFOCUS", [1.0, 2.0, 2]),
                                     ("CTF_AMPLITUDE", [0.1, 0.1, 1]), ("DISPLACE_CENTER", [3, 1])],
                                 algos=[1], snr=0.3, maxshift=2, seed=126, particles="multimrc", stacks=(2, 3))
    only = os.environ.get("BIOEM_GOLDEN_ONLY")  # e.g. "g16,g17": restrict every stage to cases with these prefixes
    if only:
        C = {k: v for k, v in C.items() if any(k.startswith(o + "_") for o in only.split(","))}
    return C


def write_mrc_stack(path, data):
    """mode-2 little-endian MRC stack: 1024-byte header (nc, nr, ns, mode, ...), no symmetry bytes."""
    import struct
    ns, nr, nc = data.shape
    hdr = np.zeros(256, dtype="<i4")
    hdr[0:4] = [nc, nr, ns, 2]
    hdr[7:10] = [nc, nr, ns]
    raw = hdr.tobytes()
    raw = raw[:40] + struct.pack("<6f", 100., 100., 100., 90., 90., 90.) + raw[64:]
    with open(path, "wb") as f:
        f.write(raw + data.astype("<f4").tobytes())


def orientation_rots(spec, qlines):
    if spec[0] == "list":
        qs = np.array([[float(ln[c:c + 12]) for c in range(0, 48, 12)] for ln in qlines[:spec[1]]])
        return [quat_rotmat(q) for q in qs]
    if spec[0] == "eulerlist":
        rng = np.random.default_rng(4242)
        ang = np.stack([rng.uniform(-np.pi, np.pi, spec[1]), np.arccos(rng.uniform(-1, 1, spec[1])),
                        rng.uniform(-np.pi, np.pi, spec[1])], axis=1)
        return [euler_rotmat(*a) for a in np.round(ang, 8)]
    if spec[0] == "eulergrid":
        na, nb = spec[1], spec[2]
        rots = []
        for ia in range(na):
            for ib in range(nb):
                for ig in range(na):
                    a = ia * 2 * np.pi / na - np.pi + np.pi / na
                    b = np.arccos(ib * 2.0 / nb - 1 + 1.0 / nb)
                    g = ig * 2 * np.pi / na - np.pi + np.pi / na
                    rots.append(euler_rotmat(a, b, g))
        return rots
    if spec[0] == "quatgrid":
        n = spec[1]
        dq = 2.0 / (n + 1)
        rots = []
        for a in range(n + 1):
            for b in range(n + 1):
                for c in range(n + 1):
                    q = np.array([a, b, c]) * dq - 1 + 0.5 * dq
                    if q @ q <= 1:
                        w = np.sqrt(max(0.0, 1 - q @ q))
                        rots.append(quat_rotmat([q[0], q[1], q[2], w]))
                        rots.append(quat_rotmat([q[0], q[1], q[2], -w]))
        return rots
    raise ValueError(spec)


def prepare():
    qlines = load_quat576_lines()
    os.makedirs(CASES_DIR, exist_ok=True)
    for name, c in case_table().items():
        d = os.path.join(CASES_DIR, name)
        os.makedirs(d, exist_ok=True)
        rng = np.random.default_rng(c["seed"])
        mformat = c.get("model_format", "text")
        if mformat == "pdb":
            model, resnames = synth_pdb_model(rng, c["npts"], c["extent"])
        elif mformat == "mrc":
            vol = np.round(rng.uniform(0.1, 1.0, size=c["voxels"]), 6).astype(np.float32)
            model = iof.mrc_volume_points(vol, c["px"])
        else:
            model = synth_model(rng, c["npts"], c["extent"], *c["rad"])
        rots = orientation_rots(c["orient"], qlines)
        maps = synth_particles(rng, model, rots, c["N"], c["px"], c["nP"], c["snr"], c["maxshift"])
        kw = [("PIXEL_SIZE", [c["px"]]), ("NUMBER_PIXELS", [c["N"]])]
        if c["orient"][0] == "list":
            kw.append(("USE_QUATERNIONS", []))
        kw += c["kw"]
        iof.write_param_file(os.path.join(d, "param.txt"), kw)
        if mformat == "pdb":
            iof.write_pdb_model(os.path.join(d, "model.pdb"), model, resnames)
        elif mformat == "mrc":
            iof.write_mrc_volume(os.path.join(d, "model.mrc"), vol)
        else:
            iof.write_text_model(os.path.join(d, "model.txt"), model)
        pformat = c.get("particles", "text")
        if pformat == "multimrc":
            # several stacks + a list file with one path per line (written by the run stage: absolute paths)
            raw = (2.5 * maps + 6.0).astype(np.float32)
            lo = 0
            for k, cnt in enumerate(c["stacks"]):
                write_mrc_stack(os.path.join(d, "stack%d.mrc" % k), raw[lo:lo + cnt])
                lo += cnt
            assert lo == len(raw)
            maps = raw
        elif pformat == "mrc":
            # raw un-normalised counts in FILE order (section, row, column); the reader transposes and z-scores
            raw = (2.5 * maps + 6.0).astype(np.float32)
            write_mrc_stack(os.path.join(d, "particles.mrc"), raw)
            maps = raw
        else:
            iof.write_text_particles(os.path.join(d, "particles.txt"), maps)
        orient_lines = []
        if c["orient"][0] == "list":
            orient_lines = qlines[:c["orient"][1]]
            with open(os.path.join(d, "orient.txt"), "w") as f:
                f.write("%d\n" % len(orient_lines))
                f.write("\n".join(orient_lines) + "\n")
        if c["orient"][0] == "eulerlist":
            rng2 = np.random.default_rng(4242)
            n = c["orient"][1]
            ang = np.stack([rng2.uniform(-np.pi, np.pi, n), np.arccos(rng2.uniform(-1, 1, n)),
                            rng2.uniform(-np.pi, np.pi, n)], axis=1)
            ang = np.round(ang, 8)
            pri = np.round(np.random.default_rng(77).uniform(0.2, 1.0, n), 8)
            orient_lines = ["".join("%12.8f" % v for v in a) + "%12.8f" % pr for a, pr in zip(ang, pri)]
            with open(os.path.join(d, "orient.txt"), "w") as f:
                f.write("%d\n" % n)
                f.write("\n".join(orient_lines) + "\n")
        np.savez_compressed(os.path.join(d, "inputs.npz"), model=model, maps=maps,
                            orient_lines=np.array(orient_lines), N=c["N"], px=c["px"],
                            algos=np.array(c["algos"]), trace=bool(c.get("trace", False)),
                            particles=pformat, model_format=mformat,
                            stacks=np.array(c.get("stacks", ()), dtype=np.int64), env_keys=np.array(list(c.get("env", {}).keys())),
                            env_vals=np.array(list(c.get("env", {}).values())))
        print("prepared", name, maps.shape)


def run():
    binary = os.path.join(HERE, "_ref", "bioEM_ref")
    binary_trace = os.path.join(HERE, "_ref", "bioEM_ref_trace")
    # the reference driving libbioem_hip.so through its own compareRefMaps virtual (oracle/ref_plugin, GPU=1)
    binary_hip = os.path.join(HERE, "_ref", "bioEM_ref_hip")
    ok = True
    for name in sorted(case_table()):
        d = os.path.join(CASES_DIR, name)
        inp = np.load(os.path.join(d, "inputs.npz"))
        out = os.path.join(OUT_DIR, name)
        os.makedirs(out, exist_ok=True)
        for algo in [int(a) for a in inp["algos"]]:
            runs = [(binary, "_algo%d" % algo)] + ([(binary_trace, "_algo%d_trace" % algo)] if bool(inp["trace"]) else [])
            if os.path.exists(binary_hip) and os.environ.get("BIOEM_GOLDEN_PLUGIN", "1") != "0":
                runs.append((binary_hip, "_plugin_algo%d" % algo))
            if os.environ.get("BIOEM_GOLDEN_PLUGIN_ONLY"):
                runs = [r_ for r_ in runs if r_[0] == binary_hip]
            for exe, tag in runs:
                env = dict(os.environ, OMP_NUM_THREADS="1", BIOEM_ALGO=str(algo), BIOEM_DEBUG_OUTPUT="0")
                env.pop("GPU", None)
                if exe == binary_hip:
                    env["GPU"] = "1"
                    if algo == 2:  # several convolutions per compareRefMaps call (nTotParallelConv = min(nCTF, 3))
                        env["BIOEM_PROJ_CONV_AT_ONCE"] = "3"
                if "env_keys" in inp.files:
                    for k_, v_ in zip(inp["env_keys"], inp["env_vals"]):
                        env[str(k_)] = str(v_)
                pformat = str(inp["particles"]) if "particles" in inp.files else "text"
                mformat = str(inp["model_format"]) if "model_format" in inp.files else "text"
                pfile = {"mrc": "particles.mrc", "multimrc": "list.txt"}.get(pformat, "particles.txt")
                mfile = {"pdb": "model.pdb", "mrc": "model.mrc"}.get(mformat, "model.txt")
                if pformat == "multimrc":
                    with open(os.path.join(d, "list.txt"), "w") as f:
                        for k_ in range(len(inp["stacks"])):
                            f.write(os.path.join(d, "stack%d.mrc" % k_) + "\n")
                cmd = [exe, "--Modelfile", os.path.join(d, mfile), "--Particlesfile",
                       os.path.join(d, pfile), "--Inputfile", os.path.join(d, "param.txt"),
                       "--OutputFile", "Output_Probabilities%s" % tag]
                if pformat in ("mrc", "multimrc"):
                    cmd.append("--ReadMRC")
                if pformat == "multimrc":
                    cmd.append("--ReadMultipleMRC")
                if mformat == "pdb":
                    cmd.append("--ReadPDB")
                if mformat == "mrc":
                    cmd.append("--ReadModelMRC")
                if os.path.exists(os.path.join(d, "orient.txt")):
                    cmd += ["--ReadOrientation", os.path.join(d, "orient.txt")]
                r = subprocess.run(cmd, cwd=out, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                with open(os.path.join(out, "stdout%s.txt" % tag), "w") as f:
                    f.write(r.stdout)
                if os.path.exists(os.path.join(out, "ANG_PROB")):
                    os.replace(os.path.join(out, "ANG_PROB"), os.path.join(out, "ANG_PROB%s" % tag))
                print(name, tag, "rc", r.returncode, flush=True)
                ok = ok and r.returncode == 0
    sys.exit(0 if ok else 1)


def collect():
    for name in sorted(case_table()):
        src_in = os.path.join(CASES_DIR, name)
        src_out = os.path.join(OUT_DIR, name)
        if not os.path.isdir(src_out):
            print("no outputs for", name)
            continue
        dst = os.path.join(GOLD_DIR, name)
        os.makedirs(dst, exist_ok=True)
        for f in ["inputs.npz", "param.txt", "model.pdb", "model.mrc"]:
            if os.path.exists(os.path.join(src_in, f)):
                shutil.copy(os.path.join(src_in, f), dst)
        for f in os.listdir(src_out):
            if f.startswith("Output_Probabilities") or f.startswith("ANG_PROB"):
                shutil.copy(os.path.join(src_out, f), dst)
            if f.startswith("stdout") and f.endswith("_trace.txt"):
                # keep only the per-displacement trace lines (DEBUG_PROB), compressed
                import gzip
                with open(os.path.join(src_out, f)) as fi, gzip.open(os.path.join(dst, f + ".gz"), "wt") as fo:
                    for ln in fi:
                        if "Prob" in ln or "Parameters:" in ln:
                            fo.write(ln)
        print("collected", name)


if __name__ == "__main__":
    {"prepare": prepare, "run": run, "collect": collect}[sys.argv[1]]()

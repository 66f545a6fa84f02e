"""The CPU oracle against outputs of the reference itself (tests/golden/, produced by
oracle/make_golden.py from oracle/_ref/bioEM_ref = unmodified reference sources + hipFFTW).

This is what pins the oracle: log P, the maximising tuple and (where float rounding of the FFT
backend allows) the byte-identical Output_Probabilities text, for CTF/PSF mode, quaternion list /
quaternion grid / Euler grid, point and sphere projection branches, odd N, ALGO 1 and ALGO 2,
maxD % grid != 0, WRITE_PROB_ANGLES and the per-displacement DEBUG_PROB trace.
"""
import gzip
import os
import re

import numpy as np
import pytest

import io_formats as iof
import oracle as orc
from golden_util import CASES, GOLDEN, golden_output, load_case, oracle_setup

# |log P| differences come only from float rounding inside the FFT backend (hipFFT float vs the
# oracle's double DFT); observed <= 2.2e-3 at 224^2 (3e-8 relative).  north_star tolerance: 1e-4 relative.
ABS_TOL = 5e-3
_setups = {}


def setup_for(name):
    if name not in _setups:
        case = load_case(name)
        _setups[name] = (case, oracle_setup(case))
    return _setups[name]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_output(name):
    case, S = setup_for(name)
    for algo in case["algos"]:
        pmap, pang = S.run(algo)
        mine = iof.parse_output_probabilities(orc.format_output_probabilities(S, pmap))
        gold = iof.parse_output_probabilities(golden_output(case, algo))
        assert len(mine) == len(gold) == S.nMaps
        for a, b in zip(gold, mine):
            assert abs(a["logp"] - b["logp"]) <= ABS_TOL
            assert abs(a["logp"] - b["logp"]) <= 1e-4 * abs(a["logp"])
            assert abs(a["constant"] - b["constant"]) <= ABS_TOL
            # identical maximising tuple (orientation, CTF, displacement) and printed norm/offset
            assert a["angles"] == b["angles"]
            assert a["ctf"] == b["ctf"]
            assert (a["cx"], a["cy"]) == (b["cx"], b["cy"])
            assert abs(a["norm"] - b["norm"]) <= 2e-4 and abs(a["mu"] - b["mu"]) <= 2e-4


@pytest.mark.parametrize("name", ["g3_n32_trace", "g5_n32_psf", "g6_n32_euler", "g9_n35_odd"])
def test_oracle_output_text_is_byte_identical(name):
    case, S = setup_for(name)
    for algo in case["algos"]:
        pmap, _ = S.run(algo)
        assert orc.format_output_probabilities(S, pmap) == golden_output(case, algo)


def test_algo1_equals_algo2_when_grid_divides_maxd():
    # invariant implied by the reference (displacement sets coincide iff maxD % g == 0)
    case, S = setup_for("g1_n48")
    p1, _ = S.run(1)
    p2, _ = S.run(2)
    for a, b in zip(p1, p2):
        assert abs(S.final_logp(a) - S.final_logp(b)) < 1e-6
        assert (a["cent_x"], a["cent_y"], a["orient"], a["conv"]) == (b["cent_x"], b["cent_y"], b["orient"], b["conv"])


@pytest.mark.parametrize("name", ["g4_n32_angles", "g11_n32_eulerlist"])
def test_ang_prob_matches_reference(name):
    """ANG_PROB: K best orientations per particle; with PRIOR_ANGLES the per-orientation prior is added to the
    printed log P and appended as a column (bioem.cpp:1297-1311)."""
    case, S = setup_for(name)
    for algo in case["algos"]:
        pmap, pang = S.run(algo)
        rows = orc.ang_prob_rows(S, pmap, pang)
        gold = iof.parse_ang_prob(os.path.join(case["dir"], "ANG_PROB_algo%d" % algo))
        assert sorted(gold) == sorted(rows)
        for m in gold:
            assert len(gold[m]) == len(rows[m]) == S.pd.writeAngles
            for g, r in zip(gold[m], rows[m]):
                q = S.angles[r["orient"]]
                nang = 4 if S.isQuat else 3
                assert g["angles"] == [float("%.4f" % v) for v in q[:nang]]
                assert abs(g["logp"] - r["logp"]) <= ABS_TOL
                assert abs(g["sep"][0] - r["logsum"]) <= ABS_TOL
                assert abs(g["sep"][1] - r["const"]) <= ABS_TOL
                assert abs(g["sep"][2] - r["numconst"]) <= 1e-3
                if r["prior"] is not None:
                    assert abs(g["sep"][3] - r["prior"]) <= 1e-4


def test_per_displacement_trace():
    """DEBUG_PROB trace of the reference (bioem_algorithm.h:88-92): every (map, orient, conv, dx, dy)
    cross-correlation value and logpro, ALGO 1."""
    case, S = setup_for("g3_n32_trace")
    pat = re.compile(r"Prob: iRefMap (\d+), iOrient (\d+), iConv (\d+), disx (-?\d+), disy (-?\d+), address -, "
                     r"value (\S+), logpro (\S+)")
    gold = {}
    with gzip.open(os.path.join(case["dir"], "stdout_algo1_trace.txt.gz"), "rt") as f:
        for ln in f:
            m = pat.search(ln)
            if m:
                k = tuple(int(v) for v in m.groups()[:5])
                gold[k] = (float(m.group(6)), float(m.group(7)))
    pd = S.pd
    N = S.N
    assert len(gold) == S.nMaps * S.nAngles * S.nCTF * pd.NtotDisp
    worst_v = worst_l = 0.0
    vscale = max(abs(v[0]) for v in gold.values())
    for io in range(S.nAngles):
        conv, p5 = S.conv_spectra(io)
        for c in range(S.nCTF):
            for m in range(S.nMaps):
                cc = orc.cc_map(conv[c], S.refFFT[m])
                for (mm, oo, cc_i, dx, dy), (gv, gl) in gold.items():
                    if (mm, oo, cc_i) != (m, io, c):
                        continue
                    val = np.float32(cc[dx % N, dy % N]) / np.float32(N * N)
                    lp = np.float32(orc.lib().orc_calc_logpro(pd, p5[c]["amp"], p5[c]["pha"], p5[c]["env"],
                                                              p5[c]["sumC"], p5[c]["sumsquareC"], val,
                                                              S.sumRef[m], S.sumsqRef[m]))
                    worst_v = max(worst_v, abs(float(val) - gv))
                    worst_l = max(worst_l, abs(float(lp) - gl))
    # printed with %f (6 decimals); value differences are float rounding of the FFT backend
    assert worst_v <= 2e-6 * vscale + 1e-5
    assert worst_l <= 2e-3


@pytest.mark.parametrize("name", CASES)
def test_reference_driving_the_hip_plugin_equals_reference_cpu(name):
    """The compiled drop-in proof: oracle/_ref/bioEM_ref_hip = the UNMODIFIED reference sources built with -DWITH_CUDA
    plus oracle/ref_plugin/bioem_hip_plugin.cpp (class bioem_hip : public bioem, bioem_cuda_create()), linked against
    libbioem_hip.so.  Run with GPU=1 on the MI355X box by `oracle/make_golden.py run`, the reference's own run() loop
    (host projection / convolution, bioem.cpp:763-891) called the engine through the compareRefMaps virtual
    (bioem.cpp:853): one convolution per call for ALGO 1, up to three (BIOEM_PROJ_CONV_AT_ONCE=3) for ALGO 2.  Its
    committed outputs must equal the reference CPU path's outputs on the same inputs."""
    case = load_case(name)
    for algo in case["algos"]:
        ref = iof.parse_output_probabilities(golden_output(case, algo))
        plug = iof.parse_output_probabilities(golden_output(case, algo, plugin=True))
        assert len(ref) == len(plug) > 0
        for a, b in zip(ref, plug):
            assert abs(a["logp"] - b["logp"]) <= 1e-4 * abs(a["logp"]) and abs(a["logp"] - b["logp"]) <= ABS_TOL
            assert (a["angles"], a["ctf"], a["cx"], a["cy"]) == (b["angles"], b["ctf"], b["cx"], b["cy"])
            assert abs(a["norm"] - b["norm"]) <= 2e-4 and abs(a["mu"] - b["mu"]) <= 2e-4
        # the header block (notation, units) is the reference's own writer either way
        assert golden_output(case, algo).split("\n\n")[0] == golden_output(case, algo, plugin=True).split("\n\n")[0]
        ang = os.path.join(case["dir"], "ANG_PROB_algo%d" % algo)
        if os.path.exists(ang):
            ga = iof.parse_ang_prob(ang)
            pa = iof.parse_ang_prob(os.path.join(case["dir"], "ANG_PROB_plugin_algo%d" % algo))
            assert sorted(ga) == sorted(pa)
            for m in ga:
                assert len(ga[m]) == len(pa[m])
                for g, p in zip(ga[m], pa[m]):
                    assert g["angles"] == p["angles"] and abs(g["logp"] - p["logp"]) <= ABS_TOL


def test_c2r_of_a_non_hermitian_half_spectrum_three_implementations():
    """The one convention of the reference's FFT dependency that the goldens pin only through hipFFTW (FFTW is absent
    from the image; the reference holds no fixtures): the unnormalised 2-D c2r (fftwf_plan_dft_c2r_2d,
    bioem.cpp:1458 / param.cpp:1521) of a half-spectrum that is NOT Hermitian in columns 0 and N/2 -- what the CTF row
    quirk (param.cpp:1560-1568) hands to it.  FFTW's documented rdft2 behaviour: complex transform along the first
    axis, then a c2r along the last that ignores the imaginary parts of its DC and Nyquist inputs.  Three independent
    implementations must agree on seeded random spectra: the oracle's double-precision DFT, numpy's pocketfft irfft2,
    and hipFFTW's output committed in tests/golden/c2r_nonhermitian.npz (oracle/fft_probe/make_fixture.py, run on
    the GPU box).  This documents the convention; it is not a pin against FFTW itself."""
    d = np.load(os.path.join(GOLDEN, "c2r_nonhermitian.npz"))
    for N in [int(n) for n in d["sizes"]]:
        spec = d["in_%d" % N]
        assert np.abs(spec[:, 0, 1]).max() > 0.1 and np.abs(spec[0, :, 1]).max() > 0.1         # really not Hermitian
        mine = orc.fft2_c2r(spec).astype(np.float64)
        z = spec[..., 0].astype(np.float64) + 1j * spec[..., 1].astype(np.float64)
        pocket = np.fft.irfft2(z, s=(N, N)) * (N * N)
        hipfftw = d["hipfftw_%d" % N].astype(np.float64)
        scale = np.abs(pocket).max()
        assert np.abs(mine - pocket).max() <= 2e-6 * scale
        assert np.abs(hipfftw - pocket).max() <= 2e-5 * scale
        assert np.abs(hipfftw - mine).max() <= 2e-5 * scale

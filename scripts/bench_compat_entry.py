#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference-compatible entry bioem_hip_compare (== bioem::compareRefMaps): the host hands
over conv spectra per call (2-slot pipeline buffers, bioem.cpp:825-853), as the reference's run loop does.  The conv
spectra are taken from the device path once (debug_convolution) so that only the hand-over + comparison is timed.
usage: python scripts/bench_compat_entry.py [--pixels 224] [--particles 1000] [--orientations 64]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bioem_amd.engine as eng  # noqa: E402
from bioem_amd.synthetic import Workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pixels", type=int, default=224)
ap.add_argument("--particles", type=int, default=1000)
ap.add_argument("--orientations", type=int, default=64)
ap.add_argument("--convs-per-call", type=int, default=0,
                help="convolutions handed over per call (default: all CTFs of an orientation; 1 = the reference's "
                     "default ALGO-1 loop, nTotParallelConv = 1, bioem.cpp:534)")
args = ap.parse_args()

W = Workload(N=args.pixels, nP=args.particles, nOrient=args.orientations)
E, nC = W.engine, W.nCTF
H = args.pixels // 2 + 1
convs = np.zeros((args.orientations, nC, args.pixels, H, 2), dtype=np.float32)
pars = np.zeros((args.orientations, nC), dtype=eng.PARAM5_DTYPE)
for o in range(args.orientations):
    for c in range(nC):
        spec, sumC, sumsqC = E.debug_convolution(o, c)
        convs[o, c] = spec
        pars[o, c] = (W.ctfParam[c][0], W.ctfParam[c][1], W.ctfParam[c][2], sumC, sumsqC)
nPar = args.convs_per_call if args.convs_per_call > 0 else nC
conv_base = np.zeros((2 * nPar, args.pixels, H, 2), dtype=np.float32)
par_base = np.zeros(2 * nPar, dtype=eng.PARAM5_DTYPE)
ncalls = args.orientations * ((nC + nPar - 1) // nPar)


def one_pass():
    raw, pmap, _ = eng.new_prob_block(W.nP, W.nOrient, 0)
    E.start_run(raw)
    ipipe = 0
    for o in range(args.orientations):
        for c0 in range(0, nC, nPar):
            n = min(nPar, nC - c0)
            k = (ipipe & 1) * nPar
            conv_base[k:k + n] = convs[o, c0:c0 + n]   # the host's "createConvolutedProjectionMap" output lands here
            par_base[k:k + n] = pars[o, c0:c0 + n]
            E.compare(ipipe, o, c0, n, nPar, conv_base, par_base)
            ipipe += 1
    E.finish_run(raw)
    return pmap


ref = one_pass()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    got = one_pass()
dt = (time.perf_counter() - t0) / reps
# same result as the all-device path
raw, pmap, _ = eng.new_prob_block(W.nP, W.nOrient, 0)
E.start_run(raw)
E.project_convolve_compare(0, args.orientations)
E.finish_run(raw)
la = np.log(got["Total"]) + got["Constoadd"]
lb = np.log(pmap["Total"]) + pmap["Constoadd"]
n = args.orientations * nC * args.particles
print("compat entry: %d comparisons in %.1f ms = %.2f M comparisons/s (%.2f MB handed over per call, %d calls); "
      "max |dlogP| vs device path %.2e" % (n, dt * 1e3, n / dt / 1e6, nPar * args.pixels * H * 8 / 1e6, ncalls,
                                          np.abs(la - lb).max()))

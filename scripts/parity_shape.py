#!/usr/bin/env python3
"""Parity of ONE bench shape at small counts (5 particles x 7 orientations x the shape's CTF grid, ALGO 1 and 2) against
the CPU oracle -- the quick check beside an A/B timing (scripts/ab_slim.sh); takes bench.py's shape flags."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

ap = argparse.ArgumentParser()
ap.add_argument("--pixels", type=int, default=224)
ap.add_argument("--envelopes", type=int, default=5)
ap.add_argument("--defocus", type=int, default=1)
ap.add_argument("--max-displacement", type=int, default=10)
ap.add_argument("--grid", type=int, default=1)
ap.add_argument("--particles", type=int, default=5)
ap.add_argument("--orientations", type=int, default=7)
a, _ = ap.parse_known_args()

import test_gpu_parity as T
from bioem_amd.synthetic import Workload

worst, sig = 0.0, ""
for algo in (1, 2):
    W = Workload(N=a.pixels, nP=5, nOrient=7, nEnv=min(a.envelopes, 2), nDefocus=min(a.defocus, 2), maxD=a.max_displacement,
                 grid=a.grid, algo=algo, npts=300)
    sel = list(range(5))
    want, const = T.oracle_on_workload(W, sel, 7, algo)
    _, got = T.run_workload(W, 0, 7)
    T.assert_workload_matches(got, want, const, sel)
    la = np.log(got["Total"]) + got["Constoadd"]
    lb = np.log(want["Total"]) + want["Constoadd"]
    worst = max(worst, float(np.abs(la - lb).max()))
    sig = W.engine.kernel_signature
    W.engine.close()
print("parity ok: max |dlogP| %.2e, arg-max tuples equal, %s" % (worst, sig))

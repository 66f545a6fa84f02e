#!/usr/bin/env python3
"""Device time of the preparation phases alone (projection + r2c, convolution) against the batch size, from the
engine's own phase records: how the resident grids and the one-block-per-orientation kernels fill the chip.
usage: python scripts/prep_scaling.py [--pixels 224] [--particles 20] [--envelopes 5] [nO ...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bioem_amd.engine as eng  # noqa: E402
from bioem_amd.synthetic import Workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pixels", type=int, default=224)
ap.add_argument("--particles", type=int, default=20)
ap.add_argument("--envelopes", type=int, default=5)
ap.add_argument("sizes", type=int, nargs="*", default=[64, 128, 192, 256, 320, 384, 512, 768])
args = ap.parse_args()
W = Workload(N=args.pixels, nP=args.particles, nOrient=max(args.sizes), nEnv=args.envelopes)
E, nC = W.engine, W.nCTF
maxO, maxRows = E.max_batch()
raw, pmap, _ = eng.new_prob_block(W.nP, W.nOrient, 0)
E.start_run(raw)
E.set_phase_timing(True)
for nO in args.sizes:
    if nO > maxO or nO * nC > maxRows:
        print("%d orientations: beyond this handle's batch (%d, %d rows)" % (nO, maxO, maxRows))
        continue
    for rep in range(3):
        E.project(0, 0, nO)
        E.convolve(0, 0, nC)
        E.synchronize()
    rec = E.phase_records()
    p = [r["seconds"] for r in rec if r["phase"] == 0][-2:]
    c = [r["seconds"] for r in rec if r["phase"] == 1][-2:]
    print("%4d orientations x %d CTFs at %d^2: projection + r2c %7.1f us (%.3f us each), convolution %7.1f us (%.3f us per spectrum)"
          % (nO, nC, args.pixels, min(p) * 1e6, min(p) * 1e6 / nO, min(c) * 1e6, min(c) * 1e6 / (nO * nC)))
E.finish_run(raw)

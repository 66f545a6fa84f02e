#!/usr/bin/env python3
"""Whole-job rate of the staged device entries (bioem_hip_project / _convolve / _compare_device, the loop body of
bioem.cpp:763-891 piece by piece, everything device-resident) against the fused entry on the same workload.
usage: python scripts/bench_staged_entries.py [--pixels 224] [--particles 1000] [--orientations 1152] [--batch 64]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bioem_amd.engine as eng  # noqa: E402
from bioem_amd.synthetic import Workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pixels", type=int, default=224)
ap.add_argument("--particles", type=int, default=1000)
ap.add_argument("--orientations", type=int, default=1152)
ap.add_argument("--batch", type=int, default=64, help="orientations per bioem_hip_project call")
args = ap.parse_args()

W = Workload(N=args.pixels, nP=args.particles, nOrient=args.orientations)
E, nC = W.engine, W.nCTF
maxO, maxRows = E.max_batch()
B = min(args.batch, maxO, maxRows // nC)


def fused():
    raw, pmap, _ = eng.new_prob_block(W.nP, W.nOrient, 0)
    E.start_run(raw)
    E.project_convolve_compare(0, W.nOrient)
    E.finish_run(raw)
    return raw


def staged():
    raw, pmap, _ = eng.new_prob_block(W.nP, W.nOrient, 0)
    E.start_run(raw)
    for b, o0 in enumerate(range(0, W.nOrient, B)):
        E.project(b, o0, min(o0 + B, W.nOrient))
        E.convolve(b, 0, nC)
        E.compare_device(b)
    E.finish_run(raw)
    return raw


for name, fn in (("fused  bioem_hip_project_convolve_compare", fused), ("staged project / convolve / compare_device", staged)):
    fn()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        raw = fn()
    dt = (time.perf_counter() - t0) / reps
    print("%-46s %7.2f M comparisons/s  (%.1f ms per pass, %d orientations x %d CTFs x %d particles, batch %d)"
          % (name, W.comparisons_per_pass / dt / 1e6, dt * 1e3, W.nOrient, nC, W.nP, B))

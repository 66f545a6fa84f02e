#!/usr/bin/env python3
"""Diagnostic (build with `make -C bioem_amd/csrc EXTRA=-DBIOEM_W2_STAMPS`): share of wave-0 shader cycles per phase of
k_compare_wide2 on the tutorial production window (+-40 px at 224^2)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bioem_amd.engine as eng  # noqa: E402
from bioem_amd.synthetic import Workload  # noqa: E402

maxd = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W = Workload(N=224, nP=1000, nOrient=32, nEnv=4, nDefocus=8, maxD=maxd)
E = W.engine
L = eng.load_library()
out = (C.c_ulonglong * 16)()
for it in range(2):
    raw, pmap, _ = eng.new_prob_block(W.nP, W.nOrient, 0)
    E.start_run(raw)
    E.project_convolve_compare(0, W.nOrient)
    E.finish_run(raw)
    L.bioem_hip_debug_w2_stamps(out)
v = [int(x) for x in out]
n = W.nOrient * W.nCTF * W.nP
names = ["column pass: tail", "T -> LDS + barrier", "row FFT", "recombination + posterior: rest (chunk set-up)", "wave reduce + barrier",
         "column: barrier after produce", "column: fold + barrier", "-", "column: loads + spectrum product",
         "column: register FFT", "column: park outputs, request next F", "recombination over k1 (in loop)",
         "posterior batch (in loop)"]
tot = sum(v[:13])
print(E.kernel_signature, "comparisons", n)
for k in (8, 9, 10, 5, 6, 0, 1, 2, 11, 12, 3, 4):
    print("%-34s %8.0f cycles/comparison  %5.1f %%" % (names[k], v[k] / n, 100.0 * v[k] / tot))

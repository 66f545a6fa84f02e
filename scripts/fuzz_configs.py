#!/usr/bin/env python3
"""One-off fuzz (GPU box): seeded random configurations of the comparison kernels against the CPU oracle.
usage: python scripts/fuzz_configs.py [first_seed last_seed] [sizes, e.g. 192,256,320,384]   (100 configurations per seed, 25
with a size list; default seeds 1..6)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, _p)
import numpy as np
import test_gpu_parity as T
from bioem_amd.synthetic import Workload
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 6)
sizes = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else None
for seed in range(lo, hi + 1):
    for cfg in T._random_configs(25 if sizes else 100, seed, sizes):
        N, maxD, grid, algo, nEnv, nP, nO = cfg
        try:
            W = Workload(N=N, nP=nP, nOrient=nO, nEnv=nEnv, maxD=maxD, grid=grid, algo=algo, npts=150)
        except Exception as e:
            print("CREATE FAIL", cfg, str(e)[:100]); bad += 1; continue
        try:
            sel = list(range(nP))
            want, const = T.oracle_on_workload(W, sel, nO, algo)
            _, got = T.run_workload(W, 0, nO)
            T.assert_workload_matches(got, want, const, sel)
        except AssertionError as e:
            print("MISMATCH", cfg, "fast" if W.engine.fast_path else "generic"); bad += 1
        finally:
            W.engine.close()
print("done, failures:", bad)

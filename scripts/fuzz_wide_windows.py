#!/usr/bin/env python3
"""One-off fuzz (GPU box) of the wide-window kernels: seeded random even image sizes 130..384 (two and three column blocks,
every register-FFT length), window half widths 16..47 rows with row strides 1..2, ALGO 1 / 2, against the CPU oracle.
usage: python scripts/fuzz_wide_windows.py [seed [count]]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, _p)
import numpy as np
import test_gpu_parity as T
from bioem_amd.synthetic import Workload
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(seed)
bad = 0
kernels = {}
for it in range(count):
    N = int(rng.choice([2 * int(rng.integers(65, 193)), int(rng.choice([160, 192, 200, 224, 240, 250, 256, 288, 300, 320, 360, 384]))]))
    grid = int(rng.choice([1, 1, 1, 2]))
    rows = int(rng.integers(16, 48))
    maxD = rows * grid
    if 2 * maxD + 2 >= N:
        continue
    algo = int(rng.choice([1, 2]))
    nEnv, nP, nO = int(rng.integers(1, 3)), int(rng.integers(1, 5)), int(rng.integers(1, 6))
    cfg = (N, maxD, grid, algo, nEnv, nP, nO)
    try:
        W = Workload(N=N, nP=nP, nOrient=nO, nEnv=nEnv, maxD=maxD, grid=grid, algo=algo, npts=150)
    except Exception as e:
        print("CREATE FAIL", cfg, str(e)[:100]); bad += 1; continue
    try:
        sig = W.engine.kernel_signature
        kernels[sig] = kernels.get(sig, 0) + 1
        sel = list(range(nP))
        want, const = T.oracle_on_workload(W, sel, nO, algo)
        _, got = T.run_workload(W, 0, nO)
        T.assert_workload_matches(got, want, const, sel)
    except AssertionError:
        print("MISMATCH", cfg, sig); bad += 1
    finally:
        W.engine.close()
for k in sorted(kernels):
    print("%4d  %s" % (kernels[k], k))
print("done, failures:", bad)

"""Every comparison-kernel instantiation of the code object runs at least once, against the CPU oracle: for each
distinct kernel of the selection snapshot (tests/golden/selection_snapshot.txt.gz) the smallest shape that selects it,
3 particles x 3 orientations x 2 CTFs.  scripts/check_kernel_coverage.py then holds the record of the instantiations
the whole GPU run selected against the code object."""
import gzip
import os

import pytest

from test_gpu_parity import assert_workload_matches, oracle_on_workload, run_workload

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _smallest_shape_per_kernel():
    best = {}
    with gzip.open(os.path.join(ROOT, "tests", "golden", "selection_snapshot.txt.gz"), "rt") as f:
        for ln in f:
            N, d, g, algo, sig = ln.rstrip("\n").split(" ", 4)
            key = sig.split(" x ")[0]
            cost = (int(N) ** 2 * (2 * (int(d) // int(g)) + 1), int(algo))
            if key not in best or cost < best[key][0]:
                best[key] = (cost, (int(N), int(d), int(g), int(algo), sig))
    return [v[1] for _, v in sorted(best.items())]


@pytest.mark.parametrize("shape", _smallest_shape_per_kernel(), ids=lambda s: s[4].split(" x ")[0].replace(" ", ""))
def test_kernel_instantiation_against_oracle(shape):
    from bioem_amd.synthetic import Workload
    N, d, g, algo, sig = shape
    W = Workload(N=N, nP=3, nOrient=3, nEnv=2, maxD=d, grid=g, algo=algo, npts=150)
    try:
        assert W.engine.kernel_signature == sig.split(" x ")[0]
        sel = [0, 1, 2]
        want, const = oracle_on_workload(W, sel, 3, algo)
        _, got = run_workload(W, 0, 3)
        assert_workload_matches(got, want, const, sel)
    finally:
        W.engine.close()


@pytest.mark.parametrize("d", [5, 10, 13, 15, 20, 30, 40])
@pytest.mark.parametrize("nP", [3, 66])
def test_nyquist_column_kernel_against_oracle(d, nP):
    """k_nyquist_rows<WD, Q> (the Nyquist column of 128^2 / 256^2 by direct summation, compare_fast.hpp): every window
    depth the comparison kernels ask for, with four waves per 64 pairs (64 particles or fewer) and with one thread per
    pair (more)."""
    from bioem_amd.synthetic import Workload
    W = Workload(N=128, nP=nP, nOrient=2, nEnv=2, maxD=d, npts=150)
    try:
        sel = [0, 1, nP - 1]
        want, const = oracle_on_workload(W, sel, 2, 1)
        _, got = run_workload(W, 0, 2)
        assert_workload_matches(got, want, const, sel)
    finally:
        W.engine.close()

#!/usr/bin/env python3
"""Kernel selection over a grid of (image size, window half width, grid spacing, algorithm): one line per shape with the
instantiation the library picks.  With a GPU the handles are really created (bioem_hip_create); the planner entry
bioem_hip_plan gives the same answer without one (tests/test_selection_table.py compares it with the committed
snapshot tests/golden/selection_snapshot.txt).

usage: scripts/selection_snapshot.py [--plan] > snapshot.txt
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SIZES = [8, 10, 16, 24, 32, 34, 36, 40, 42, 44, 48, 50, 56, 60, 64, 72, 80, 84, 88, 90, 96, 100, 112, 120, 126, 128, 144, 150, 160,
         176, 180, 192, 198, 200, 208, 210, 224, 240, 248, 250, 256, 264, 272, 280, 288, 290, 300, 320, 360, 380, 384, 400, 432,
         448, 512, 600, 9, 33, 35, 51, 75, 99, 125, 127, 129, 135, 225]
WINDOWS = [0, 2, 4, 5, 7, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 21, 22, 24, 25, 26, 27, 30, 33, 35, 37, 38, 40, 41, 42, 43,
           44, 45, 47, 56, 60, 62, 64, 78, 80, 88]
GRIDS = [1, 2, 3, 4, 5]


def shapes():
    for N in SIZES:
        for d in WINDOWS:
            if d >= N // 2:
                continue
            for g in GRIDS:
                if g > 1 and (d // g) < 2:
                    continue
                if d // g > 47:
                    continue
                for algo in (1, 2):
                    yield N, d, g, algo


def main():
    import bioem_amd.engine as eng
    L = eng.load_library()
    plan = "--plan" in sys.argv
    if plan:
        L.bioem_hip_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
    for N, d, g, algo in shapes():
        if plan:
            buf = C.create_string_buffer(160)
            rc = L.bioem_hip_plan(N, d, g, algo, buf, 160)
            sig = buf.value.decode() if rc == 0 else "rejected"
        else:
            pd = eng.ParamDevice()
            pd.maxDisplaceCenter, pd.GridSpaceCenter, pd.NumberPixels, pd.NumberFFTPixels1D = d, g, N, N // 2 + 1
            pd.NxDisp = 2 * (d // g) + 1
            pd.NtotDisp = pd.NxDisp ** 2
            pd.Ntotpi = float(N * N)
            pd.volu = 1.0
            pd.sigmaPriorbctf = pd.sigmaPriordefo = pd.sigmaPrioramp = 1.0
            try:
                E = eng.Engine(pd, 1, 1, 1, algo=algo, device=0)
                sig = E.kernel_signature
                E.close()
            except RuntimeError as e:
                sig = "rejected"
        print("%d %d %d %d %s" % (N, d, g, algo, sig), flush=True)


if __name__ == "__main__":
    main()

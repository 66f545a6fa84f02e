"""Kernel selection (bioem_amd/csrc/kernel_select.hpp) without a GPU: bioem_hip_plan is a pure function of the image size
and the displacement set.  The committed snapshot (tests/golden/selection_snapshot.txt.gz: 17 528 shapes -- 67 image
sizes x 40 window half widths x grid spacings 1..5 x ALGO 1/2, written by scripts/selection_snapshot.py --plan) pins
what every shape runs; a change of the table or the rules shows up here as a diff, on purpose."""
import ctypes as C
import gzip
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def plan(L, N, d, g, algo):
    buf = C.create_string_buffer(160)
    rc = L.bioem_hip_plan(N, d, g, algo, buf, 160)
    return buf.value.decode() if rc == 0 else "rejected"


def test_selection_matches_the_committed_snapshot():
    import bioem_amd.engine as eng
    L = eng.load_library()
    bad = []
    n = 0
    with gzip.open(os.path.join(ROOT, "tests", "golden", "selection_snapshot.txt.gz"), "rt") as f:
        for ln in f:
            N, d, g, algo, sig = ln.rstrip("\n").split(" ", 4)
            got = plan(L, int(N), int(d), int(g), int(algo))
            n += 1
            if got != sig:
                bad.append((N, d, g, algo, sig, got))
    assert n > 16000
    assert not bad, "%d shapes changed kernel, e.g. %s" % (len(bad), bad[:5])


def test_every_table_entry_is_selected_by_some_shape_of_the_snapshot():
    """No dead instantiations: each line of kernel_table.inc is what at least one shape of the snapshot runs."""
    table = set()
    for ln in open(os.path.join(ROOT, "bioem_amd", "csrc", "kernel_table.inc")):
        m = re.match(r"K_(\w+)\((.*)\)", ln.strip())
        if m:
            fam = {"FAST": "fast", "FASTM": "fastm", "FASTM2": "fastm2", "WIDE2": "wide2", "ROWS": "rows", "ODDFFT": "oddfft"}[m.group(1)]
            table.add("k_compare_%s<%s>" % (fam, ",".join(a.strip() for a in m.group(2).split(","))))
    seen = set()
    with gzip.open(os.path.join(ROOT, "tests", "golden", "selection_snapshot.txt.gz"), "rt") as f:
        for ln in f:
            sig = ln.rstrip("\n").split(" ", 4)[4].split(" x ")[0].replace(" ", "")
            m = re.match(r"k_compare_wide2<(.*)>", sig)
            if m:
                a = m.group(1).split(",")
                a += ["1", "4"][len(a) - 4:] if len(a) < 6 else []
                sig = "k_compare_wide2<%s>" % ",".join(a)
            seen.add(sig)
    assert sorted(table - seen) == []


def test_headline_shapes():
    import bioem_amd.engine as eng
    L = eng.load_library()
    assert plan(L, 224, 10, 1, 1) == "k_compare_fast<10, 16, false, 1>"            # BASELINE config 2 / 3
    assert plan(L, 288, 10, 1, 1) == "k_compare_fast<10, 32, false, 1>"            # beyond 256 pixels: the longest length
    assert plan(L, 96, 5, 1, 1) == "k_compare_fast<5, 8, false, 1>"                # 11 rows, small image: 8 points
    assert plan(L, 192, 10, 1, 1) == "k_compare_fast<10, 16, true, 1>"             # N / 2 = 32 (mod 64): Nyquist apart + split
    assert plan(L, 128, 10, 1, 1) == "k_compare_fast<10, 16, true, 1>"             # config 1 / 4 (Nyquist split)
    assert plan(L, 256, 10, 1, 1) == "k_compare_fast<10, 16, true, 1>"             # config 5
    assert plan(L, 384, 10, 1, 1) == "k_compare_fast<10, 32, true, 1>"
    assert plan(L, 128, 5, 1, 1) == "k_compare_fast<5, 8, true, 1>"
    assert plan(L, 224, 13, 1, 1) == "k_compare_fastm<13, 16, false, 1>"           # 27 rows: matrix-core window pass
    assert plan(L, 224, 20, 1, 1) == "k_compare_fastm2<16, false, 1>"              # 41 rows: rows split over the half-waves
    assert plan(L, 128, 16, 1, 2) == "k_compare_fastm2<16, true, 1>"
    assert plan(L, 200, 20, 1, 1) == "k_compare_fastm2<10, false, 1>"              # 16 does not divide 200
    assert plan(L, 224, 40, 2, 1) == "k_compare_fastm2<16, false, 2>"              # 41 rows at stride 2
    assert plan(L, 224, 40, 1, 1) == "k_compare_wide2<32, 21, 2, false>"           # tutorial production window
    assert plan(L, 225, 10, 1, 1) == "k_compare_oddfft<10, 25>"
    assert plan(L, 224, 120, 1, 1) == "rejected"                                   # maxD >= N / 2


def test_no_window_that_fits_a_kernel_falls_to_the_generic_one():
    """A symmetric window of 13..31 rows on an even image size runs a window kernel or tiles of one, at every row stride
    the table may lack (round 3 had no 31-row instantiation at stride 4: 114 <= N <= 519 with DISPLACE_CENTER 60 4 fell
    to the direct pruned DFT)."""
    import bioem_amd.engine as eng
    L = eng.load_library()
    bad = []
    for N in list(range(64, 530, 6)) + [128, 224, 256]:
        for g in (1, 2, 3, 4):
            for m in (6, 10, 13, 14, 15):
                d = m * g
                if d >= N // 2:
                    continue
                for algo in (1, 2):
                    sig = plan(L, N, d, g, algo)
                    if "generic" in sig or sig == "rejected":
                        bad.append((N, d, g, algo, sig))
    assert not bad, bad[:8]

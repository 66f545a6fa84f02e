#!/usr/bin/env python3
"""Writes bioem_amd/csrc/kernel_table.inc -- the comparison-kernel instantiations that exist -- from a selection
snapshot (scripts/selection_snapshot.py: every instantiation some shape of the grid selects), minus the families that
were retired, plus the instantiations named in EXTRA (needed by tests that force a path).

usage: scripts/make_kernel_table.py gpurun_out/selection_before.txt [--drop-spilling list.txt]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# instantiations no shape of the snapshot selected when it was taken, added as spill-free stand-ins for the dropped
# 32-point kernels of the same rule (scripts/kernel_table_drop.txt)
EXTRA = ["k_compare_wide2<16, 11, 4, true, 2, 8>", "k_compare_wide2<16, 24, 2, true, 2, 4>"]


def entry(sig):
    sig = sig.strip()
    m = re.match(r"k_compare_(\w+)<(.*)>$", sig)
    if not m:
        return None
    fam, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    if fam == "fast":
        return "K_FAST(%s)" % ", ".join(args)
    if fam == "fastm":
        return "K_FASTM(%s)" % ", ".join(args)
    if fam == "fastm2":
        return "K_FASTM2(%s)" % ", ".join(args)
    if fam == "wide2":
        args += ["1", "4"][len(args) - 4:] if len(args) < 6 else []
        return "K_WIDE2(%s)" % ", ".join(args)
    if fam == "rows":
        return "K_ROWS(%s)" % ", ".join(args)
    if fam == "oddfft":
        return "K_ODDFFT(%s)" % ", ".join(args)
    return None  # k_compare_wide: retired; k_compare_generic: always present


def main():
    sigs = set()
    for ln in open(sys.argv[1]):
        p = ln.strip().split(" ", 4)
        if len(p) == 5:
            sigs.add(p[4])
    sigs.update(EXTRA)
    drop = set()
    if "--drop" in sys.argv:
        drop = {ln.strip() for ln in open(sys.argv[sys.argv.index("--drop") + 1]) if ln.strip()}
    lines = sorted({e for e in (entry(s) for s in sigs) if e and e not in drop})
    out = os.path.join(ROOT, "bioem_amd", "csrc", "kernel_table.inc")
    with open(out, "w") as f:
        f.write("// kernel_table.inc -- the comparison-kernel instantiations of libbioem_hip.so (kernel_select.hpp).\n"
                "// Written by scripts/make_kernel_table.py from a selection snapshot; one line = one kernel in the code object.\n")
        for ln in lines:
            f.write(ln + "\n")
    print("%d instantiations -> %s" % (len(lines), out))


if __name__ == "__main__":
    main()

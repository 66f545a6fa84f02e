#!/usr/bin/env python3
"""Every comparison-kernel instantiation in the shipped code object must have been selected by some engine handle of
the GPU test run (BIOEM_SIGNATURE_LOG, written by the library at handle creation; tests/conftest.py points it at
gpurun_out/kernel_signatures_run.txt).  Prints the instantiations that never ran and exits 1 if there are any.

usage: scripts/check_kernel_coverage.py [gpurun_out/kernel_signatures_run.txt] [--so path]
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_code_object as cco  # noqa: E402
import tempfile  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def canon(sig):
    """k_compare_wide2<32, 21, 2, false> == k_compare_wide2<32, 21, 2, false, 1, 4> (defaulted template arguments)."""
    sig = sig.strip().replace(" ", "")
    m = re.match(r"(k_compare_wide2)<(.*)>$", sig)
    if m:
        a = m.group(2).split(",")
        a += ["1", "4"][len(a) - 4:] if len(a) < 6 else []
        sig = "k_compare_wide2<%s>" % ",".join(a)
    return sig


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    log = args[0] if args else os.path.join(ROOT, "gpurun_out", "kernel_signatures_run.txt")
    so = os.path.join(ROOT, "bioem_amd", "lib", "libbioem_hip.so")
    if "--so" in sys.argv:
        so = sys.argv[sys.argv.index("--so") + 1]
    with tempfile.TemporaryDirectory() as d:
        ks = [k for co in cco.extract_code_objects(so, d) for k in cco.kernels_of(co)]  # one code object per translation unit
    names = cco.demangle([k.get("name", k.get("symbol", "?")) for k in ks])
    shipped = sorted({canon(cco.short(n)) for n in names if re.match(r".*k_(compare|nyquist)", n)})
    ran = {canon(ln) for ln in open(log) if ln.strip()}
    never = [k for k in shipped if k not in ran]
    unknown = sorted(r for r in ran if r not in shipped)
    print("%d comparison-kernel instantiations shipped, %d selected by the test run, %d never selected"
          % (len(shipped), len([k for k in shipped if k in ran]), len(never)))
    for k in never:
        print("  never run:", k)
    for k in unknown:
        print("  logged but not in the code object (signature format?):", k)
    sys.exit(1 if never or unknown else 0)


if __name__ == "__main__":
    main()

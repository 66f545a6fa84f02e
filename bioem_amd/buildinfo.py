"""Identity of the device sources a measurement belongs to: git blob hashes (`git hash-object`) of the files that go
into the translation unit of the measured kernel (round 4: one translation unit per kernel family -- the counters of
k_compare_fast do not go stale when only k_compare_wide2 changes).  scripts/pmc_summary.py stores them beside the
counters it summarises; bench.py refuses counters whose hashes differ from the tree it runs from."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bioem_amd", "csrc")


def git_blob_sha1(path):
    with open(path, "rb") as f:
        data = f.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def translation_unit_of(kernel):
    """The .hip file that instantiates a comparison kernel (kernel = a demangled name or a signature such as
    'k_compare_wide2<32, 21, 2, false>'); None = unknown: every device source counts."""
    k = (kernel or "").replace(" ", "")
    m = re.search(r"k_compare_(fastm2|fastm|fast|wide2|rows|oddfft)<(\d+)", k)
    if not m:
        return None
    fam, a0 = m.group(1), int(m.group(2))
    if fam == "wide2":
        return "kernels_wide2_%s.hip" % ("short" if a0 <= 12 else "16" if a0 == 16 else "long")
    name = {"fast": "kernels_fast.hip", "fastm": "kernels_fastm.hip", "fastm2": "kernels_fastm2.hip",
            "rows": "kernels_odd.hip", "oddfft": "kernels_odd.hip"}[fam]
    return name if os.path.exists(os.path.join(CSRC, name)) else None


def include_closure(tu):
    """tu and every file of bioem_amd/csrc it includes by name, transitively."""
    seen, todo = set(), [tu]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(os.path.join(CSRC, f)):
            continue
        seen.add(f)
        with open(os.path.join(CSRC, f), errors="replace") as fh:
            for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), re.M):
                todo.append(m.group(1))
    return seen


def source_blobs(kernel=None):
    """Hashes of the device sources behind `kernel` (all of bioem_amd/csrc when the kernel's translation unit is not
    known), the build recipe and the C ABI header."""
    tu = translation_unit_of(kernel)
    names = include_closure(tu) if tu else {n for n in os.listdir(CSRC) if n.endswith((".hip", ".hpp", ".h", ".inc"))}
    if tu:  # which instantiations exist does not change the code of one of them (bench.py matches the signature itself)
        names.discard("kernel_table.inc")
    out = {}
    for name in sorted(names | {"Makefile"}):
        out["bioem_amd/csrc/" + name] = git_blob_sha1(os.path.join(CSRC, name))
    out["include/bioem_hip.h"] = git_blob_sha1(os.path.join(ROOT, "include", "bioem_hip.h"))
    return out

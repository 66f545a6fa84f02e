"""Identity of the device sources a measurement belongs to: git blob hashes (`git hash-object`) of everything under
bioem_amd/csrc/ that goes into libbioem_hip.so.  scripts/pmc_summary.py stores them beside the counters it summarises;
bench.py refuses counters whose hashes differ from the tree it runs from."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bioem_amd", "csrc")


def git_blob_sha1(path):
    with open(path, "rb") as f:
        data = f.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_blobs():
    out = {}
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".hpp", ".h")) or name == "Makefile":
            out["bioem_amd/csrc/" + name] = git_blob_sha1(os.path.join(CSRC, name))
    out["include/bioem_hip.h"] = git_blob_sha1(os.path.join(ROOT, "include", "bioem_hip.h"))
    return out

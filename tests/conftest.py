import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


def _ensure_built():
    """Build the in-tree native pieces when a fresh checkout has none of them yet (hipcc cross-compiles gfx950
    without a GPU).  The .so files are git-ignored; on the GPU box they arrive prebuilt with the snapshot."""
    import subprocess
    need_engine = not all(os.path.exists(os.path.join(ROOT, "bioem_amd", p)) for p in
                          ("lib/libbioem_hip.so", "lib/libbioem_host.so", "bin/bioEM"))
    if need_engine:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bioem_amd", "csrc"), "all"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


def pytest_sessionstart(session):
    _ensure_built()
    # PyTorch first: the RCCL tests need the process to hold PyTorch's copy of RCCL / the HSA runtime before
    # libbioem_hip.so brings in the system's (loaded the other way round, ncclCommInitAll finds "no ROCm-capable
    # device" -- seen when tests/test_gpu_parity.py ran on its own, where no earlier file had imported torch)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # every engine handle created during the run (this process and the CLI children) records the comparison-kernel
    # instantiation it selected: scripts/check_kernel_coverage.py compares the record with the code object
    log = os.path.join(ROOT, "gpurun_out", "kernel_signatures_run.txt")
    try:
        os.makedirs(os.path.dirname(log), exist_ok=True)
        open(log, "w").close()
        os.environ.setdefault("BIOEM_SIGNATURE_LOG", log)
    except OSError:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

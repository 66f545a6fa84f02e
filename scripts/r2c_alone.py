#!/usr/bin/env python3
"""The r2c kernels with the device to themselves: scripts/r2c_alone.py N nImg (run under rocprofv3 --kernel-trace --stats;
BIOEM_R2C=dft selects the exact-DFT kernels)."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bioem_amd import engine as eng
N, n = int(sys.argv[1]), int(sys.argv[2])
img = np.random.default_rng(1).standard_normal((n, N, N)).astype(np.float32)
for _ in range(3):
    out = eng.r2c(img)
print(N, n, float(np.abs(out).max()))

#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: makes tests/golden/c2r_nonhermitian.npz -- the one convention of the reference's FFT dependency
that the golden cases pin only through hipFFTW (the image has no FFTW): the unnormalised 2-D c2r of a half-spectrum
that is NOT Hermitian in columns 0 and N/2 (the CTF row quirk of param.cpp:1560-1568 makes the conv
spectra such).  Run on the GPU box (hipFFTW executes on a GPU):

    gpurun -- 'python oracle/fft_probe/make_fixture.py gpurun_out/c2r_nonhermitian.npz'

then copy the file to tests/golden/.  Inputs are seeded random spectra; `hipfftw` holds the library's outputs."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SIZES = (8, 12, 9, 32, 35, 224)


def spectrum(N, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(N, N // 2 + 1, 2)).astype(np.float32)      # imaginary parts everywhere: not Hermitian


def main(out):
    exe = os.path.join(tempfile.gettempdir(), "c2r_probe")
    subprocess.check_call(["g++", "-O1", "-I/opt/rocm/include", "-I/opt/rocm/include/hipfft",
                           os.path.join(HERE, "c2r_probe.cpp"), "-o", exe, "-L/opt/rocm/lib", "-lhipfftw",
                           "-Wl,-rpath,/opt/rocm/lib"])
    data = {}
    with tempfile.TemporaryDirectory() as d:
        for N in SIZES:
            s = spectrum(N, 1000 + N)
            s.tofile(os.path.join(d, "in.bin"))
            subprocess.check_call([exe, str(N), os.path.join(d, "in.bin"), os.path.join(d, "out.bin")])
            data["in_%d" % N] = s
            data["hipfftw_%d" % N] = np.fromfile(os.path.join(d, "out.bin"), dtype=np.float32).reshape(N, N)
    np.savez_compressed(out, sizes=np.array(SIZES), **data)
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1])

#!/bin/bash
# gpurun with oracle/_ref/ (the reference compiled by `make -C oracle ref ref_hip`) travelling to the GPU box:
# only for `oracle/make_golden.py run` and for tests/test_gpu_parity.py::test_reference_binary_drives_this_build.
# Every other call keeps it off the box (.gpurunignore; SURVEY.md 8c travel rule).
#   scripts/gpurun_with_reference.sh [--timeout S] -- '<command>'
set -u
cd "$(dirname "$0")/.."
cp .gpurunignore .gpurunignore.saved
trap 'mv -f .gpurunignore.saved .gpurunignore' EXIT
grep -v '^oracle/_ref/' .gpurunignore.saved > .gpurunignore
/usr/local/graft/bin/gpurun "$@"

#!/bin/bash
# Experiment build of libbioem_hip.so with ONE k_compare_fast and/or ONE k_compare_wide2 instantiation (seconds
# instead of minutes); never shipped.  Load it with BIOEM_HIP_LIBRARY=abl/<name>.so.
#   scripts/slim_build.sh <name> [-DBIOEM_SLIM_FAST=10,32,false,1] [-DBIOEM_SLIM_W2=32,21,2,false,1,4] [more -D flags]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p abl
# (the fast r2c is its own translation unit with its own flag: the object of the regular build is linked in)
[ -f bioem_amd/csrc/build/kernels_r2c.o ] || make -s -C bioem_amd/csrc build/kernels_r2c.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 \
  -Wno-unused-value -DBIOEM_SLIM "$@" -Iinclude -c -o abl/$name.o bioem_amd/csrc/bioem_hip.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/$name.so abl/$name.o bioem_amd/csrc/build/kernels_r2c.o
rm -f abl/$name.o
python scripts/check_code_object.py --so abl/$name.so 2>&1 | grep -E "k_compare_(fast|fastm2|wide2)|kernels,"

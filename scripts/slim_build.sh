#!/bin/bash
# Experiment build of libbioem_hip.so with ONE k_compare_fast and/or ONE k_compare_wide2 instantiation (seconds
# instead of minutes); never shipped.  Load it with BIOEM_HIP_LIBRARY=abl/<name>.so.
#   scripts/slim_build.sh <name> [-DBIOEM_SLIM_FAST=10,32,false,1] [-DBIOEM_SLIM_W2=32,21,2,false,1,4] [more -D flags]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 \
  -Wno-unused-value -DBIOEM_SLIM "$@" -Iinclude -o abl/$name.so bioem_amd/csrc/bioem_hip.hip
python scripts/check_code_object.py --so abl/$name.so 2>&1 | grep -E "k_compare_(fast|fastm2|wide2)|kernels,"

#!/bin/bash
# gfx950 disassembly of a built library: scripts/disasm.sh <lib.so> <out.s>
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat "$1"
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
$L/llvm-objdump -d --no-show-raw-insn $T/k.co | c++filt > "$2"
rm -rf $T

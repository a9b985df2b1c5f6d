#!/bin/bash
# Build a variant of the assembly kernel for A/B runs on the GPU box:
#   tools/build_var.sh <name> "<OS13_OPT switches>"  -> tools/var/<name>.hsaco   (git-ignored; travels with gpurun snapshots)
# Run it with SS_HSACO=$PWD/tools/var/<name>.hsaco (the product loads sonicsim_amd/lib/k_os13_gfx950.hsaco).
set -e
LLVM=/opt/rocm/lib/llvm/bin
mkdir -p tools/var
OS13_OPT="$2" python tools/gen_asm/os13.py > /tmp/var_$1.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/var_$1.s -o /tmp/var_$1.o
$LLVM/ld.lld -shared /tmp/var_$1.o -o tools/var/$1.hsaco
echo tools/var/$1.hsaco

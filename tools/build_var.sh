#!/bin/bash
# Build a profiling variant of the assembly kernel: tools/build_var.sh <name> "<OS13_OPT switches>"  -> sonicsim_amd/lib/var_<name>.hsaco
set -e
LLVM=/opt/rocm/lib/llvm/bin
OS13_OPT="$2" python tools/gen_asm/os13.py > /tmp/var_$1.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/var_$1.s -o /tmp/var_$1.o
$LLVM/ld.lld -shared /tmp/var_$1.o -o sonicsim_amd/lib/var_$1.hsaco
echo sonicsim_amd/lib/var_$1.hsaco

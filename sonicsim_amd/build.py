"""Build libsonicsim_hip.so in-tree (hipcc cross-compiles gfx950 without a GPU present).

    python -m sonicsim_amd.build [--force]

The shared object lands in ``sonicsim_amd/lib/`` (git-ignored, but it travels with gpurun
snapshots).  There is no JIT cache and no pip install: the product loads exactly this file.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG, "csrc", "sonicsim_hip.hip")
DEPS = [SRC, os.path.join(PKG, "csrc", "tvfir_core.h"), os.path.join(PKG, "csrc", "plan.h"), os.path.join(PKG, "csrc", "tvfir13.h"),
        os.path.join(PKG, "csrc", "stream13.h"), os.path.join(PKG, "csrc", "hostpipe.h"),
        os.path.join(os.path.dirname(PKG), "include", "sonicsim_hip.h")]
OUT = os.path.join(PKG, "lib", "libsonicsim_hip.so")
ASM_GEN = os.path.join(os.path.dirname(PKG), "tools", "gen_asm", "os13.py")
ASM_SRC = os.path.join(PKG, "csrc", "k_os13_gfx950.s")          # generated, committed (reviewable)
ASM_OUT = os.path.join(PKG, "lib", "k_os13_gfx950.hsaco")
ASM_OUT_DYNQ = os.path.join(PKG, "lib", "k_os13_gfx950_dynq.hsaco")
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def is_stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def llvm_bin(name: str) -> str:
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root and os.path.exists(os.path.join(root, "lib", "llvm", "bin", name)):
            return os.path.join(root, "lib", "llvm", "bin", name)
    raise RuntimeError(f"{name} not found under /opt/rocm/lib/llvm/bin")


def _assemble(src: str, out: str, verbose: bool):
    obj = out + ".o"
    cmds = [[llvm_bin("clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}", "-c", src, "-o", obj],
            [llvm_bin("ld.lld"), "-shared", obj, "-o", out + ".tmp"]]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    os.remove(obj)


def build_asm(force: bool = False, verbose: bool = False) -> str:
    """Generate, assemble and link the hand-scheduled render kernel (gfx950 code objects): the dynamic-queue build (``OS13_OPT=dynq``,
    the default: ``csrc/k_os13_gfx950_dynq.s``) and the static-list build (``csrc/k_os13_gfx950.s``; ``ss_set_task_queue(0)``)."""
    def stale(out):
        return force or not os.path.exists(out) or os.path.getmtime(ASM_GEN) > os.path.getmtime(out)
    os.makedirs(os.path.dirname(ASM_OUT), exist_ok=True)
    if stale(ASM_OUT):
        with open(ASM_SRC + ".tmp", "w") as f:
            subprocess.run([sys.executable, ASM_GEN], check=True, stdout=f, env={**os.environ, "OS13_OPT": ""})
        os.replace(ASM_SRC + ".tmp", ASM_SRC)
        _assemble(ASM_SRC, ASM_OUT, verbose)
    if stale(ASM_OUT_DYNQ):
        src = ASM_SRC[:-len(".s")] + "_dynq.s"                       # csrc/k_os13_gfx950_dynq.s: the committed listing of the default kernel
        with open(src + ".tmp", "w") as f:
            subprocess.run([sys.executable, ASM_GEN], check=True, stdout=f, env={**os.environ, "OS13_OPT": "dynq"})
        os.replace(src + ".tmp", src)
        _assemble(src, ASM_OUT_DYNQ, verbose)
    return ASM_OUT


def build(force: bool = False, verbose: bool = False, extra=None, out=None) -> str:
    """extra / out: A/B builds with additional hipcc flags into another file (tools/); the product build uses neither."""
    if out is None:
        build_asm(force=force, verbose=verbose)
    if out is None and not force and not is_stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value", "-Wno-unused-result", *(extra or []), SRC, "-o", (out or OUT) + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace((out or OUT) + ".tmp", out or OUT)
    return out or OUT


TUNING_OUT = os.path.join(PKG, "lib", "libsonicsim_hip_tuning.so")


def build_tuning(force: bool = False, verbose: bool = False) -> str:
    """The library the tools/ scripts use: the product source with -DSS_TUNING_KNOBS, i.e. with the environment switches of the
    experiments (SS_HSACO, SS_TRACE_FILE, SS_DYNQ, SS_OS_GEOM, SS_HOP_RS, ...).  The product library reads no environment variable
    (one test switch aside).  Select it with `_lib.use_library(path)` (bench.py: --lib / BENCH_LIB)."""
    build_asm(force=force, verbose=verbose)
    srcs = [SRC] + [os.path.join(os.path.dirname(SRC), f) for f in os.listdir(os.path.dirname(SRC)) if f.endswith(".h")]
    if not force and os.path.exists(TUNING_OUT) and all(os.path.getmtime(f) <= os.path.getmtime(TUNING_OUT) for f in srcs):
        return TUNING_OUT
    return build(force=True, verbose=verbose, extra=["-DSS_TUNING_KNOBS"], out=TUNING_OUT)


if __name__ == "__main__":
    if "--tuning" in sys.argv:
        print(build_tuning(force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))

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
DEPS = [SRC, os.path.join(PKG, "csrc", "tvfir_core.h"), os.path.join(PKG, "csrc", "plan.h"),
        os.path.join(os.path.dirname(PKG), "include", "sonicsim_hip.h")]
OUT = os.path.join(PKG, "lib", "libsonicsim_hip.so")
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


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value", "-Wno-unused-result", SRC, "-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

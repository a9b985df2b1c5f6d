"""The boundary is a C-ABI so that a host that is NOT Python can bind it: tests/c_abi/host_render.c is such a host (plain C99, no torch in
the process).  CPU test: it compiles and links against include/sonicsim_hip.h + the in-tree library.  GPU test: it runs -- two renders
through host pointers (pageable and pinned output) against a double-precision direct-form evaluation of SonicSim_moving.py:86-94, and the
error convention."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "host_render.c")
EXE = os.path.join(ROOT, "tests", "c_abi", "host_render")


def _build(built_lib):
    libdir = os.path.dirname(built_lib)
    cmd = ["gcc", "-O2", "-std=c99", "-Wall", "-Werror", SRC, "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lsonicsim_hip", "-lm",
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_c_host_compiles_and_links(built_lib):
    exe = _build(built_lib)
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    for sym in ("ss_convolve_moving_seg_f32", "ss_host_alloc", "ss_host_free", "ss_last_error", "ss_version", "ss_shutdown"):
        assert sym in out, sym


@pytest.mark.gpu
def test_c_host_renders_on_the_gpu(gpu, built_lib):
    exe = _build(built_lib)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "all cases within 1e-4" in r.stdout, (r.stdout, r.stderr)

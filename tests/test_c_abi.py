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


GSRC = os.path.join(ROOT, "tests", "c_abi", "host_gather.c")
GEXE = os.path.join(ROOT, "tests", "c_abi", "host_gather")


def _build_gather(built_lib):
    """the two-process gather host also talks to the HIP runtime itself (it owns its device buffers): plain C against hip_runtime_api.h"""
    libdir = os.path.dirname(built_lib)
    cmd = ["gcc", "-O2", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", GSRC, "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           "-L", libdir, "-lsonicsim_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined", "-o", GEXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return GEXE


def test_c_gather_host_compiles_and_links(built_lib):
    exe = _build_gather(built_lib)
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    for sym in ("ss_gather_create", "ss_gather_attach", "ss_gather_slot", "ss_gather_put", "ss_gather_flush", "ss_gather_close"):
        assert sym in out, sym


@pytest.mark.gpu
def test_two_c_processes_gather_scenes_through_an_ipc_handle(gpu, built_lib, tmp_path):
    """root + peer, two processes of a C host on one GPU: the peer's scenes arrive in the root's IPC-shared array through the copy engines and
    the root finds every scene bit-identical to its own render (ss_gather_*; SonicSet.py:183-211 sharded)"""
    exe = _build_gather(built_lib)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, role, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for role in ("root", "peer")]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=180)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert [p.returncode for p in procs] == [0, 0], outs
    assert "5 scenes gathered, 0 mismatching" in outs[0] and "delivered" in outs[1], outs

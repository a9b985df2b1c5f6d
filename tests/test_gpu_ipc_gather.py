"""The CU-free gather behind the C-ABI (ss_gather_create / attach / slot / put / wait_src / flush / close; round 6, SURVEY 8b "ss_gather_scenes"):
two PROCESSES on one GPU -- all that a one-GPU box offers, and what a HIP IPC handle needs -- shard seven scenes, the scenes travel to rank 0's
IPC-shared array with the copy engines, rank 0 compares every gathered scene bit for bit with its own render.  Replaces SonicSet.py:183-211's
serial loop for N > 1 without RCCL's send / recv kernels taking compute units from the persistent render kernel."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_two(env_extra, port):
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "workers", "ipc_gather_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    return [p.returncode for p in procs], outs


def test_two_processes_one_gpu_ipc_gather(gpu):
    rcs, outs = _run_two({"GATHER": "ipc"}, 29571)
    assert rcs == [0, 0], "\n".join(outs)
    assert outs[0].count("bits equal") == 2, outs[0]


def test_same_harness_through_the_default_gather(gpu):
    """the same two ranks through SceneGather (gloo point-to-point on one GPU): the harness itself is sound"""
    rcs, outs = _run_two({"GATHER": "rccl"}, 29573)
    assert rcs == [0, 0], "\n".join(outs)

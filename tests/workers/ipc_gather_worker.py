"""One rank of tests/test_gpu_ipc_gather.py: two processes share ONE GPU (control plane: gloo), every rank renders its shard of small scenes and
the scenes travel to rank 0's IPC-shared array through the library's copy-engine gather (ss_gather_*).  Rank 0 re-renders every scene itself
and compares bits.  Exit code 0 = all good."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from sonicsim_amd import ops, parallel  # noqa: E402


def scene_inputs(s, dev):
    rng = np.random.default_rng(700 + s)
    T, P, C, L = 50000, 5, 2, 6000
    x = rng.standard_normal(T).astype(np.float32)
    bank = (rng.standard_normal((P, C, L)) * np.exp(-4 * np.arange(L) / L)).astype(np.float32)
    w = rng.uniform(0.3, 1.8, P - 1)
    seg = np.floor(w / w.sum() * T).astype(np.int64)
    seg[-1] += T - seg.sum()
    return torch.from_numpy(x).to(dev), torch.from_numpy(bank).to(dev), seg, (C, T)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    ops.init(0)
    num = int(os.environ.get("NUM_SCENES", "7"))          # odd: the last rank gets fewer
    kind = os.environ.get("GATHER", "ipc")
    _, _, _, shape = scene_inputs(0, dev)
    for rep in range(2):                                  # two gathers: the second one reuses streams / lanes / rotating buffers
        g = parallel.make_gather(kind, num, shape, device=dev, depth=2)
        for j in range(g.steps()):
            s = g.scene(j)
            out = g.slot(j)
            if s is not None:
                x, bank, seg, _ = scene_inputs(s, dev)
                ops.convolve_moving_seg(x, bank, seg, out=out)
            g.submit(j)
        res = g.finish()
        torch.cuda.synchronize()
        if rank == 0:
            assert res.shape == (num,) + shape
            for s in range(num):
                x, bank, seg, _ = scene_inputs(s, dev)
                want = ops.convolve_moving_seg(x, bank, seg, out=torch.empty(shape, device=dev))
                assert torch.equal(res[s], want), f"scene {s} differs (rep {rep})"
            print(f"rank 0: {num} scenes gathered through '{kind}', bits equal (rep {rep})", flush=True)
        if hasattr(g, "close"):
            g.close()
        else:
            dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

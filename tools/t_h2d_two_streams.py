"""GPU box: does a second copy stream hide the gaps between the pieces of a pipelined host-to-device transfer?  307 MB from pinned memory in 32 MiB (and 16 MiB)
pieces on one stream, alternating over two / three streams, and as one copy (round 4, host-pointer path: the link carries 56 GB/s in one copy, the staged pipeline
reaches 52).  usage: python tools/t_h2d_two_streams.py"""
import time, torch
dev = torch.device("cuda:0")
n = 307200000 // 4
src = torch.empty(n, dtype=torch.float32).pin_memory()
dst = torch.empty(n, dtype=torch.float32, device=dev)
def run(piece_mib, nstreams):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    pe = piece_mib * (1 << 20) // 4
    def once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        i = 0
        for off in range(0, n, pe):
            with torch.cuda.stream(streams[i % nstreams]):
                dst[off:off + pe].copy_(src[off:off + pe], non_blocking=True)
            i += 1
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    once()
    return min(once() for _ in range(7)) * 1e3
def whole():
    def once():
        torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(src, non_blocking=True); torch.cuda.synchronize(); return time.perf_counter() - t0
    once()
    return min(once() for _ in range(7)) * 1e3
print("one copy        %.3f ms" % whole(), flush=True)
for piece in (32, 16, 8):
    for ns in (1, 2, 3):
        print("pieces of %2d MiB on %d stream(s): %.3f ms" % (piece, ns, run(piece, ns)), flush=True)

"""ss_convolve_scene_f32: all renders of a scene (moving and static sources) in ONE persistent launch are bit-identical to the
separate calls (SonicSet.py:61-94 renders them one after the other)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(gpu, T, Ps, C, L, seed):
    rng = np.random.default_rng(seed)
    xs, banks, segs, peaks = [], [], [], []
    for P in Ps:
        xs.append(torch.from_numpy((rng.standard_normal(T) * 0.1).astype(np.float32)).to(gpu))
        h = (rng.standard_normal((P, C, L)) * np.exp(-4.0 * np.arange(L) / L)[None, None, :]).astype(np.float32)
        banks.append(torch.from_numpy(h).to(gpu))
        if P > 1:
            cuts = np.sort(rng.integers(0, T + 1, P - 2))
            segs.append(np.diff(np.concatenate([[0], cuts, [T]])).astype(np.int64))
            peaks.append(torch.tensor([float(np.abs(h).max())], dtype=torch.float32, device=gpu))
        else:
            segs.append(None)
            peaks.append(None)
    return xs, banks, segs, peaks


@pytest.mark.parametrize("static_lists", [False, True])
def test_scene_launch_is_bit_identical_to_separate_renders(gpu, static_lists):
    from sonicsim_amd import ops
    ops.set_task_queue(not static_lists)
    try:
        for (T, Ps, C, L, seed) in ((150000, (9, 14, 6, 1, 1), 3, 9000, 1), (70000, (1, 5), 2, 20000, 2), (200000, (11,), 4, 12000, 3),
                                    (90000, (1, 1, 1), 8, 8200, 4), (60000, (7, 1, 3), 2, 3000, 5), (40000, (1, 12), 3, 4096, 6), (30000, (4, 1), 1, 300, 7)):
            xs, banks, segs, peaks = _inputs(gpu, T, Ps, C, L, seed)
            want = []
            for x, b, sg, pk in zip(xs, banks, segs, peaks):
                want.append(ops.convolve_fixed(x, b[0]) if sg is None else ops.convolve_moving_seg(x, b, sg, bank_peak=pk))
            stack = torch.full((len(Ps), C, T), float("nan"), dtype=torch.float32, device=gpu)
            got = ops.convolve_scene(xs, [b if sg is not None else b[0] for b, sg in zip(banks, segs)], segs, peaks=peaks,
                                     outs=[stack[i] for i in range(len(Ps))])
            for i in range(len(Ps)):
                assert got[i].data_ptr() == stack[i].data_ptr()
                assert torch.equal(got[i], want[i]), (Ps, i)
            again = ops.convolve_scene(xs, banks, segs, peaks=peaks)                 # fresh outputs, (1, C, L) static banks: same bits
            assert all(torch.equal(a, w) for a, w in zip(again, want))
    finally:
        ops.set_task_queue(True)


def test_scene_launch_argument_checks(gpu):
    from sonicsim_amd import ops
    xs, banks, segs, _ = _inputs(gpu, 60000, (4, 1), 2, 9000, 7)
    with pytest.raises(ValueError):
        ops.convolve_scene(xs, banks, [segs[0][:-1], None])                           # wrong number of segment lengths
    with pytest.raises(ValueError):
        ops.convolve_scene(xs, banks, [segs[0] + 1, None])                            # sum != T
    with pytest.raises(ValueError):
        ops.convolve_scene(xs, [banks[0][:, :, :100].contiguous(), banks[1][:, :, :100].contiguous()], segs)  # filters too short for the transform engines (L <= 128)
    with pytest.raises(ValueError):
        ops.convolve_scene(xs * 5, banks * 5, segs * 5)                               # more than 8 sources


def test_two_host_threads_on_their_own_streams(gpu):
    """Threading contract of the C-ABI (ctypes releases the GIL; one lock per device context, a stream switch serialises against
    the previous stream): two host threads, each on its own HIP stream, issue renders, bank syntheses with the tracked peak and scene
    launches concurrently -- every result equals the single-threaded one bit for bit (plan staging, the K1 arrival ticket and the
    workspace are shared state)."""
    import threading
    from sonicsim_amd import ops
    xs, banks, segs, peaks = _inputs(gpu, 120000, (7, 5, 1), 3, 9000, 11)
    rng = np.random.default_rng(5)
    delay = rng.integers(5, 200, (6, 3)).astype(np.int32)
    dgain = rng.uniform(0.2, 1.5, (6, 3)).astype(np.float32)
    want_r = [ops.convolve_moving_seg(xs[i], banks[i], segs[i], bank_peak=peaks[i]) for i in (0, 1)]
    want_s = ops.convolve_scene(xs, banks, segs, peaks=peaks)
    want_b, want_p = ops.rir_bank_synth(delay, dgain, 12000, 16000, 0.5, 42, device=gpu, return_peak=True)
    torch.cuda.synchronize()
    errors = []

    def worker(k):
        try:
            st = torch.cuda.Stream(device=gpu)
            with torch.cuda.stream(st):
                for it in range(12):
                    y = ops.convolve_moving_seg(xs[k], banks[k], segs[k], bank_peak=peaks[k])
                    b, p = ops.rir_bank_synth(delay, dgain, 12000, 16000, 0.5, 42, device=gpu, return_peak=True)
                    sc = ops.convolve_scene(xs, banks, segs, peaks=peaks)
                    st.synchronize()
                    if not torch.equal(y, want_r[k]):
                        errors.append((k, it, "render"))
                    if not (torch.equal(b, want_b) and float(p) == float(want_p)):
                        errors.append((k, it, "bank / peak"))
                    if not all(torch.equal(a, w) for a, w in zip(sc, want_s)):
                        errors.append((k, it, "scene"))
        except Exception as e:                                   # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_scene_renderer_prefetched_provider_same_bits(gpu):
    """round 4: the NEXT scene's five K1 launches run on a second stream beside the current scene's loudness / mix kernels
    (double-buffered banks).  Same mixes and gains, bit for bit, as the in-line form -- also when a prefetched scene is skipped or a
    scene arrives that was not announced."""
    from sonicsim_amd import SonicSim_audio as A
    from sonicsim_amd import pipeline
    specs = [pipeline.make_scene_spec(gpu, scene=s, config="tiny") for s in range(3)]
    order = [(0, 11), (1, 12), (2, 13), (0, 14), (1, 15)]

    def run(prefetch, announce=None):
        r = pipeline.SceneRenderer(specs[0], gpu)
        outs = []
        for i, (si, seed) in enumerate(order):
            np.random.seed(100 + i)                                   # the loudness targets come from the global NumPy stream
            nxt = None
            if prefetch and i + 1 < len(order):
                a = order[i + 1] if announce is None else announce[i]
                nxt = (specs[a[0]], a[1]) if a is not None else None
            mix, gains = r.render(specs[si], seed=seed, sirs=(1.5,), snr=12.0, next_scene=nxt, sync=False)
            outs.append((mix.clone(), gains.clone()))
        torch.cuda.synchronize()
        return outs

    base = run(False)
    for variant in (run(True), run(True, announce=[(1, 12), (0, 99), None, (1, 15), None])):   # a wrong announcement, a missing one
        for (m0, g0), (m1, g1) in zip(base, variant):
            assert torch.equal(m0, m1) and torch.equal(g0, g1)
    assert all(torch.isfinite(m).all() for m, _ in base) and float(base[0][0].abs().max()) > 0
    g = A.lufs_gains_from_result(base[0][1].cpu().numpy())
    np.random.seed(100)
    mix_s, gains_s = pipeline.SceneRenderer(specs[0], gpu).render(specs[0], seed=11, sirs=(1.5,), snr=12.0)      # default: synchronous, Python floats
    assert len(g) == 5 and torch.equal(mix_s, base[0][0]) and np.allclose(gains_s, g, rtol=0, atol=0)


def test_static_render_is_stored_not_added(gpu):
    """a static source on the assembly engine has one task per (channel, output block): the kernel STORES y and the spectra kernel skips the
    zero fill -- a buffer full of NaN comes back complete and equal to a fresh render, for lengths that end inside a block / a hop"""
    from oracle import moving as O
    from sonicsim_amd import ops
    rng = np.random.default_rng(11)
    for (T, C, L) in ((16000, 1, 4096), (70001, 3, 8200), (4095, 2, 9000), (300000, 8, 48000)):
        x = torch.from_numpy((rng.standard_normal(T) * 0.1).astype(np.float32)).to(gpu)
        h = torch.from_numpy((rng.standard_normal((C, L)) * np.exp(-4.0 * np.arange(L) / L)[None, :]).astype(np.float32)).to(gpu)
        want = ops.convolve_fixed(x, h, path="asm")
        out = torch.full((C, T), float("nan"), dtype=torch.float32, device=gpu)
        got = ops.convolve_fixed(x, h, path="asm", out=out)
        assert got.data_ptr() == out.data_ptr() and torch.isfinite(got).all()
        assert torch.equal(got, want)
        ref = O.convolve_fixed_receiver(x.cpu().numpy(), h.cpu().numpy())
        assert O.rel_rms(got.cpu().numpy(), ref) <= 1e-4

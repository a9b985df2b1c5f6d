"""One full SonicSet sample on the GPU (BASELINE config 3): the audio half of ``process_single``
(``SonicSim-SonicSet/SonicSet.py:61-101``) plus the downstream 2-speaker mix
(``separation/look2hear/datas/movingdatamodule.py:105-124``), on synthetic inputs.

    3 x  generate_rir_combination (:61-63)  -> interpolate_moving_audio (:77-79)      moving speakers
    2 x  render_ir / create_custom_arrayir (:86-91) -> convolve_fixed_receiver (:93-94)  noise, music
    5 x  get_lufs_norm_audio with targets -17/-17/-17/-24/-29 LUFS (:97-101)
    1 x  SIR/SNR mix of speakers {1,2} + noise

Everything stays in HBM; only O(P) schedules, O(blocks) loudness gating and scalars touch the host.
File I/O (torchaudio.save at :102-106, json at :108-136) is out of scope for the timed path (``formats.py`` writes it).

Two forms:
  * ``make_scene_inputs`` + ``render_sonicset_sample``: the banks are resident inputs (what the parity tests compare stage by stage).
  * ``make_scene_spec`` + ``render_scene``: the banks are PRODUCED inside the scene's work by the RIR provider (K1, with the
    peak of SonicSim_audio.py:398 tracked by the generator and deferred into the render) -- config 4's unit of work: a rank
    cannot keep 64 scenes x 3 banks (59 GB) resident, and in SonicSet the provider runs once per scene anyway (SonicSet.py:61-63).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import SonicSim_audio as A
from . import mixing, ops, synth

LUFS_TARGETS = (-17, -17, -17, -24, -29)          # SonicSet.py:97-101


@dataclass
class SceneInputs:
    """Resident inputs of one scene (built once, outside any timed region)."""
    speakers: list        # 3 x (x (T,), bank (P,C,L), seg_len (P-1,), peak (1,) or None)
    statics: list         # 2 x (x (T,), h (C,L))
    fs: int


def make_scene_inputs(device, scene=0, config="cfg2", defer_norm=True) -> SceneInputs:
    """defer_norm=True: the banks stay as generated and their global peak (SonicSim_audio.py:398), tracked by the generator,
    is applied inside the render (``bank_peak=``); False materialises ``bank / peak`` like generate_rir_combination does."""
    import torch
    spk = []
    for s in range(3):
        sc = synth.make_scene(config, scene=scene * 8 + s)
        seg = synth.scene_segments(sc, scene * 8 + s)
        bank, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=device, return_peak=True)
        if not defer_norm:
            ops.divide_by_(bank, peak)                              # generate_rir_combination (:398), one pass
            peak = None
        spk.append((torch.from_numpy(sc.x).to(device), bank, seg, peak))
    st = []
    for s in range(2):
        sc = synth.make_scene(config, scene=scene * 8 + 4 + s, P=1)
        h = ops.rir_bank_synth(sc.delay[:1], sc.dgain[:1], sc.L, sc.fs, sc.rt60, sc.bank_seed, device=device)[0]
        st.append((torch.from_numpy(sc.x).to(device), h))              # static IRs are NOT normalised (SonicSet.py:86-94)
    return SceneInputs(spk, st, sc.fs)


def _normalise_and_mix(stack, fs, nstem, sirs, snr, out, sync=True):
    # row U for all stems in one device call (targets drawn in stem order like successive reference calls);
    # (C,T) stems in place of the reference's transposed (T,C)
    presums = None
    if sync:
        nstack, gains = A.get_lufs_norm_audio_batch(stack, fs, LUFS_TARGETS[:nstem], allow_many_channels=True, sync=True)
    else:       # round 5: the energy of every normalised stem rides on the pass that writes it -- the mix does not measure the stems again
        nstack, gains, sq = A.get_lufs_norm_audio_batch(stack, fs, LUFS_TARGETS[:nstem], allow_many_channels=True, sync=False, want_sumsq=True,
                                                        cross_speakers=2)
        presums = (sq[:2], sq[3:4]) + ((sq[nstem:nstem + 1],) if sq.numel() == nstem + 1 else ())     # round 6: with the two speakers' cross sum the mix is ONE pass
    normed = [nstack[j] for j in range(nstem)]
    noise = normed[3][None]                                                      # 2-speaker separation mixture; the reference scales the interferer in place
    mix, _ = mixing.mix_sources(nstack[:2], noise, np.asarray(sirs, dtype=np.float32), float(snr), out=out,      # (:113) -- here the stems stay as
                                keep_speakers=True, presums=presums)                                              # normalised (no clone): row M
    return mix, normed, gains


def render_sonicset_sample(inp: SceneInputs, sirs=(0.0,), snr=15.0, lufs_seed=None):
    """Returns (mix (C,T), stems [5 x (C,T)], gains) -- all torch tensors on the device."""
    if lufs_seed is not None:
        np.random.seed(lufs_seed)
    import torch
    x0, bank0 = inp.speakers[0][0], inp.speakers[0][1]
    nstem = len(inp.speakers) + len(inp.statics)
    stack = torch.empty((nstem, bank0.shape[1], x0.shape[-1]), dtype=torch.float32, device=x0.device)   # renders land in one stack
    i = 0
    for (x, bank, seg, peak) in inp.speakers:                                                  # rows (G+)I+V
        ops.convolve_moving_seg(x, bank, seg, out=stack[i], bank_peak=peak)
        i += 1
    for (x, h) in inp.statics:                                                                 # row F
        ops.convolve_fixed(x, h, out=stack[i])
        i += 1
    return _normalise_and_mix(stack, inp.fs, nstem, sirs, snr, None)


# ------------------------------------------------------------------------------------------------ config 4's unit of work
@dataclass
class SceneSpec:
    """Host-side description of one scene + its dry signals in HBM (everything the RIR provider and the renders need)."""
    speakers: list        # 3 x (x (T,) device, delay (P,C) i32 device, dgain (P,C) f32 device, seg_len (P-1,) host, rt60)
    statics: list         # 2 x (x (T,) device, delay (1,C) device, dgain (1,C) device, rt60)
    T: int
    C: int
    L: int
    fs: int


def make_scene_spec(device, scene=0, config="cfg2") -> SceneSpec:
    """Trajectories, direct-path geometry, segment lengths (host NumPy RNG, like SonicSim_moving.py:32-39) and the five dry
    signals of one scene.  Host work + H2D: done outside timed regions."""
    import torch
    spk, st = [], []
    for s in range(3):
        sc = synth.make_scene(config, scene=scene * 8 + s)
        spk.append((torch.from_numpy(sc.x).to(device), torch.from_numpy(sc.delay).to(device), torch.from_numpy(sc.dgain).to(device),
                    synth.scene_segments(sc, scene * 8 + s), sc.rt60))          # the provider's geometry sits in HBM next to the dry signal
    for s in range(2):
        sc = synth.make_scene(config, scene=scene * 8 + 4 + s, P=1)
        st.append((torch.from_numpy(sc.x).to(device), torch.from_numpy(sc.delay[:1].copy()).to(device), torch.from_numpy(sc.dgain[:1].copy()).to(device),
                   sc.rt60))
    return SceneSpec(spk, st, sc.T, sc.C, sc.L, sc.fs)


class SceneRenderer:
    """Renders SceneSpecs with a reused stem stack (5, C, T).  one_launch=True (default): the five banks / IRs of the scene are produced
    first and all five renders run as ONE persistent launch; False: render by render, the three speakers' banks reuse one block of the
    caching allocator one after the other.

    Round 4 -- the provider of the NEXT scene runs beside the loudness / mix kernels of the current one: ``render(spec, seed, ...,
    next_scene=(spec2, seed2))`` enqueues the five K1 launches of scene 2 on a second HIP stream right behind scene 1's render launch
    (which needs every CU to itself) -- they fill the GPU while scene 1's loudness and mix kernels, short and latency bound, run on the
    first stream -- into the other half of a double-buffered set of banks; the following ``render(spec2, seed2)`` finds them ready.
    Same kernels, same bits; a scene whose banks were not prefetched generates them in line as before."""

    def __init__(self, spec: SceneSpec, device, one_launch=True):
        import torch
        self.device = device
        self.one_launch = one_launch          # all five renders of a scene in ONE persistent launch (bit-identical to the separate calls)
        self.stack = torch.empty((5, spec.C, spec.T), dtype=torch.float32, device=device)
        self._k1_stream = None                # created on first use
        self._sets = [None, None]             # double-buffered banks + peaks: set i = ([banks], [peaks])
        self._set_next = 0
        self._ready = None                    # (key, set index, event) of the prefetched scene
        self._set_free = [None, None]         # event after which set i may be overwritten (its render has finished)

    # ---- the provider (K1 x 5) of one scene into bank set `si`
    def _ensure_set(self, spec, si):
        """Bank set `si` with this scene's shapes.  ALWAYS called with the caller's stream current (never inside the side-stream context): the
        caching allocator then owns the blocks on behalf of that stream, and the side stream's use is declared with record_stream.  A set is
        replaced (other shapes) only after everything that may still read or write it has finished (the old tensors used to be dropped
        while a render or the generator on the other stream could still be using them)."""
        import torch
        P = int(spec.speakers[0][1].shape[0])
        if self._sets[si] is not None and self._sets[si][0][0].shape == (P, spec.C, spec.L):
            return
        if self._sets[si] is not None:
            torch.cuda.synchronize(self.device)      # rare (the scene shapes changed): nothing of the old set is in flight when it goes back to the allocator
            self._set_free[si] = None
        banks = [torch.empty((P, spec.C, spec.L), dtype=torch.float32, device=self.device) for _ in range(3)]
        banks = banks + [torch.empty((1, spec.C, spec.L), dtype=torch.float32, device=self.device) for _ in range(2)]
        peaks = [torch.empty(1, dtype=torch.float32, device=self.device) for _ in range(3)]
        self._sets[si] = (banks, peaks)

    def _provide(self, spec, seed, si, background=False):
        banks, peaks = self._sets[si]
        geoms = [(delay, dgain, rt60, (seed * 8 + k) & 0x7FFFFFFF) for k, (x, delay, dgain, seg, rt60) in enumerate(spec.speakers)] + \
                [(delay, dgain, rt60, (seed * 8 + 4 + k) & 0x7FFFFFFF) for k, (x, delay, dgain, rt60) in enumerate(spec.statics)]
        ops.rir_bank_synth_batch(geoms, spec.L, spec.fs, banks, list(peaks) + [None, None], background=background)      # the five banks of the scene: ONE launch
        return banks, peaks

    def prefetch(self, spec: SceneSpec, seed: int):
        """Enqueue the provider of (spec, seed) on the side stream, behind everything enqueued so far on the current stream."""
        import torch
        if not (self.one_launch and spec.L > 128):
            return
        if self._k1_stream is None:
            self._k1_stream = torch.cuda.Stream(device=self.device)
        si = self._set_next
        self._set_next ^= 1
        cur = torch.cuda.current_stream(self.device)
        self._ensure_set(spec, si)                                 # (allocated under the caller's stream)
        self._k1_stream.wait_stream(cur)                           # behind the current scene's render launch
        if self._set_free[si] is not None:
            self._k1_stream.wait_event(self._set_free[si])
        for t in self._sets[si][0] + self._sets[si][1]:
            t.record_stream(self._k1_stream)                       # written by the side stream, read by the caller's
        with torch.cuda.stream(self._k1_stream):
            self._provide(spec, seed, si, background=True)       # beside the current scene's loudness / mix: leave them wave slots
            ev = torch.cuda.Event()
            ev.record(self._k1_stream)
        self._ready = ((spec, int(seed)), si, ev)                  # the spec OBJECT (kept alive), not its id(): an id can be reused after a collection

    def render(self, spec: SceneSpec, seed: int, sirs=(0.0,), snr=15.0, out=None, sync=True, next_scene=None):
        """K1 x 3 (bank + tracked peak) -> moving renders with the normalisation deferred; K1 x 2 -> static renders; loudness of
        the five stems in one call; mix of speakers {1, 2} + noise into ``out`` (or a fresh tensor).  Returns (mix, gains).
        sync=True (default, the reference's behaviour): the five loudness gains come back as Python floats (one synchronisation per
        scene).  sync=False: nothing in a scene waits for the GPU -- gains is the float64 device record tensor (5, 4) the caller reads when
        it writes the scene's metadata (``SonicSim_audio.lufs_gains_from_result``); a scene generator that pipelines scenes wants this.
        next_scene = (spec, seed) of the scene rendered next: its provider runs on the side stream beside this scene's loudness / mix."""
        import torch
        if self.one_launch and spec.L > 128:
            # the provider first (five K1 launches, or the prefetched set), then ALL five renders in one persistent launch (ss_convolve_scene_f32)
            cur = torch.cuda.current_stream(self.device)
            if self._ready is not None and self._ready[0][0] is spec and self._ready[0][1] == int(seed):
                _, si, ev = self._ready
                self._ready = None
                cur.wait_event(ev)
                banks, peaks = self._sets[si]
            else:
                si = self._set_next
                self._set_next ^= 1
                self._ensure_set(spec, si)
                if self._set_free[si] is not None:
                    cur.wait_event(self._set_free[si])
                banks, peaks = self._provide(spec, seed, si)
            xs = [sp[0] for sp in spec.speakers] + [st[0] for st in spec.statics]
            segs = [sp[3] for sp in spec.speakers] + [None, None]
            ops.convolve_scene(xs, banks, segs, peaks=list(peaks) + [None, None], outs=[self.stack[i] for i in range(len(xs))])
            done = torch.cuda.Event()
            done.record(cur)
            self._set_free[si] = done                               # the set may be refilled once this render has finished
            if next_scene is not None:
                self.prefetch(*next_scene)
        else:
            i = 0
            for k, (x, delay, dgain, seg, rt60) in enumerate(spec.speakers):
                bank, peak = ops.rir_bank_synth(delay, dgain, spec.L, spec.fs, rt60, (seed * 8 + k) & 0x7FFFFFFF, device=self.device, return_peak=True)
                ops.convolve_moving_seg(x, bank, seg, out=self.stack[i], bank_peak=peak)
                del bank                                                   # back to the caching allocator: the next speaker reuses the block
                i += 1
            for k, (x, delay, dgain, rt60) in enumerate(spec.statics):
                h = ops.rir_bank_synth(delay, dgain, spec.L, spec.fs, rt60, (seed * 8 + 4 + k) & 0x7FFFFFFF, device=self.device)[0]
                ops.convolve_fixed(x, h, out=self.stack[i])
                i += 1
        mix, _, gains = _normalise_and_mix(self.stack, spec.fs, 5, sirs, snr, out, sync=sync)
        return mix, gains

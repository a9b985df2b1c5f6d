"""One full SonicSet sample on the GPU (BASELINE config 3): the audio half of ``process_single``
(``SonicSim-SonicSet/SonicSet.py:61-101``) plus the downstream 2-speaker mix
(``separation/look2hear/datas/movingdatamodule.py:105-124``), on synthetic inputs.

    3 x  generate_rir_combination (:61-63)  -> interpolate_moving_audio (:77-79)      moving speakers
    2 x  render_ir / create_custom_arrayir (:86-91) -> convolve_fixed_receiver (:93-94)  noise, music
    5 x  get_lufs_norm_audio with targets -17/-17/-17/-24/-29 LUFS (:97-101)
    1 x  SIR/SNR mix of speakers {1,2} + noise

Everything stays in HBM; only O(P) schedules, O(blocks) loudness gating and scalars touch the host.
File I/O (torchaudio.save at :102-106, json at :108-136) is out of scope for the timed path.
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
    speakers: list        # 3 x (x (T,), bank (P,C,L), seg_len (P-1,))
    statics: list         # 2 x (x (T,), h (C,L))
    fs: int


def make_scene_inputs(device, scene=0, config="cfg2") -> SceneInputs:
    import torch
    spk = []
    for s in range(3):
        sc = synth.make_scene(config, scene=scene * 8 + s)
        seg = synth.scene_segments(sc, scene * 8 + s)
        bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=device)
        ops.peak_normalize_(bank)                                   # generate_rir_combination (:398)
        spk.append((torch.from_numpy(sc.x).to(device), bank, seg))
    st = []
    for s in range(2):
        sc = synth.make_scene(config, scene=scene * 8 + 4 + s, P=1)
        h = ops.rir_bank_synth(sc.delay[:1], sc.dgain[:1], sc.L, sc.fs, sc.rt60, sc.bank_seed, device=device)[0]
        st.append((torch.from_numpy(sc.x).to(device), h))              # static IRs are NOT normalised (SonicSet.py:86-94)
    return SceneInputs(spk, st, sc.fs)


def render_sonicset_sample(inp: SceneInputs, sirs=(0.0,), snr=15.0, lufs_seed=None):
    """Returns (mix (C,T), stems [5 x (C,T)], gains) -- all torch tensors on the device."""
    if lufs_seed is not None:
        np.random.seed(lufs_seed)
    import torch
    x0, bank0, _ = inp.speakers[0]
    nstem = len(inp.speakers) + len(inp.statics)
    stack = torch.empty((nstem, bank0.shape[1], x0.shape[-1]), dtype=torch.float32, device=x0.device)   # renders land in one stack
    i = 0
    for (x, bank, seg) in inp.speakers:                                                        # rows I+V
        ops.convolve_moving_seg(x, bank, seg, out=stack[i])
        i += 1
    for (x, h) in inp.statics:                                                                 # row F
        ops.convolve_fixed(x, h, out=stack[i])
        i += 1
    # row U for all stems in one device call (targets drawn in stem order like successive reference calls);
    # (C,T) stems in place of the reference's transposed (T,C)
    nstack, gains = A.get_lufs_norm_audio_batch(stack, inp.fs, LUFS_TARGETS[:nstem], allow_many_channels=True)
    normed = [nstack[j] for j in range(nstem)]
    spk = nstack[:2].clone()                                                     # 2-speaker separation mixture (the mix scales interferers in place, :113)
    noise = normed[3][None]
    mix, _ = mixing.mix_sources(spk, noise, np.asarray(sirs, dtype=np.float32), float(snr))   # row M
    return mix, normed, gains

"""CPU oracle for the SonicSim moving-source render hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s CPU legs may
import it, and there only as the checker / the timed CPU baseline.  The product path
(``sonicsim_amd``) never imports this package and fails loudly when the HIP library is
missing.

Parity status.  "Pinned" = checked bit for bit against golden vectors produced by
importing the reference's own modules, unmodified, under stubs for the packages this image lacks
(``tests/golden/make_golden.py``, ``make_golden_aux.py`` -> ``tests/golden/*.npz``):
  * rows I, V, W, F -- PINNED: ``oracle/moving.py`` vs ``SonicSim_moving.py`` (g1-g8; ``tests/test_oracle_golden.py``).
  * row G (all_pairs / clip_all / stack / global peak normalise) -- PINNED: ``oracle/rir_synth.py`` vs
    ``SonicSim_audio.generate_rir_combination`` with a stubbed provider returning ragged IRs (g9).
  * rows M, N2 (rms dB, SIR / SNR mix, dataset items with crop + silence rejection, overlap_audio) -- PINNED:
    ``oracle/mix.py``, ``oracle/datamodule.py`` vs both ``movingdatamodule.py`` files (g10; ``tests/test_oracle_golden_aux.py``).
  * row N3 -- layout logic PINNED by goldens from the reference's ``create_long_audio`` / ``create_background_audio`` (g11);
    the resampler (``oracle/resample.py``) is UNPINNED: torchaudio is absent, its published algorithm is restated.
  * row U (LUFS) -- PARITY UNPINNED: pyloudnorm 0.1.1 is an absent third-party dependency
    (``SonicSim-SonicSet/ss-2.0.yaml:201``); ``oracle/loudness.py`` restates its published
    BS.1770-4 algorithm (incl. its input-dtype behaviour) and is anchored on the call sites ``SonicSim_audio.py:68-86``.
  * row R (RIR provider) -- PARITY UNPINNED: arithmetic lives in the closed-source
    habitat-sim RLR audio propagation library.  ``oracle/rir_synth.py`` is the NumPy
    definition of the synthetic bank that the HIP generator must reproduce.
"""

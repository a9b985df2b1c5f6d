"""CPU oracle for the SonicSim moving-source render hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and there only as the checker / the timed CPU baseline.  The product path
(``sonicsim_amd``) never imports this package and fails loudly when the HIP library is
missing.

Parity status (see DESIGN.md "Oracle"):
  * rows I, V, W, F (setup_dynamic_interp / convolve_moving_receiver /
    interpolate_moving_audio / convolve_fixed_receiver) -- PINNED: the restatement in
    ``oracle/moving.py`` is checked against golden vectors produced by importing the
    reference's own ``SonicSim_moving.py`` in the authoring container
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
  * row G (clip_all / stack / global peak normalise) -- restated from
    ``SonicSim-SonicSet/SonicSim_audio.py:111-127,391-398`` (not importable: needs habitat).
  * row M (rms dB / SIR / SNR mix) -- restated from
    ``separation/look2hear/datas/movingdatamodule.py:29-32,105-124`` (module not importable).
  * row U (LUFS) -- PARITY UNPINNED: pyloudnorm 0.1.1 is an absent third-party dependency
    (``SonicSim-SonicSet/ss-2.0.yaml:201``); ``oracle/loudness.py`` restates its published
    BS.1770-4 algorithm and is anchored on the call sites ``SonicSim_audio.py:68-86``.
  * row R (RIR provider) -- PARITY UNPINNED: arithmetic lives in the closed-source
    habitat-sim RLR audio propagation library.  ``oracle/rir_synth.py`` is the NumPy
    definition of the synthetic bank that the HIP generator must reproduce.
"""

"""sonicsim_amd -- MI355X-native moving-source audio renderer behind SonicSim's own function names.

Drop-in modules (same names / signatures as ``SonicSim-SonicSet/*.py`` of JusperLee/SonicSim):
    sonicsim_amd.SonicSim_moving   setup_dynamic_interp / convolve_fixed_receiver /
                                   convolve_moving_receiver / interpolate_moving_audio
    sonicsim_amd.SonicSim_audio    generate_rir_combination / lufs_norm / get_lufs_norm_audio / ...
    sonicsim_amd.SonicSim_rir      render_ir / create_custom_arrayir / render_rir_parallel (synthetic provider)
``sonicsim_amd/compat`` holds top-level aliases so ``import SonicSim_moving`` resolves here when that
directory is placed on ``sys.path`` ahead of the reference's.

All arithmetic runs in ``lib/libsonicsim_hip.so`` (hand-written gfx950 HIP, C-ABI in
``include/sonicsim_hip.h``).  There is no CPU fallback.  Importing never touches the GPU.
"""
__version__ = "0.1.0"

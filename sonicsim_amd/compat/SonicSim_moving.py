"""Top-level alias so that ``import SonicSim_moving`` (as SonicSet.py:16-21 does) resolves to the MI355X implementation when
``sonicsim_amd/compat`` precedes the reference directory on sys.path.  Names this package does not implement are passed
through from the reference's own ``SonicSim_moving.py`` when that module is importable (see _passthrough.py); the accelerated
functions always win."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _passthrough import reference_names as _reference_names  # noqa: E402

_ref, REFERENCE_SOURCE = _reference_names("SonicSim_moving")
globals().update(_ref)
from sonicsim_amd.SonicSim_moving import *  # noqa: F401,F403,E402
from sonicsim_amd import SonicSim_moving as _impl  # noqa: E402

ACCELERATED = sorted(n for n in dir(_impl) if not n.startswith("_"))
for _n in ACCELERATED:
    globals()[_n] = getattr(_impl, _n)
__all__ = sorted(set(ACCELERATED) | set(_ref))

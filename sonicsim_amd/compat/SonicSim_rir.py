"""Top-level alias so that ``import SonicSim_rir`` (as SonicSet.py:16-21 does) resolves to the MI355X
implementation when ``sonicsim_amd/compat`` precedes the reference directory on sys.path."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
from sonicsim_amd.SonicSim_rir import *  # noqa: F401,F403,E402
from sonicsim_amd import SonicSim_rir as _impl  # noqa: E402

__all__ = [n for n in dir(_impl) if not n.startswith("_")]

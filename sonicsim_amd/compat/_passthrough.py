"""Shared by the three alias modules in this directory: locate the SAME-NAMED module of the reference further down sys.path
(``SonicSim-SonicSet/``), import it under a private name and hand back its public names, so that everything the MI355X package
does not accelerate (``Scene``, ``get_nav_idx``, ``save_trace_gif``, ...: Habitat-side code, out of scope) still resolves when
``sonicsim_amd/compat`` shadows the reference's modules.  If the reference is not on the path -- or cannot be imported because
Habitat / magnum / torchaudio are missing (ImportError) -- only the accelerated names exist, exactly as before; any other exception
raised while the reference module executes is a bug in that file and is re-raised."""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def reference_names(module_name):
    for entry in sys.path:
        if not entry or os.path.abspath(entry) == HERE:
            continue
        cand = os.path.join(entry, module_name + ".py")
        if os.path.isfile(cand):
            alias = "_sonicsim_reference_" + module_name
            if alias in sys.modules:
                mod = sys.modules[alias]
            else:
                spec = importlib.util.spec_from_file_location(alias, cand)
                mod = importlib.util.module_from_spec(spec)
                sys.modules[alias] = mod
                try:
                    spec.loader.exec_module(mod)
                except ImportError as e:                    # (incl. ModuleNotFoundError) missing Habitat / magnum / torchaudio: the
                    del sys.modules[alias]                  # accelerated subset still works.  Anything ELSE -- a genuine bug in a user's
                    return {}, f"{cand}: {type(e).__name__}: {e}"     # modified reference file -- propagates instead of degrading silently
                except BaseException:
                    sys.modules.pop(alias, None)
                    raise
            return {k: v for k, v in vars(mod).items() if not k.startswith("_")}, cand
    return {}, None

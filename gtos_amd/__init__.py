"""gtos_amd: the gtos graph-transformer hot path on MI355X (gfx950), behind the reference's module API.

``install_reference_names()`` makes the reference's own import lines resolve to this package: the reference imports its
model code by bare module name from the script directory (generator/generator.py:6-9 ``from graph_transformer import ...``,
``from encoder import ...``, ``from transformer import ...``, ``from decoder import ...``; generator/train.py ``from generator
import Generator``; generator/work.py ``from search import ...``), and Python consults ``sys.modules`` before the path, so one
line at the top of train.py / work.py --

    import gtos_amd; gtos_amd.install_reference_names()

-- swaps in the HIP-backed modules without touching any other import.  Nothing is installed implicitly.
"""
import importlib
import sys

REFERENCE_MODULE_NAMES = ("graph_transformer", "transformer", "encoder", "decoder", "generator", "search")


def install_reference_names(names=REFERENCE_MODULE_NAMES, force=False):
    """Alias ``gtos_amd.<name>`` as top-level module ``<name>`` for every name given.  An already imported module of that name
    (the reference's own file) is only replaced with ``force=True``.  Returns the list of names installed."""
    done = []
    for n in names:
        if n in sys.modules and not force and not getattr(sys.modules[n], "__name__", "").startswith("gtos_amd."):
            continue
        sys.modules[n] = importlib.import_module("gtos_amd." + n)
        done.append(n)
    return done


def uninstall_reference_names(names=REFERENCE_MODULE_NAMES):
    for n in names:
        m = sys.modules.get(n)
        if m is not None and getattr(m, "__name__", "").startswith("gtos_amd."):
            del sys.modules[n]

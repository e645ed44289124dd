"""Importable alias of the package directory ``cycle-gan-rna_amd/`` (a hyphen is
not valid in a Python module name).  Submodules resolve to files in that
directory: ``import cycle_gan_rna_amd.engine`` loads
``cycle-gan-rna_amd/engine.py``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cycle-gan-rna_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f

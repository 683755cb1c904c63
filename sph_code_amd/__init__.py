"""Import shim: the product package lives in the directory `sph-code_amd/` (the name the
project layout prescribes); a hyphen cannot appear in a Python module name, so this package
points its search path there.  `import sph_code_amd.compat as nsc` is the drop-in for
`import navier_stokes_cleaned as nsc` (sph/code_running.py:12)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "sph-code_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f

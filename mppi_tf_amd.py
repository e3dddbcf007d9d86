"""Import alias. The package directory is `mppi-tf_amd/` (the repo's layout uses the hyphenated
name); a hyphen is not importable, so `import mppi_tf_amd` loads that directory as a package."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mppi-tf_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

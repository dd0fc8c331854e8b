"""Import shim: `two-towers-overlords_amd/` is not a legal Python identifier, so
this module loads that directory as the package `two_towers_overlords_amd`."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "two-towers-overlords_amd")
_spec = _u.spec_from_file_location(__name__, _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

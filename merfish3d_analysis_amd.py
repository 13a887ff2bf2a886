"""Importable alias for the hyphenated package directory ``merfish3d-analysis_amd/``.

``import merfish3d_analysis_amd`` (and any ``merfish3d_analysis_amd.x.y``
sub-module import) resolves to the very same module objects as
``importlib.import_module("merfish3d-analysis_amd.x.y")`` - no code lives here.
"""

import importlib
import importlib.abc
import importlib.util
import sys

_REAL = "merfish3d-analysis_amd"
_ALIAS = __name__


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real_name):
        self._real_name = real_name

    def create_module(self, spec):
        return importlib.import_module(self._real_name)

    def exec_module(self, module):  # already executed under its real name
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != _ALIAS and not fullname.startswith(_ALIAS + "."):
            return None
        real_name = _REAL + fullname[len(_ALIAS):]
        real_spec = importlib.util.find_spec(real_name)
        if real_spec is None:
            return None
        spec = importlib.util.spec_from_loader(fullname, _AliasLoader(real_name))
        spec.submodule_search_locations = real_spec.submodule_search_locations
        return spec


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

sys.modules[_ALIAS] = importlib.import_module(_REAL)

"""Importable alias of the package directory `alphazero-openspiel_amd/` (a hyphen is not a valid
Python identifier, the project layout asks for that directory name).  This shim loads that
directory as the package `alphazero_openspiel_amd` and replaces itself in sys.modules."""
import importlib.util
import os
import sys

_real = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir,
                                      "alphazero-openspiel_amd"))
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_real, "__init__.py"),
                                               submodule_search_locations=[_real])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

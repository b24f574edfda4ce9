"""Import shim: makes the package directory ``vmp-for-svae_amd/`` importable as ``vmp_for_svae_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'vmp-for-svae_amd')
_spec = importlib.util.spec_from_file_location('vmp_for_svae_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['vmp_for_svae_amd'] = _mod
_spec.loader.exec_module(_mod)

"""Import alias: the package directory is ``gcn-vae_amd/`` (not a valid identifier), so
``import gcn_vae_amd`` is served by loading that directory as a package under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gcn-vae_amd')
_spec = importlib.util.spec_from_file_location('gcn_vae_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['gcn_vae_amd'] = _mod
_spec.loader.exec_module(_mod)

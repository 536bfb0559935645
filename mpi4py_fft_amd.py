"""Import alias.  The package directory is named ``mpi4py-fft_amd`` (project layout); a hyphen
cannot appear in a Python module name, so ``import mpi4py_fft_amd`` lands here and this stub
replaces itself in ``sys.modules`` with the real package loaded from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mpi4py-fft_amd')
_spec = importlib.util.spec_from_file_location(
    'mpi4py_fft_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['mpi4py_fft_amd'] = _mod
_spec.loader.exec_module(_mod)

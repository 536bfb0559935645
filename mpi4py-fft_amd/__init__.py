"""mpi4py_fft_amd -- MI355X-native engine behind mpi4py-fft's PFFT hot path.

Drop-in for ``PFFT / DistArray / newDistArray`` (mpi4py_fft/__init__.py:22-26) with the arrays in
HBM: hand-written HIP kernels (libgfft.so) for the serial per-axis FFTs, pack/unpack and
dealiasing copies; RCCL all-to-all over xGMI for the global transposes.  See DESIGN.md.
"""
__version__ = '0.1.0'

from . import _lib
from . import comm
from .array import DeviceArray, asdevice, empty, zeros, host_empty
from .distarray import DistArray, newDistArray, Function
from .mpifft import PFFT
from .pencil import Pencil, Subcomm, Transfer
from .libfft import FFT
from . import fftw
from .fftw import fftlib
from . import spectral
from .io import HDF5File, NCFile        # (generate_xdmf is not rebuilt and not exported: a script that needs it fails at its import line, not at run time)

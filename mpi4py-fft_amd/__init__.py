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
# (generate_xdmf: the name is exported as the reference exports it, mpi4py_fft/__init__.py:26, so that drop-in import lines
# keep working; calling it raises NotImplementedError with the way out -- it is host-side XML outside the hot path)
from .io import HDF5File, NCFile, generate_xdmf
from . import selftest

"""Import-compatibility surface of mpi4py_fft.io (HDF5File, NCFile, generate_xdmf).

Parallel HDF5 / NetCDF output is a storage feature outside the transform path this package
accelerates: the names exist so that `from mpi4py_fft_amd import HDF5File`
in a ported script still imports, and using them says what to do instead -- bring a field to the
host with ``np.asarray(u)`` (or one global slice with ``u.get(gslice)``) and hand it to the
reference's writers, which take numpy arrays.
"""


def _unavailable(name):
    class _Stub:
        def __init__(self, *args, **kwargs):
            raise NotImplementedError(
                '%s: file I/O is outside the MI355X transform path; copy the field to the host '
                '(np.asarray(u), or u.get(gslice) for one global slice) and use mpi4py_fft.io' % name)
    _Stub.__name__ = _Stub.__qualname__ = name
    return _Stub


HDF5File = _unavailable('HDF5File')
NCFile = _unavailable('NCFile')


def generate_xdmf(*args, **kwargs):
    raise NotImplementedError('generate_xdmf: file I/O is outside the MI355X transform path')

"""Stand-in for the h5py module (TEST INFRASTRUCTURE; h5py is not in the image).

Implements the handful of h5py calls `mpi4py_fft_amd.io.HDF5File` makes -- File(name, mode),
require_group, require_dataset, attrs, item access by path, region assignment -- on nested dicts
pickled to the file name at close(), so that ranks opening the file one after the other see each
other's blocks, as they would with the real library.  Checks the writer's LOGIC (dataset paths,
global shapes, regions, turn taking), not the HDF5 byte format.
"""
import os
import pickle

import numpy as np


class _Attrs(dict):
    def create(self, key, value):
        self[key] = np.asarray(value)


class Dataset:
    def __init__(self, shape, dtype, data=None):
        self.data = np.zeros(shape, dtype=dtype)
        if data is not None:
            self.data[...] = data
        self.attrs = _Attrs()

    shape = property(lambda self: self.data.shape)
    dtype = property(lambda self: self.data.dtype)

    def __setitem__(self, key, value):
        self.data[key] = value

    def __getitem__(self, key):
        return self.data[key]


class Group:
    def __init__(self):
        self.items = {}
        self.attrs = _Attrs()

    def _walk(self, path, create):
        node = self
        for part in [p for p in path.split('/') if p]:
            if part not in node.items:
                if not create:
                    raise KeyError(path)
                node.items[part] = Group()
            node = node.items[part]
        return node

    def require_group(self, path):
        g = self._walk(path, True)
        assert isinstance(g, Group)
        return g

    def require_dataset(self, name, shape, dtype, data=None, **kw):
        d = self.items.get(name)
        if d is None:
            d = self.items[name] = Dataset(tuple(shape), dtype, data)
        assert isinstance(d, Dataset) and d.shape == tuple(shape) and d.dtype == np.dtype(dtype), \
            'require_dataset: existing dataset differs'
        return d

    def __getitem__(self, path):
        return self._walk(path, False)

    def __contains__(self, path):
        try:
            self._walk(path, False)
            return True
        except KeyError:
            return False

    def keys(self):
        return self.items.keys()


class File(Group):
    def __init__(self, name, mode='r', **kw):
        Group.__init__(self)
        assert 'driver' not in kw, 'the writer must not ask for the MPI-IO driver'
        self.name, self.mode = name, mode
        if mode in ('r', 'r+') or (mode == 'a' and os.path.exists(name)):
            with open(name, 'rb') as f:
                root = pickle.load(f)
            self.items, self.attrs = root.items, root.attrs

    def close(self):
        if self.mode != 'r':
            root = Group()
            root.items, root.attrs = self.items, self.attrs
            with open(self.name, 'wb') as f:
                pickle.dump(root, f)

"""A minimal in-memory stand-in for the part of the h5py API the checkpoint
code uses (File / Group / Dataset / attrs), persisted with pickle.  h5py is
not installed in this image; the layout written through this shim is the
reference's, so the same code drives the real library where it exists."""

import os
import pickle

import numpy as np


class Dataset:
    def __init__(self, data, maxshape=None):
        self.data = np.array(data)
        self.maxshape = maxshape

    @property
    def shape(self):
        return self.data.shape

    @property
    def dtype(self):
        return self.data.dtype

    def resize(self, shape):
        new = np.zeros(shape, dtype=self.data.dtype)
        n = min(len(new), len(self.data)) if new.ndim else 0
        if new.ndim:
            new[:n] = self.data[:n]
        self.data = new

    def __setitem__(self, key, value):
        self.data[key] = value

    def __getitem__(self, key):
        return self.data[key]

    def __array__(self, dtype=None, copy=None):
        return self.data if dtype is None else self.data.astype(dtype)

    def __len__(self):
        return len(self.data)


class Attrs(dict):
    """Attribute store with h5py's conversion rules: values pass through
    numpy; what has no HDF5 equivalent (objects, None, ragged sequences) is
    refused with the exception types h5py raises -- the reference's emulator
    ``write`` relies on that to skip such entries (neural.py:131-137) -- and
    reads return numpy scalars / arrays, str for text."""

    def __setitem__(self, key, value):
        arr = np.asarray(value)          # ragged -> ValueError, as h5py
        if arr.dtype.kind == 'O':
            raise TypeError("Object dtype dtype('O') has no native HDF5 "
                            "equivalent")
        dict.__setitem__(self, key, arr)

    def __getitem__(self, key):
        arr = dict.__getitem__(self, key)
        if not isinstance(arr, np.ndarray):
            return arr
        if arr.ndim == 0:
            return str(arr[()]) if arr.dtype.kind in 'US' else arr[()]
        return arr

    def items(self):
        return [(k, self[k]) for k in self.keys()]


class Group:
    def __init__(self):
        self.attrs = Attrs()
        self.items_ = {}

    def create_group(self, name):
        self.items_[name] = Group()
        return self.items_[name]

    def create_dataset(self, name, data=None, maxshape=None):
        self.items_[name] = Dataset(data, maxshape)
        return self.items_[name]

    def __getitem__(self, name):
        return self.items_[name]

    def __contains__(self, name):
        return name in self.items_

    def keys(self):
        return self.items_.keys()


class File(Group):
    def __init__(self, path, mode='r'):
        super().__init__()
        self.path, self.mode = str(path), mode
        if mode == 'x':
            if os.path.exists(self.path):
                raise FileExistsError(self.path)
        else:
            with open(self.path, 'rb') as f:
                root = pickle.load(f)
            self.attrs, self.items_ = root.attrs, root.items_

    def close(self):
        if self.mode in ('x', 'r+', 'w', 'a'):
            root = Group()
            root.attrs, root.items_ = self.attrs, self.items_
            with open(self.path, 'wb') as f:
                pickle.dump(root, f)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def tree_schema(group, prefix=''):
    """Flat description of a group tree: what a reader of the file can rely
    on -- names (indices replaced by '#'), kind and rank of every attribute
    and dataset, dtype kind, ``maxshape`` of the resizable datasets.  Used to
    hold the checkpoint layout against the one the reference's ``write``
    methods emit (tests/golden/h5_layout.json, make_golden_h5.py)."""
    import re

    def norm(name):
        return re.sub(r'\d+', '#', name)

    def kind(value):
        arr = np.asarray(value)
        k = arr.dtype.kind
        if k in 'US':
            k = 'str'
        elif k in 'iu':
            k = 'int'
        elif k == 'f':
            k = 'float'
        elif k == 'b':
            k = 'bool'
        elif k == 'V':
            k = 'struct' + str(tuple(
                np.dtype(arr.dtype.fields[f][0]).kind
                for f in arr.dtype.names))
        return k, arr.ndim

    out = {}
    for key, val in group.attrs.items():
        k, ndim = kind(val)
        out[prefix + '@' + norm(key)] = dict(kind=k, ndim=ndim)
    for key, item in group.items_.items():
        path = prefix + '/' + norm(key)
        if isinstance(item, Group):
            out[path] = dict(kind='group')
            out.update(tree_schema(item, path))
        else:
            k, ndim = kind(item.data)
            out[path] = dict(
                kind=k, ndim=ndim,
                maxshape=None if item.maxshape is None else
                [None if m is None else int(m) for m in item.maxshape])
    return out

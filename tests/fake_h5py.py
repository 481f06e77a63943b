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


class Group:
    def __init__(self):
        self.attrs = {}
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

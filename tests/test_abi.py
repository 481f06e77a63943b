"""The C-ABI library builds, loads and exports every symbol that
include/nautilus_hip.h declares (no GPU needed, no compute calls)."""

import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, 'include', 'nautilus_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nb_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_all_declared_symbols():
    from nautilus_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run make / __graft_entry__.build()'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.exported_symbols()) == names
    header = open(os.path.join(ROOT, 'include', 'nautilus_hip.h')).read()
    declared = int(re.search(r'#define NB_ABI_VERSION (\d+)', header).group(1))
    assert lib.nb_abi_version() == declared == _lib.ABI_VERSION == 6


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    from nautilus_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libnautilus_hip.so')
    with pytest.raises(RuntimeError):
        _lib.load()

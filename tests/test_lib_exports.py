"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/far_hip.h declares."""
import os
import re

from far_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'far_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(far_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_exported_and_bound():
    lib = _lib.load()
    names = _declared()
    assert names, 'no declarations parsed'
    for n in names:
        assert hasattr(lib, n), f'{n} declared in far_hip.h but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature in far_amd/_lib.py'
    assert sorted(_lib.SIGNATURES) == names
    assert lib.far_abi_version() == _lib.EXPECTED_ABI        # load() refuses any other library


def test_workspace_query_needs_no_gpu():
    lib = _lib.load()
    assert lib.far_dual_softmax_workspace_bytes(2, 4800, 4800) > 0

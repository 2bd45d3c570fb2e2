"""Thin torch-tensor front ends for the C ABI in include/far_hip.h, one submodule per kernel family (round 6: the former 1 738-line
far_amd/ops.py, split):

    _base      streams, pointers, workspaces, the activation-range state
    packs      weight images (PackedConv, PackedWino, the pack table, PackCache)
    conv       K9 / K17 / K10 convolutions, K19 BatchNorm, K8 upsample-add, K7 affine epilogues
    linear     K9 in Linear mode (plain, gather, k|v-state, q-apply)
    attention  K5 linear attention, K6 LayerNorm
    coarse     K1
    fine       K3, K13, K14
    head       K2 + contraction, K4 front end, K11, K15, K12

PyTorch supplies device memory and the current HIP stream; all arithmetic of these ops happens in libfar_hip.so.  Every op raises on
CPU tensors -- there is no eager fallback.

The rest of the package (and the tests) use the FLAT namespace `far_amd.ops.<name>`, including module-level switches that are
assigned from outside (`ops.USE_WINO = False`, `ops.activation_overflowed = probe`, far_amd.flags.target).  This module re-exports
every top-level name of the submodules and forwards assignments to the submodule(s) that own the name, so a switch flipped through
`far_amd.ops` is the switch its kernel family reads.
"""
import sys
import types

from . import _base, packs, conv, linear, attention, coarse, fine, head

_SUBMODULES = (_base, packs, conv, linear, attention, coarse, fine, head)
_SKIP = {'ctypes', 'os', 'threading', 'torch', '_lib', 'flags'}
for _m in _SUBMODULES:
    for _k, _v in vars(_m).items():
        if not _k.startswith('__') and _k not in _SKIP and not isinstance(_v, types.ModuleType):
            globals()[_k] = _v
from .. import _lib, flags  # noqa: E402,F401  (ops._lib / ops.flags were reachable on the flat module too)


class _FlatNamespace(types.ModuleType):
    def __setattr__(self, name, value):
        for m in _SUBMODULES:
            if name in m.__dict__:
                m.__dict__[name] = value          # the owner, and every submodule that imported the name from it
        super().__setattr__(name, value)


sys.modules[__name__].__class__ = _FlatNamespace

"""far_amd.ops.packs: weight images: PackedConv (K9), PackedWino (K17), the whole-model pack table, PackCache (one family of the torch-tensor front ends for the C ABI in include/far_hip.h; far_amd/ops/__init__.py
re-exports everything under the flat far_amd.ops namespace the rest of the package uses)."""
import ctypes
import os
import threading

import torch

from .. import _lib, flags
from ._base import _p, _stream, tensor_version


def train_pack(cache, name, weight, bias=None, split=True):
    """The K9 image of a Linear layer's weight for the training forward (re-packed when the weight's version changes)."""
    return cache.get((name, split), [weight] + ([bias] if bias is not None else []), lambda: PackedConv(weight, None, bias, split=split),
                     refresh=(lambda pc: pc.refresh(weight)) if bias is None else None)

def train_pack_t(cache, name, weight, bias=None, split=True):
    """The transposed image (dgrad): the same tensor read through strides, with the forward image's scale (same maximum)."""
    fwd = train_pack(cache, name, weight, bias, split)             # first: the transposed image borrows its (refreshed) scale
    pt = cache.get((name, 'T', split), [weight], lambda: PackedConv(weight, split=split, dgrad=True, pack_scale=fwd.pack_scale),
                   refresh=lambda pc: pc.refresh(weight))
    return pt.follow_scale(fwd, weight)

class PackedConv:
    """Weights of one convolution / linear layer in K9's packed split-fp16 image, plus the folded epilogue vectors."""

    def __init__(self, weight, scale=None, shift=None, split=True, stride=1, dgrad=False, pack_scale=None):
        """weight: (Cout, Cin, k, k) or (Cout, Cin).  dgrad=True packs the image of the layer's input-gradient convolution --
        channels exchanged, taps reversed (a Linear layer: the transposed weight) -- read from the SAME tensor through strides.
        pack_scale: the two device floats of another image of the same weight (its maximum is the same): skips the reduction."""
        lib = _lib.load()
        w = weight.detach()
        if w.dim() == 2:
            w = w[:, :, None, None]
        Cout, Cin, kh, kw = w.shape
        if kh != kw or kh not in (1, 3):
            raise _lib.FarHipError(f'K9 supports 1x1 and 3x3 kernels, got {kh}x{kw}')
        w = w.contiguous().float()
        if stride not in (1, 2) or (stride == 2 and kh != 3):
            raise _lib.FarHipError('K9 supports stride 1, and stride 2 for 3x3 kernels')
        T = kh * kh
        if dgrad:
            view = (T, Cin * T, -1 if T > 1 else 0, T - 1)          # (s_co, s_ci, s_tap, offset of tap 0) of the dgrad image
            Cin, Cout = Cout, Cin
        else:
            view = (Cin * T, T, 1, 0)
        self.Cin, self.Cout, self.ksize, self.split, self.stride = Cin, Cout, kh, bool(split), stride
        nbytes = lib.far_conv_packed_bytes(Cin, Cout, kh, stride, int(self.split))
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        # the power-of-two weight scale 2^w_exp (max |w| 2^w_exp in [2^13, 2^14)) is chosen on the device: no host read of the
        # weights, so re-packing after every optimizer step costs two small launches and no synchronisation
        self._own_scale = pack_scale is None
        self.pack_scale = torch.empty(2, dtype=torch.float32, device=w.device) if pack_scale is None else pack_scale   # { 2^w_exp, 2^-(w_exp + 4) }
        self._view, self._wshape = view, tuple(w.shape)
        self._base = None if scale is None else scale.detach().float().contiguous()
        self.scale = torch.empty(Cout, dtype=torch.float32, device=w.device)      # base scale x 2^-(w_exp + 4), written by the pack kernel
        self.shift = None if shift is None else shift.detach().float().contiguous()
        self._wino, self._wino_stale = None, False
        self._pack(w)

    def _pack(self, w):
        lib = _lib.load()
        if self._own_scale:
            _lib.check(lib.far_weight_scale_f32(_p(w, torch.float32), w.numel(), _p(self.pack_scale), _stream()), 'far_weight_scale_f32')
        v = self._view
        rc = lib.far_conv_pack_view_scaled_f32(ctypes.c_void_p(w.data_ptr() + 4 * v[3]), v[0], v[1], v[2], self.Cin, self.Cout, self.ksize,
                                               self.stride, int(self.split), _p(self.pack_scale), _p(self.packed),
                                               _p(self._base) if self._base is not None else None, _p(self.scale), _stream())
        _lib.check(rc, 'far_conv_pack_view_scaled_f32')
        self._w = w                                               # keeps the (possibly temporary) contiguous weight alive until the pack ran

    def follow_scale(self, owner, weight):
        """For an image that borrows another image's pack_scale (dgrad / transposed images): when the owner was REBUILT rather than
        refreshed (a biased layer, a changed stamp) it holds a new scale tensor and this image would keep packing with the orphaned,
        never-updated one -- rebind to the owner's current tensor and re-pack.  Returns self."""
        if not self._own_scale and self.pack_scale.data_ptr() != owner.pack_scale.data_ptr():
            self.pack_scale = owner.pack_scale
            self.refresh(weight)
            PACK_TABLE.dirty = True                     # the device table holds the old scale pointer
        return self

    def refresh(self, weight):
        """Re-pack in place after the weight changed (an optimizer step): the same buffers, one launch (+ the scale reduction when
        this image owns it; an image that borrows another's pack_scale must be refreshed after that one).  Epilogue scale / shift
        vectors passed at construction are kept as they were."""
        w = weight.detach()
        if w.dim() == 2:
            w = w[:, :, None, None]
        if tuple(w.shape) != self._wshape or w.dtype != torch.float32 or not w.is_contiguous() or w.device != self.packed.device:
            raise _lib.FarHipError('PackedConv.refresh: the weight changed shape, dtype, layout or device')
        self._pack(w)
        self.invalidate_wino()
        return self

    def invalidate_wino(self):
        """The weight changed: the K17 image (if one was built) holds the old weights.  It is re-packed in place at its next use
        (wino()); every path that re-packs this image -- refresh() and the whole-model table (_PackTable.refresh_all) -- ends here."""
        if self._wino:
            self._wino_stale = True

    def wino(self):
        """The K17 image of the same layer (built at the first inference launch that can use it, from the weight this image was
        packed from); None for layers K17 does not serve (1x1, stride 2, plain-fp16 operands, dgrad images, channel counts not
        divisible by four)."""
        if self._wino is None:
            ok = (self.ksize == 3 and self.stride == 1 and self.split and self._view[3] == 0 and self.Cin % 4 == 0 and self.Cout % 4 == 0
                  and self._w.dim() == 4)
            self._wino = PackedWino(self._w, self._base, self.shift) if ok else False
            self._wino_stale = False
        elif self._wino and self._wino_stale:
            self._wino.refresh(self._w)                           # self._w shares the parameter's storage: the current weights
            self._wino_stale = False
        return self._wino or None

class PackedWino:
    """Weights of one stride-1 3x3 convolution in K17's Winograd image (U = G g G^T, split fp16 planes), plus the folded epilogue
    vectors; the same constructor meaning as PackedConv (scale / shift: the inference BatchNorm as a per-channel affine map)."""

    ksize, stride, split = 3, 1, True

    def __init__(self, weight, scale=None, shift=None):
        lib = _lib.load()
        w = weight.detach()
        if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
            raise _lib.FarHipError(f'K17 is a 3x3 kernel, got a weight of shape {tuple(w.shape)}')
        w = w.contiguous().float()
        self.Cout, self.Cin = int(w.shape[0]), int(w.shape[1])
        if self.Cin % 4:
            raise _lib.FarHipError('K17 needs Cin % 4 == 0')
        self.packed = torch.empty(lib.far_wino_packed_bytes(self.Cin, self.Cout), dtype=torch.uint8, device=w.device)
        self.pack_scale = torch.empty(2, dtype=torch.float32, device=w.device)       # { 2^w_exp, 2^-(w_exp + 4) }
        self._base = None if scale is None else scale.detach().float().contiguous()
        self.scale = torch.empty(self.Cout, dtype=torch.float32, device=w.device)
        self.shift = None if shift is None else shift.detach().float().contiguous()
        self._wshape = tuple(w.shape)
        self.refresh(w)

    def refresh(self, weight):
        """(Re-)packs the image from `weight` into the same buffers: two launches, no allocation."""
        lib = _lib.load()
        w = weight.detach()
        if tuple(w.shape) != self._wshape or w.dtype != torch.float32 or not w.is_contiguous() or w.device != self.packed.device:
            raise _lib.FarHipError('PackedWino.refresh: the weight changed shape, dtype, layout or device')
        _lib.check(lib.far_weight_scale_f32(_p(w, torch.float32), w.numel(), _p(self.pack_scale), _stream()), 'far_weight_scale_f32')
        rc = lib.far_wino_pack_view_scaled_f32(_p(w), 9 * self.Cin, 9, 1, self.Cin, self.Cout, _p(self.pack_scale), _p(self.packed),
                                               _p(self._base) if self._base is not None else None, _p(self.scale), _stream())
        _lib.check(rc, 'far_wino_pack_view_scaled_f32')
        self._w = w
        return self

WINO_MIN_ACT_EXP = 0        # K17 splits its operands unscaled (|a| <= 16376): used while the activation exponent is >= 0

WINO_MIN_PIXELS = 1024      # per image; below, a 16x16-output workgroup tile is mostly padding

class _PackTable:
    """Every refreshable weight image of the process (training: PackedConv objects whose cache entry depends on the weight
    alone), re-packed together after an optimizer step: far_pack_table_run = two launches for all of them instead of two per
    image.  Entries are weak: an image lives as long as the PackCache of its module does."""

    def __init__(self):
        self.entries = []          # (weakref(cache), key, weakref(weight), weakref(pc))
        self.table = None          # (device table tensor, n, [(weakref(cache), key, weakref(weight), weakref(pc))]): weak, like entries
        self.dirty = True
        self._skip = 0             # stale lookups still to come in the step for which the per-entry path was chosen

    def register(self, cache, key, weight, pc):
        import weakref
        self.entries.append((weakref.ref(cache), key, weakref.ref(weight), weakref.ref(pc)))
        self.dirty = True

    def _live(self):
        out = []
        for e in self.entries:
            cache, w, pc = e[0](), e[2](), e[3]()
            if cache is not None and w is not None and pc is not None and cache._store.get(e[1], (None, None))[1] is pc:
                out.append((cache, e[1], w, pc))
        return out

    def _build(self, live):
        import weakref
        lib = _lib.load()
        dev = live[0][3].packed.device
        live = [e for e in live if e[3].packed.device == dev and e[2].is_contiguous() and e[2].dtype == torch.float32]
        owner = {}
        for i, (_, _, _, pc) in enumerate(live):
            if pc._own_scale:
                owner[pc.pack_scale.data_ptr()] = i
        keep = [e for e in live if e[3].pack_scale.data_ptr() in owner]
        owner = {pc.pack_scale.data_ptr(): i for i, (_, _, _, pc) in enumerate(keep) if pc._own_scale}
        keep = [e for e in keep if e[3].pack_scale.data_ptr() in owner]          # (a borrower whose owner dropped out goes too)
        n = len(keep)
        if n == 0 or n > 4096:
            return None
        items = (_lib.PackItem * n)()
        for i, (_, _, w, pc) in enumerate(keep):
            v, it = pc._view, items[i]
            it.w, it.s_co, it.s_ci, it.s_tap = w.data_ptr() + 4 * v[3], v[0], v[1], v[2]
            it.Cin, it.Cout, it.ksize, it.stride, it.split = pc.Cin, pc.Cout, pc.ksize, pc.stride, int(pc.split)
            it.scale_owner = owner[pc.pack_scale.data_ptr()]
            it.w_all, it.n_all = w.data_ptr(), w.numel()
            it.pack_scale, it.packed = pc.pack_scale.data_ptr(), pc.packed.data_ptr()
            it.base_scale = pc._base.data_ptr() if pc._base is not None else None
            it.scale_vec = pc.scale.data_ptr()
        table = torch.empty(int(lib.far_pack_table_bytes(n)), dtype=torch.uint8, device=dev)
        _lib.check(lib.far_pack_table_build(ctypes.cast(items, ctypes.c_void_p), n, _p(table), _stream()), 'far_pack_table_build')
        # only weak references are kept next to the device table (which holds raw pointers): the table must not pin the weights,
        # images and caches of a model that was deleted; a dead reference found later marks the table dirty
        return table, n, [(weakref.ref(c), k, weakref.ref(w), weakref.ref(pc)) for c, k, w, pc in keep]

    def refresh_all(self):
        """Re-packs every live image whose weight version changed, through the table when most of them did.  Returns True when
        the table ran (the caller's entry is then fresh)."""
        if self._skip > 0:                              # the rest of a step's stale lookups after the per-entry path was chosen
            self._skip -= 1
            return False
        if self.dirty:
            live = self._live()
            self.entries = [e for e in self.entries if e[0]() is not None and e[3]() is not None]
            self.table = self._build(live) if live else None
            self.dirty = False
        if self.table is None:
            return False
        table, n, refs = self.table
        keep, stamps, stale = [], [], 0
        for rc, key, rw, rpc in refs:
            cache, w, pc = rc(), rw(), rpc()
            if cache is None or w is None or pc is None:
                self.dirty = True                       # a model went away: the table's raw pointers are stale, rebuild next time
                self.table = None
                return False
            st = ((w.data_ptr(), tensor_version(w)),)
            hit = cache._store.get(key)
            if hit is None or hit[1] is not pc or hit[0][0][0] != st[0][0]:
                self.dirty = True                       # an entry was replaced or its weight moved: rebuild next time, per-entry now
                return False
            keep.append((cache, key, w, pc))
            stamps.append(st)
            stale += hit[0] != st
        if 2 * stale < n:
            # a few images only (fine-tuning a sub-module): per-entry refresh -- and no second walk over all n entries for each of
            # the other stale images of this step (they each come through here once)
            self._skip = max(stale - 1, 0)
            return False
        _lib.check(_lib.load().far_pack_table_run(_p(table), n, _stream()), 'far_pack_table_run')
        for (cache, key, w, pc), st in zip(keep, stamps):
            cache._store[key] = (st, pc)
            pc.invalidate_wino()                        # the table re-packs K9's images only: K17's follow lazily, in place
        return True

PACK_TABLE = _PackTable()

USE_PACK_TABLE = True       # False: every stale image re-packs itself (two launches each)

class PackCache:
    """K9 weight images keyed by name, rebuilt when any tensor they were derived from changes (in-place update,
    load_state_dict, optimizer step: data_ptr / _version stamp)."""

    def __init__(self):
        self._store = {}

    def get(self, key, tensors, build, refresh=None):
        """refresh(obj): optional in-place update of the stored object when only tensor versions changed (same storage): a
        training step re-packs every weight, and reusing the buffers saves the allocations and two launches per image."""
        stamp = tuple((t.data_ptr(), tensor_version(t)) for t in tensors)
        hit = self._store.get(key)
        if hit is None or hit[0] != stamp:
            same_storage = hit is not None and refresh is not None and tuple(p for p, _ in hit[0]) == tuple(p for p, _ in stamp)
            if same_storage and USE_PACK_TABLE and len(tensors) == 1 and isinstance(hit[1], PackedConv) and PACK_TABLE.refresh_all():
                hit = self._store[key]                   # the whole model's images were re-packed together
                if hit[0] == stamp:
                    return hit[1]
            new = refresh(hit[1]) if same_storage else build()
            hit = (stamp, new)
            self._store[key] = hit
            if refresh is not None and not same_storage and len(tensors) == 1 and isinstance(new, PackedConv):
                PACK_TABLE.register(self, key, tensors[0], new)
        return hit[1]

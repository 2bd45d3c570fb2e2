"""AdamW for a whole model in one launch (K20, far_amd/csrc/adamw_f32.hip): the optimizer of the reference's training step
(mp3d_loftr/src/optimizers/__init__.py:5-16 -> torch.optim.AdamW) with the same constructor and the same arithmetic, for fp32
parameters on one GPU.  Everything else (other dtypes, CPU parameters, amsgrad, maximize) is torch.optim.AdamW's business."""
import ctypes
import math

import numpy as np
import torch

from . import _lib

_CHUNK = 4096


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, maximize=False):
        if amsgrad or maximize:
            raise ValueError('far_amd.optim.AdamW: amsgrad / maximize are not implemented (use torch.optim.AdamW)')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}

    def _table(self, gi, group):
        ps = [p for p in group['params'] if p.requires_grad]
        if not ps:
            return None
        dev = ps[0].device
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise _lib.FarHipError('far_amd.optim.AdamW needs contiguous fp32 parameters on one GPU')
            st = self.state[p]
            if 'exp_avg' not in st:
                st['step'] = torch.tensor(0.0, dtype=torch.float32)      # a CPU float tensor, as torch.optim.AdamW keeps it
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif (st['exp_avg'].dtype != torch.float32 or not st['exp_avg'].is_contiguous() or st['exp_avg'].device != dev
                  or not st['exp_avg_sq'].is_contiguous()):
                raise _lib.FarHipError('far_amd.optim.AdamW: loaded moment tensors must be contiguous fp32 on the parameters\' GPU')
        # the table holds raw pointers: rebuilt when a parameter or a moment tensor was replaced (load_state_dict, .to(), a new model)
        key = tuple((p.data_ptr(), p.numel(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr()) for p in ps)
        t = self._tables.get(gi)
        if t is not None and t['key'] == key:
            return t
        n = len(ps)
        rows = np.zeros((n, 5), dtype=np.int64)                              # {p, g, m, v, n}
        blocks = []
        for i, p in enumerate(ps):
            st = self.state[p]
            rows[i] = (p.data_ptr(), 0, st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel())
            blocks += [(i, c) for c in range((p.numel() + _CHUNK - 1) // _CHUNK)]
        blocks = np.asarray(blocks, dtype=np.int32).reshape(-1, 2)
        lib = _lib.load()
        nbytes = int(lib.far_adamw_table_bytes(n, len(blocks)))
        host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        hv = host.numpy()
        hv[:n * 40] = rows.view(np.uint8).reshape(-1)
        hv[n * 40:n * 40 + blocks.size * 4] = blocks.view(np.uint8).reshape(-1)
        t = dict(key=key, params=ps, n=n, nblocks=len(blocks), host=host, rows=hv[:n * 40].view(np.int64).reshape(n, 5),
                 dev=torch.empty(nbytes, dtype=torch.uint8, device=dev))
        t['dev'].copy_(host, non_blocking=True)
        self._tables[gi] = t
        return t

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            t = self._table(gi, group)
            if t is None:
                continue
            # the gradient pointers change whenever autograd re-allocates them (zero_grad(set_to_none=True)): refresh that column --
            # after the previous step's upload has left the pinned buffer (nothing else orders the host against it)
            if t.get('uploaded') is not None:
                t['uploaded'].synchronize()
            # validate first, mutate after: a raise must leave the optimizer state as it was
            active = []
            for i, p in enumerate(t['params']):
                g = p.grad
                if g is not None:
                    if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse:
                        raise _lib.FarHipError('far_amd.optim.AdamW needs dense contiguous fp32 gradients')
                    active.append((i, p, g))
            steps = {int(self.state[p]['step']) for _, p, _ in active}
            if len(steps) > 1:
                raise _lib.FarHipError('far_amd.optim.AdamW: parameters of one group have different step counts '
                                       '(a parameter skipped earlier steps): use torch.optim.AdamW')
            t['rows'][:, 1] = 0
            if not steps:
                continue
            k = steps.pop() + 1
            for i, p, g in active:
                t['rows'][i, 1] = g.data_ptr()
                st = self.state[p]
                if torch.is_tensor(st['step']):           # the step stays a tensor (torch.optim.AdamW can resume this state dict)
                    st['step'] += 1
                else:
                    st['step'] = k
            n = t['n']
            t['dev'][:n * 40].copy_(t['host'][:n * 40], non_blocking=True)
            if t.get('uploaded') is None:
                t['uploaded'] = torch.cuda.Event()
            t['uploaded'].record(torch.cuda.current_stream(t['dev'].device))
            b1, b2 = group['betas']
            rc = lib.far_adamw_step_f32(ctypes.c_void_p(t['dev'].data_ptr()), n, t['nblocks'], float(group['lr']), float(b1), float(b2),
                                        float(group['eps']), float(group['weight_decay']), 1.0 - b1 ** k, math.sqrt(1.0 - b2 ** k),
                                        ctypes.c_void_p(torch.cuda.current_stream(t['dev'].device).cuda_stream))
            _lib.check(rc, 'far_adamw_step_f32')
        return loss

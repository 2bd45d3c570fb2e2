"""Matcher stages of the drop-in module on the far_amd kernels: CoarseMatching (K1), FinePreprocess (K3a),
FineMatching (K3b).  Same class names, constructor dicts and data-dict keys as the reference's
mp3d_loftr/src/loftr/utils/coarse_matching.py, loftr_module/fine_preprocess.py and utils/fine_matching.py
(one module here: they are thin front ends over three kernels).

CoarseMatching on kernel K1.  Mirrors the interface of mp3d_loftr/src/loftr/utils/coarse_matching.py
(CoarseMatching :56-265): same constructor dict, same data-dict keys written.

FinePreprocess on kernel K3a.  Mirrors mp3d_loftr/src/loftr/loftr_module/fine_preprocess.py:7-59
(same parameters: down_proj, merge_feat).

FineMatching on kernel K3b.  Mirrors mp3d_loftr/src/loftr/utils/fine_matching.py:8-76.
"""
import math

import os

import torch
import torch.nn as nn

from .. import _vendor as ag          # needs_grad + the door to the test-side vendor compositions
from .. import train_glue
from .. import flags, ops


# =====================================================================================================
# coarse level
# =====================================================================================================
class CoarseMatching(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.thr = config['thr']
        self.border_rm = config['border_rm']
        self.train_coarse_percent = config['train_coarse_percent']
        self.train_pad_num_gt_min = config['train_pad_num_gt_min']
        self.match_type = config['match_type']
        if self.match_type != 'dual_softmax':
            raise NotImplementedError("only match_type='dual_softmax' (the FAR configuration) has a kernel")
        self.temperature = config['dsmax_temperature']
        # data['conf_matrix'] (92 MB / pair) is consumed only by the coarse loss and by plotting
        # (loftr_loss.py:307-311); it is materialised when training or when asked for explicitly.
        self.materialize_conf = False
        # K1 arithmetic: 'f16s' (default) split-fp16 operand pairs on the f16 matrix cores -- an fp32-grade similarity,
        # held to the same parity tests as 'f32' (the exact-f32 MFMA kernels); C must be 256 (other widths use 'f32').
        # bf16 = True: bf16 operands (far_coarse_match_bf16), exact on bf16-rounded features, deviation reported as
        # match-set IoU by tests/test_coarse_gpu.py
        self.variant = 'f16s'
        self.bf16 = False

    def forward(self, feat_c0, feat_c1, data, mask_c0=None, mask_c1=None, overlap=None):
        """feat_c0 [N, L, C], feat_c1 [N, S, C]; updates data with conf_matrix (optional), b_ids, i_ids,
        j_ids, gt_mask, m_bids, mkpts0_c, mkpts1_c, mconf (coarse_matching.py:144-147, :243-263).
        overlap (inference on the GPU): see ops.coarse_match -- work enqueued behind K1 while the host waits for the match count."""
        if not feat_c0.is_cuda:
            ag.require('CoarseMatching on CPU tensors')
        if self.training or ag.needs_grad(feat_c0, feat_c1):
            if overlap is not None:
                overlap()
            return self._forward_train(feat_c0, feat_c1, data, mask_c0, mask_c1)
        hw0, hw1 = data['hw0_c'], data['hw1_c']
        valid_hw = None
        as_u8 = lambda m: None if m is None else m.to(torch.uint8).contiguous()
        if 'mask0' in data:
            # mask_border_with_padding (:28-43): per-sample valid extents of the padded coarse masks
            m0, m1 = data['mask0'], data['mask1']
            valid_hw = torch.stack([m0.sum(1).max(-1)[0], m0.sum(-1).max(-1)[0],
                                    m1.sum(1).max(-1)[0], m1.sum(-1).max(-1)[0]], 1).to(torch.int32).contiguous()
        scale = data['hw0_i'][0] / data['hw0_c'][0]
        s0 = data['scale0'].float().contiguous() if 'scale0' in data else None
        s1 = data['scale1'].float().contiguous() if 'scale1' in data else None
        out = ops.coarse_match(feat_c0.float().contiguous(), feat_c1.float().contiguous(), self.temperature,
                               self.thr, self.border_rm, hw0, hw1, scale, as_u8(mask_c0), as_u8(mask_c1),
                               valid_hw, s0, s1, want_conf=self.materialize_conf,
                               variant='bf16' if self.bf16 else (self.variant if feat_c0.shape[-1] == 256 else 'f32'), overlap=overlap)
        mconf = out['mconf']
        data.update({
            'conf_matrix': out['conf_matrix'],
            'b_ids': out['b_ids'], 'i_ids': out['i_ids'], 'j_ids': out['j_ids'],
            'gt_mask': mconf == 0,
            'm_bids': out['b_ids'],          # eval: every match has mconf > thr > 0, nothing is dropped (:257-263)
            'mkpts0_c': out['mkpts0_c'], 'mkpts1_c': out['mkpts1_c'], 'mconf': mconf,
            'match_counts': out['counts'],   # per-pair M (host), reused by the batched solver
        })
        if 'spv_b_ids' in data and not self.materialize_conf and feat_c0.shape[-1] == 256 and 'mask0' not in data:
            # validation (lightning_loftr.py:266-267: _trainval_inference with the matcher in eval mode, then the loss):
            # the coarse loss reads conf_matrix at the ground-truth positions only (loftr_loss.py:86-91), so those are
            # evaluated instead of the dense matrix -- the forward half of the training kernels, no graph
            with torch.no_grad():
                data['conf_pos'] = ops.coarse_pos_conf(feat_c0.float().contiguous(), feat_c1.float().contiguous(),
                                                       data['spv_b_ids'], data['spv_i_ids'], data['spv_j_ids'], self.temperature)
        else:
            data.pop('conf_pos', None)

    # ------------------------------------------------------------------------------------------------------
    # training (coarse_matching.py:86-147 + :199-240).  The coarse loss of this configuration (dual-softmax, sparse
    # supervision, focal) reads conf_matrix only at the ground-truth positions (loftr_loss.py:86-91), so on the GPU
    # the dense matrix is NOT built: K1's fused kernels select the predicted matches (no grad, as in the reference
    # where get_coarse_match runs under no_grad) and ops.coarse_pos_conf gives the differentiable confidences at
    # spv_b/i/j_ids with a HIP backward (data['conf_pos']; data['conf_matrix'] is None).  far_amd.losses mirrors the
    # loss on them.  Padded masks take the dense autograd composition of far_amd/train_glue.py; `materialize_conf`, CPU tensors or a
    # feature width other than 256 need the test-side helper (far_amd/_vendor.py) -- explicitly, for drop-in use of the reference's
    # own dense loss.
    # ------------------------------------------------------------------------------------------------------
    def _forward_train(self, feat_c0, feat_c1, data, mask_c0=None, mask_c1=None):
        # padded-mask batches (coarse_matching.py:28-43, 110-117, 199-204: images of different sizes padded to one grid) take the
        # dense differentiable form: the sparse training kernels of K1 (far_coarse_pos_conf_*) carry no masks
        sparse = (feat_c0.is_cuda and feat_c0.shape[-1] == 256 and not self.materialize_conf and 'spv_b_ids' in data
                  and self.config.get('sparse_spvs', True) and 'mask0' not in data)
        if not sparse:
            if feat_c0.is_cuda and not self.materialize_conf and 'mask0' in data:
                # padded-mask batches on the GPU: the dense form as an autograd composition (far_amd/train_glue.py, accepted)
                conf = train_glue.masked_conf_matrix(feat_c0, feat_c1, self.temperature, mask_c0, mask_c1)
            else:       # CPU tensors, materialize_conf (the reference's own dense loss), other widths: test-side helper
                conf = ag.require('the dense differentiable conf_matrix (CPU tensors / materialize_conf / sparse_spvs off / C != 256)').conf_matrix(
                    feat_c0, feat_c1, self.temperature, mask_c0, mask_c1)
            data.update({'conf_matrix': conf})
            with torch.no_grad():
                b_ids, i_ids, j_ids, mconf = self._select_dense(conf.detach(), data)
                data.update(**self._sample_train(b_ids, i_ids, j_ids, mconf, data, conf.shape[0]))
            return
        with torch.no_grad():
            out = ops.coarse_match(feat_c0.detach().float().contiguous(), feat_c1.detach().float().contiguous(),
                                   self.temperature, self.thr, self.border_rm, data['hw0_c'], data['hw1_c'],
                                   data['hw0_i'][0] / data['hw0_c'][0], variant=self.variant)
            picked = self._sample_train(out['b_ids'], out['i_ids'], out['j_ids'], out['mconf'], data, feat_c0.shape[0])
        data.update({'conf_matrix': None,
                     'conf_pos': ops.coarse_pos_conf(feat_c0, feat_c1, data['spv_b_ids'], data['spv_i_ids'],
                                                     data['spv_j_ids'], self.temperature)})
        data.update(**picked)

    def _select_dense(self, conf, data):
        """coarse_matching.py:174-197 on a dense conf_matrix: threshold, border, mutual nearest neighbour."""
        N, L, S = conf.shape
        h0, w0 = data['hw0_c']
        h1, w1 = data['hw1_c']
        mask = (conf > self.thr).view(N, h0, w0, h1, w1).clone()
        b = self.border_rm
        if b > 0:
            mask[:, :b] = False; mask[:, :, :b] = False; mask[:, :, :, :b] = False; mask[:, :, :, :, :b] = False
            if 'mask0' not in data:                                                        # mask_border :8-25
                mask[:, -b:] = False; mask[:, :, -b:] = False; mask[:, :, :, -b:] = False; mask[:, :, :, :, -b:] = False
            else:                                                                          # mask_border_with_padding :28-43: the far
                m0, m1 = data['mask0'], data['mask1']                                      # border follows each image's valid extent
                h0s, w0s = m0.sum(1).max(-1)[0].int().tolist(), m0.sum(-1).max(-1)[0].int().tolist()
                h1s, w1s = m1.sum(1).max(-1)[0].int().tolist(), m1.sum(-1).max(-1)[0].int().tolist()
                for n, (a0, c0, a1, c1) in enumerate(zip(h0s, w0s, h1s, w1s)):
                    mask[n, a0 - b:] = False; mask[n, :, c0 - b:] = False; mask[n, :, :, a1 - b:] = False; mask[n, :, :, :, c1 - b:] = False
        mask = mask.view(N, L, S)
        mask = mask * (conf == conf.max(dim=2, keepdim=True)[0]) * (conf == conf.max(dim=1, keepdim=True)[0])
        mask_v, all_j = mask.max(dim=2)
        b_ids, i_ids = torch.where(mask_v)
        j_ids = all_j[b_ids, i_ids]
        return b_ids, i_ids, j_ids, conf[b_ids, i_ids, j_ids]

    @torch.no_grad()
    def _sample_train(self, b_ids, i_ids, j_ids, mconf, data, N):
        """coarse_matching.py:199-265: the training-time sampling / GT padding of the predicted matches (ordered by
        (b, i), as torch.where and K1 both emit them).  The two torch.randint draws are issued in the reference's order
        with the reference's arguments, so a seeded generator reproduces its choices."""
        h0, w0 = data['hw0_c']
        h1, w1 = data['hw1_c']
        L, S = h0 * w0, h1 * w1
        dev = b_ids.device
        if self.training:
            if 'mask0' not in data:
                n_cand = N * max(L, S)                                                     # :202-203
            else:                                                                          # compute_max_candidates :46-57
                m0, m1 = data['mask0'], data['mask1']
                a0 = m0.sum(1).max(-1)[0] * m0.sum(-1).max(-1)[0]
                a1 = m1.sum(1).max(-1)[0] * m1.sum(-1).max(-1)[0]
                n_cand = torch.sum(torch.min(torch.stack([a0, a1], -1), -1)[0])
            n_train = int(n_cand * self.train_coarse_percent)                              # :205-210
            n_pred = len(b_ids)
            assert self.train_pad_num_gt_min < n_train, "min-num-gt-pad should be less than num-train-matches"
            if n_pred <= n_train - self.train_pad_num_gt_min:                              # :216-222
                pred_idx = torch.arange(n_pred, device=dev)
            else:
                pred_idx = torch.randint(n_pred, (n_train - self.train_pad_num_gt_min,), device=dev)
            gt_idx = torch.randint(len(data['spv_b_ids']), (max(n_train - n_pred, self.train_pad_num_gt_min),),
                                   device=dev)                                             # :225-229
            zeros = torch.zeros(len(data['spv_b_ids']), device=dev)                        # :230
            b_ids = torch.cat([b_ids[pred_idx], data['spv_b_ids'][gt_idx]])
            i_ids = torch.cat([i_ids[pred_idx], data['spv_i_ids'][gt_idx]])
            j_ids = torch.cat([j_ids[pred_idx], data['spv_j_ids'][gt_idx]])
            mconf = torch.cat([mconf[pred_idx], zeros[gt_idx]])
        scale = data['hw0_i'][0] / data['hw0_c'][0]
        s0 = scale * data['scale0'][b_ids] if 'scale0' in data else scale
        s1 = scale * data['scale1'][b_ids] if 'scale1' in data else scale
        mk0 = torch.stack([i_ids % w0, torch.div(i_ids, w0, rounding_mode='floor')], dim=1) * s0
        mk1 = torch.stack([j_ids % w1, torch.div(j_ids, w1, rounding_mode='floor')], dim=1) * s1
        keep = mconf != 0
        return {'b_ids': b_ids, 'i_ids': i_ids, 'j_ids': j_ids, 'gt_mask': mconf == 0, 'm_bids': b_ids[keep],
                'mkpts0_c': mk0[keep], 'mkpts1_c': mk1[keep], 'mconf': mconf[keep]}


# =====================================================================================================
# fine level
# =====================================================================================================
class FinePreprocess(nn.Module):
    hip_training = True          # training on the GPU: K3 gather + its scatter backward; False: the test-side unfold composition

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.cat_c_feat = config['fine_concat_coarse_feat']
        self.W = config['fine_window_size']
        d_c, d_f = config['coarse']['d_model'], config['fine']['d_model']
        self.d_model_f = d_f
        if self.cat_c_feat:
            self.down_proj = nn.Linear(d_c, d_f, bias=True)
            self.merge_feat = nn.Linear(2 * d_f, d_f, bias=True)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")

    fused_gather = not flags.off('FAR_NO_GATHER_FUSE')   # inference: merge_feat reads the fine map through the window indices

    def _fused_gather_ok(self, f0, f1, feat_c0, data):
        """K9's gather mode applies: inference on the GPU with fine_concat_coarse_feat, and the two fine maps are the halves of one
        NHWC (channels_last) fp32 buffer of equal shape (what the fused backbone returns for equally sized images)."""
        if not (self.fused_gather and self.cat_c_feat and f0.is_cuda and f0.dtype == torch.float32 and f0.shape == f1.shape
                and data['hw0_c'][1] == data['hw1_c'][1] and not ag.needs_grad(feat_c0, self.merge_feat.weight, self.down_proj.weight)):
            return False
        n, C, H, Wd = f0.shape
        nhwc = (H * Wd * C, 1, Wd * C, C)
        return (tuple(f0.stride()) == nhwc and tuple(f1.stride()) == nhwc and C == self.d_model_f and
                f0.untyped_storage().data_ptr() == f1.untyped_storage().data_ptr() and f1.storage_offset() == f0.storage_offset() + f0.numel())

    def forward(self, feat_f0, feat_f1, feat_c0, feat_c1, data):
        W = self.W
        if not feat_f0.is_cuda:
            ag.require('FinePreprocess on CPU tensors')
        stride = data['hw0_f'][0] // data['hw0_c'][0]
        data.update({'W': W})
        b, i, j = data['b_ids'], data['i_ids'], data['j_ids']
        if b.shape[0] == 0:                                                          # :34-37
            e = torch.empty(0, W ** 2, self.d_model_f, device=feat_f0.device)
            return e, e.clone()
        # direct gather of the M x 25 x C window values instead of unfolding both full maps (:40-47)
        if ag.needs_grad(feat_f0, feat_f1):
            if feat_f0.is_cuda and self.hip_training:     # K3 gather forward + scatter backward (no unfold of the full maps)
                w0 = ops.fine_windows_train(feat_f0, b, i, data['hw0_c'][1], W, stride)
                w1 = ops.fine_windows_train(feat_f1, b, j, data['hw1_c'][1], W, stride)
            else:
                vend = ag.require('FinePreprocess under autograd on CPU tensors or with hip_training = False')
                w0 = vend.fine_windows(feat_f0, b, i, W, stride)
                w1 = vend.fine_windows(feat_f1, b, j, W, stride)
        elif self._fused_gather_ok(feat_f0, feat_f1, feat_c0, data):
            # the windows are never stored: merge_feat's Linear launch reads its rows through the window indices (K9 gather mode),
            # both images in one launch -- feat_f0 / feat_f1 are the halves of the backbone's one NHWC buffer
            M = b.shape[0]
            n = feat_f0.shape[0]
            fmap = torch.as_strided(feat_f0, (2 * n, feat_f0.shape[2], feat_f0.shape[3], feat_f0.shape[1]),
                                    (feat_f0.stride(0), feat_f0.stride(2), feat_f0.stride(3), 1))
            pk = self.__dict__.setdefault('_packs', ops.PackCache())
            dw, db = self.down_proj.weight, self.down_proj.bias
            b01, c01 = torch.cat([b, b + n]), torch.cat([i, j])
            pdown = pk.get('down', [dw, db], lambda: ops.PackedConv(dw, None, db))
            if (feat_c0.shape == feat_c1.shape and feat_c0.is_contiguous() and feat_c1.is_contiguous() and feat_c0.dtype == torch.float32
                    and feat_c0.untyped_storage().data_ptr() == feat_c1.untyped_storage().data_ptr()
                    and feat_c1.storage_offset() == feat_c0.storage_offset() + feat_c0.numel()):
                # down_proj(cat[feat_c0[b, i], feat_c1[b, j]]) (:50-51) the same way: a 1 x 1 "window" per match, read from the coarse
                # transformer's joint output buffer -- no index / cat kernels, no (2M, 256) tensor
                tok = torch.as_strided(feat_c0, (2 * n, 1, feat_c0.shape[1], feat_c0.shape[2]),
                                       (feat_c0.stride(0), feat_c0.stride(0), feat_c0.stride(1), 1))
                c_win = ops.linear_gather_f16s(tok, b01, c01, feat_c0.shape[1], 1, 1, pdown).view(2 * M, -1)
            else:
                c_win = ops.linear_f16s(torch.cat([feat_c0[b, i], feat_c1[b, j]], 0).contiguous(), pdown)
            d = self.d_model_f
            wt = self.merge_feat.weight
            pf = pk.get('merge_f', [wt], lambda: ops.PackedConv(wt[:, :d].contiguous()))
            pc = pk.get('merge_c', [wt, self.merge_feat.bias],
                        lambda: ops.PackedConv(wt[:, d:].contiguous(), None, self.merge_feat.bias))
            cw = ops.linear_f16s(c_win.contiguous(), pc)                              # (2M, d)
            both = ops.linear_gather_f16s(fmap, b01, c01, data['hw0_c'][1], W, stride, pf, residual=cw, res_group=W ** 2)
            return both[:M], both[M:]
        else:       # both images' windows into the halves of one buffer: the later cat([w0, w1]) is then free
            M = b.shape[0]
            w01 = torch.empty(2 * M, W ** 2, feat_f0.shape[1], dtype=torch.float32, device=feat_f0.device)
            w0 = ops.fine_gather(feat_f0.float(), b, i, data['hw0_c'][1], W, stride, out=w01[:M])
            w1 = ops.fine_gather(feat_f1.float(), b, j, data['hw1_c'][1], W, stride, out=w01[M:])
        if self.cat_c_feat:
            c_in = torch.cat([feat_c0[b, i], feat_c1[b, j]], 0)                      # [2M, C_coarse]
            if ag.needs_grad(w0, w1, c_in, self.merge_feat.weight, self.down_proj.weight) or not w0.is_cuda:
                c_win = self.down_proj(c_in)                                         # [2M, C]
                both = torch.cat([torch.cat([w0, w1], 0), c_win.unsqueeze(1).expand(-1, W ** 2, -1)], -1)
                w0, w1 = torch.chunk(self.merge_feat(both), 2, dim=0)
            else:
                # down_proj on K9 as well: every output row depends on its input row only, so a pair's windows do not
                # depend on how many other pairs' matches share the launch (a vendor GEMM picks its kernel by the row
                # count, which moved sub-pixel positions by an ulp between batch sizes)
                pk = self.__dict__.setdefault('_packs', ops.PackCache())
                dw, db = self.down_proj.weight, self.down_proj.bias
                c_win = ops.linear_f16s(c_in.contiguous(), pk.get('down', [dw, db], lambda: ops.PackedConv(dw, None, db)))
                # merge_feat(cat[window, repeat(c_win)]) = window W_f^T + (c_win W_c^T + b) repeated over the WW tokens
                # (:52-57): neither the repeat nor the (2M, 25, 256) concatenation is materialised -- K9 adds one
                # residual row per group of WW consecutive rows.
                d = self.d_model_f
                wt = self.merge_feat.weight
                pf = pk.get('merge_f', [wt], lambda: ops.PackedConv(wt[:, :d].contiguous()))
                pc = pk.get('merge_c', [wt, self.merge_feat.bias],
                            lambda: ops.PackedConv(wt[:, d:].contiguous(), None, self.merge_feat.bias))
                cw = ops.linear_f16s(c_win.contiguous(), pc)                          # (2M, d)
                both = ops.linear_f16s(w01, pf, residual=cw, res_group=W ** 2)
                w0, w1 = torch.chunk(both, 2, dim=0)
        return w0, w1


class FineMatching(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config

    def forward(self, feat_f0, feat_f1, data, train=False):
        M, WW, C = feat_f0.shape
        W = int(math.sqrt(WW))
        scale = data['hw0_i'][0] / data['hw0_f'][0]
        if M == 0:                                                                   # :33-41
            assert self.training is False, "M is always >0 when training (coarse_matching.py)"
            data.update({'expec_f': torch.empty(0, 3, device=feat_f0.device),
                         'mkpts0_f': data['mkpts0_c'], 'mkpts1_f': data['mkpts1_c']})
            return
        if ag.needs_grad(feat_f0, feat_f1):
            if not feat_f0.is_cuda:
                ag.require('FineMatching on CPU tensors')
            coords, std = train_glue.fine_expect(feat_f0, feat_f1)
            data.update({'expec_f': torch.cat([coords, std.unsqueeze(1)], -1)})
            if not self.config['regress_rt'] or not train or self.config['regress']['use_simple_moe']:
                with torch.no_grad():
                    sc1 = scale * data['scale1'][data['b_ids']] if 'scale0' in data else scale
                    n = len(data['mconf'])
                    data.update({'mkpts0_f': data['mkpts0_c'],
                                 'mkpts1_f': data['mkpts1_c'] + (coords * (W // 2) * sc1)[:n]})
            return
        s1 = data['scale1'].float().contiguous() if 'scale0' in data else None       # :70
        expec, mk1 = ops.fine_expect(feat_f0.float().contiguous(), feat_f1.float().contiguous(),
                                     data['mkpts1_c'].contiguous(), (W // 2) * scale, s1,
                                     data['b_ids'] if s1 is not None else None)
        data.update({'expec_f': expec})
        if not self.config['regress_rt'] or not train or self.config['regress']['use_simple_moe']:   # :59-62
            n = len(data['mconf'])
            data.update({'mkpts0_f': data['mkpts0_c'], 'mkpts1_f': mk1[:n]})

"""The drop-in module: same constructor dict, methods, data-dict protocol and parameter names as
mp3d_loftr/src/loftr/loftr.py (LoFTR :14-211), with the hot operators on libfar_hip.so.

  forward(data)                          -> forward_feature_extraction + forward_correspondence_prediction
  forward_rt_prediction(data)            -> EMM head, writes regressed_rt / expec_rt / priorRT
All results are side effects on the caller's `data` dict.
Batched use: loftr_rt may be (3, 4) [reference, B = 1] or (B, 3, 4); count tensors (1,) or (B,).

(PositionEncodingSine lives here too: 2-D sinusoidal position encoding added to the 1/8 feature map. Mirrors mp3d_loftr/src/loftr/utils/position_encoding.py:6-42 (same buffer name `pe`, non-persistent).)
"""
import math

import numpy as np
import os

import torch
import torch.nn as nn

from .. import flags, ops
from ..pose6d import compute_normalized_6d, pose_mean_6d, pose_std_6d, rotation_6d_to_matrix
from .backbone import build_backbone
from .stages import CoarseMatching, FineMatching, FinePreprocess
from .transformer import LocalFeatureTransformer, LocalFeatureTransformerRegressor


class PositionEncodingSine(nn.Module):
    def __init__(self, d_model, max_shape=(256, 256), temp_bug_fix=True):
        super().__init__()
        ys = torch.ones(max_shape).cumsum(0).float().unsqueeze(0)
        xs = torch.ones(max_shape).cumsum(1).float().unsqueeze(0)
        k = torch.arange(0, d_model // 2, 2).float()
        if temp_bug_fix:
            div = torch.exp(k * (-math.log(10000.0) / (d_model // 2)))
        else:  # the historical operator-precedence variant kept by the reference (:28-29)
            div = torch.exp(k * (-math.log(10000.0) / d_model // 2))
        div = div[:, None, None]
        pe = torch.zeros((d_model, *max_shape))
        pe[0::4] = torch.sin(xs * div)
        pe[1::4] = torch.cos(xs * div)
        pe[2::4] = torch.sin(ys * div)
        pe[3::4] = torch.cos(ys * div)
        self.register_buffer('pe', pe.unsqueeze(0), persistent=False)

    def forward(self, x):
        pe = self.pe[:, :, :x.size(2), :x.size(3)]
        if x.is_cuda and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
            # the fused backbone hands over NHWC memory: add an NHWC copy of the table (cached per shape), so that the
            # sum is one contiguous pass (the NCHW table against NHWC features was a strided 250 us kernel) and the
            # 'n c h w -> n (h w) c' that follows stays a free view
            key = (x.size(2), x.size(3), x.device)
            cache = self.__dict__.setdefault('_pe_nhwc', {})
            if key not in cache:
                cache[key] = pe.contiguous(memory_format=torch.channels_last)
            pe = cache[key]
        return x + pe


def _tokens(fmap, pos_enc):
    """(N, C, H, W) coarse map -> (N, H*W, C) tokens with the sinusoidal encoding added ('n c h w -> n (h w) c')."""
    return pos_enc(fmap).flatten(2).transpose(1, 2).contiguous()


_POSE_STATS = {}
_SIDE_STREAMS = {}


def _side_stream(device, which=0):
    """The side streams of `device` (per process and device): 0 = the head's feature stage, 1 = the FPN's fine branch."""
    key = (str(device), which)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device)
    return _SIDE_STREAMS[key]


def _pose_stats(device):
    """pose_mean_6d / pose_std_6d on `device` (uploaded once)."""
    key = str(device)
    if key not in _POSE_STATS:
        _POSE_STATS[key] = (pose_mean_6d.to(device), pose_std_6d.to(device))
    return _POSE_STATS[key]


class LoFTR(nn.Module):
    """Reference interface (loftr.py:14-211): forward / forward_feature_extraction /
    forward_correspondence_prediction / forward_rt_prediction / preprocess_helper / load_state_dict."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        save = config.get('save_preds')
        has_matcher = config['from_saved_preds'] is None or (save is not None and 'ground_truth' in save)
        if has_matcher:                                                       # loftr.py:20-27
            if config.get('predict_translation_scale'):
                raise NotImplementedError('predict_translation_scale is off in every FAR script (loftr.py:29-53)')
            cc = config['coarse']
            self.backbone = build_backbone(config)
            self.pos_encoding = PositionEncodingSine(cc['d_model'], temp_bug_fix=cc['temp_bug_fix'])
            self.loftr_coarse = LocalFeatureTransformer(cc)
            self.coarse_matching = CoarseMatching(config['match_coarse'])
            self.fine_preprocess = FinePreprocess(config)
            self.loftr_fine = LocalFeatureTransformer(config["fine"])
            self.fine_matching = FineMatching(config)
        if config['regress_rt']:
            self.loftr_regress = LocalFeatureTransformerRegressor(config)
        self.precision_stages = ()      # stages on 16-bit operands (set_precision); () = fp32-grade everywhere = the parity configuration
        # exponent of the power-of-two activation scale of the split-fp16 kernels (ops.activation_exponent): 4 covers
        # |activation| <= 4094; lowered by _widen_activation_range when a launch reports an overflow
        self.act_exp = 4

    # Operand precision of the matrix products, by stage.  Tensors stay fp32, accumulation stays fp32, the float64 solver is untouched.
    # The reference computes in fp32 throughout (no mixed-precision region, no half / bfloat16 cast anywhere in /root/reference): 'fp32' -- split-fp16 operand pairs,
    # fp32-grade products -- is the ONLY parity configuration.  Every 16-bit stage below changes results (ids are no longer bit-exact,
    # the solver's pose error on the bench pairs grows: bench.py `other_modes` / `precision_stages` report it per stage); they exist
    # because BASELINE configs[1] names a 16-bit operand class, and are never what `value` is measured on.
    STAGES = ('trunk',          # backbone trunk (stem excluded): K9 on plain fp16 operands instead of K17 / K9 on split pairs
              'fpn',            # the FPN's fine branch (layer1 / layer2 outconvs): coarse features and match decisions unchanged
              'coarse_dense',   # d_model-256 layers (coarse transformer + the head's two): merge / MLP launches on plain fp16
              'coarse_state',   # ... and their fused k|v-state / q-apply launches too (implies coarse_dense)
              'fine_layers',    # the fine level's d_model-128 layers: K13 / K14 on plain fp16 (far_mlp_fused_f16 / far_attn_block_f16)
              'k1',             # coarse matching on bf16 operands (far_coarse_match_bf16)
              'k2')             # the head's CrossAttention: qkv projection and K2 on plain fp16 (far_emm_pv_f16)
    MODES = {'fp32': (), 'fp16-fine': ('fpn',),
             'mixed16': ('trunk', 'fpn', 'coarse_dense', 'k1', 'k2'),
             'fp16': STAGES}
    PRECISIONS = tuple(MODES)
    head_prefetch = not flags.off('FAR_NO_PREFETCH')   # inference: the head's feature stage enqueued behind K1 (see below)
    head_side_stream = not flags.off('FAR_NO_SIDE_STREAM')   # ... on a second HIP stream, next to K1 and the fine level
    fpn_side_stream = flags.off('FAR_FPN_STREAM')            # opt-in: the FPN's fine branch on a side stream, next to the coarse transformer

    def set_precision(self, mode):
        """mode: a name of MODES or an iterable of STAGES names (the stages that run on 16-bit operands).
          'fp32'      every product on split-fp16 operand pairs: fp32-grade, the parity configuration (default);
          'fp16-fine' plain-fp16 operands in the FPN branch only: coarse features, ids and mconf bit-identical to 'fp32';
          'mixed16'   16-bit operands where they pay most (backbone, K1, K2, the d256 layers' merge / MLP launches); the attention-state
                      launches and the fine level keep fp32-grade products;
          'fp16'      16-bit operands in every large matrix product of the step: the fastest mode and the least accurate
                      (FinePreprocess's two projections, the small attention-state products inside the epilogues and the head's
                      small dense layers keep fp32-grade products: < 2 % of the step)."""
        if isinstance(mode, str):
            if mode not in self.MODES:
                raise ValueError(f'precision must be one of {self.PRECISIONS} or an iterable of {self.STAGES}')
            st = set(self.MODES[mode])
        else:
            st = set(mode)
            if not st <= set(self.STAGES):
                raise ValueError(f'unknown precision stages {sorted(st - set(self.STAGES))}; known: {self.STAGES}')
        self.precision_stages = tuple(s for s in self.STAGES if s in st)
        if hasattr(self, 'backbone'):                      # (the cached-prediction configuration builds the head only)
            self.backbone.trunk_split = 'trunk' not in st
            self.backbone.fpn_split = 'fpn' not in st
        from .transformer import CrossAttention, LoFTREncoderLayer
        fine = set(self.loftr_fine.modules()) if hasattr(self, 'loftr_fine') else set()
        for m in self.modules():
            if isinstance(m, LoFTREncoderLayer):
                if m in fine:
                    m.split_operands, m.dense_split = 'fine_layers' not in st, True
                else:
                    m.split_operands, m.dense_split = 'coarse_state' not in st, 'coarse_dense' not in st
            if isinstance(m, CrossAttention):
                m.plain16 = 'k2' in st
        if hasattr(self, 'coarse_matching'):
            self.coarse_matching.bf16 = 'k1' in st
        return self

    # -------------------------------------------------------------------------------------------------
    # stage 1: local feature CNN on both images at once (loftr.py:56-89)
    # -------------------------------------------------------------------------------------------------
    def _run_backbone(self, images, side_ok=False):
        from .backbone import _fused_ok
        if side_ok and self.fpn_side_stream and _fused_ok(self.backbone, images) and images.shape[1] == 1:
            # the FPN's fine branch on a second stream, next to the coarse transformer and K1 (backbone.py: _forward_fused);
            # _correspondence_prediction joins it in front of FinePreprocess, the guards in front of their flag read
            self._join_side()
            feats, self._fpn_pending = self.backbone(images, side=_side_stream(images.device, 1))
            return feats
        return self.backbone(images)

    def forward_feature_extraction(self, data):
        with ops.activation_exponent(self.act_exp):
            self._feature_extraction(data)

    def _feature_extraction(self, data, side_ok=False):
        """side_ok: the caller runs _correspondence_prediction next (forward): the fine maps may still be in flight on the side stream
        when this returns.  The public forward_feature_extraction hands out finished maps."""
        im0, im1 = data['image0'], data['image1']
        n = im0.size(0)
        data.update(bs=n, hw0_i=im0.shape[2:], hw1_i=im1.shape[2:])
        if im0.shape[2:] == im1.shape[2:]:
            feats_c, feats_f = self._run_backbone(torch.cat([im0, im1], dim=0), side_ok)
            c0, c1 = feats_c.split(n)
            f0, f1 = feats_f.split(n)
        else:                                                                  # different input shapes (:76-77)
            (c0, f0), (c1, f1) = self._run_backbone(im0), self._run_backbone(im1)
            feats_c = None
        data.update(hw0_c=c0.shape[2:], hw1_c=c1.shape[2:], hw0_f=f0.shape[2:], hw1_f=f1.shape[2:],
                    featmap0=c0, featmap1=c1, featmap_f0=f0, featmap_f1=f1, feats_c=feats_c)

    # -------------------------------------------------------------------------------------------------
    # stages 2-5: coarse transformer, K1 coarse match, K3 fine refinement (loftr.py:91-135)
    # -------------------------------------------------------------------------------------------------
    def forward_correspondence_prediction(self, data, train=False):
        with ops.activation_exponent(self.act_exp):
            self._correspondence_prediction(data, train)

    def _correspondence_prediction(self, data, train=False):
        data.pop(self._HEAD_KEY, None)                     # new coarse features: the head's cached stage is void
        tok0 = _tokens(data['featmap0'], self.pos_encoding)
        tok1 = _tokens(data['featmap1'], self.pos_encoding)
        m0 = m1 = None
        if 'mask0' in data:
            m0, m1 = data['mask0'].flatten(-2), data['mask1'].flatten(-2)
        tok0, tok1 = self.loftr_coarse(tok0, tok1, m0, m1)
        # The head's feature stage (2 LoFTR layers, K2, CrossBlock: ~13 ms per 32 pairs) reads the coarse tokens only -- not the
        # matches, not the solver's numbers.  Enqueued behind K1 it keeps the GPU busy while the host waits for the match count and
        # prepares the fine-level launches (0.6 ms of idle GPU per step otherwise); forward_rt_prediction finds it in the data dict
        # (_head_features: same tensors, same stamps).  It runs inside this call's activation-range guard like everything else here.
        # Only when the caller says the head follows (far_amd.pipeline.test_step sets data['_far_head_follows']): a matcher-only
        # caller (Map-free: match + solve at another resolution) must not pay for -- or trip over -- a stage it never runs.
        # On a SECOND stream (head_side_stream): the stage forks off where the coarse tokens are complete -- in front of K1 -- and
        # joins where its features are first read (_head_features) or where an activation-range guard reads its flag, whichever
        # comes first.  K1, the fine level and the first solver round are one branch of the step's dependency graph, this stage the
        # other (12.7 against 13.1 ms per 32 pairs alone); side by side the workgroups of one fill the ramps, tails and launch gaps of
        # the other -- ~40 launches of 0.1-0.5 ms each on either side.  No buffer is shared between the branches: workspaces are
        # per-call allocations of the (stream-aware) caching allocator, the library's scratch slots are handed out per launch
        # (conv_igemm_f16s.hip: g_amax_slots), the overflow flag is an atomic OR.  Results are bit-identical (tests/test_flags_gpu.py).
        overlap = None
        if (self.head_prefetch and data.get('_far_head_follows') and self.config['regress_rt'] and tok0.is_cuda and not train
                and not torch.is_grad_enabled() and not self.training and getattr(self.loftr_regress, 'cache_features', True)):
            if self.head_side_stream:
                side, main = _side_stream(tok0.device), torch.cuda.current_stream()
                fork = torch.cuda.Event()
                fork.record()                              # tok0 / tok1 are complete here; K1 is not enqueued yet

                def overlap():
                    self._join_side('_side_pending')       # a stage nobody consumed (should not happen; keeps one event pending at most)
                    side.wait_event(fork)
                    with torch.cuda.stream(side):
                        self._head_features(data, tok0, tok1, None, None, join=False, reader=main)
                        done = torch.cuda.Event()
                        done.record()
                    self._side_pending = done
            else:
                overlap = lambda: self._head_features(data, tok0, tok1, None, None)
        self.coarse_matching(tok0, tok1, data, mask_c0=m0, mask_c1=m1, overlap=overlap)
        self._join_side('_fpn_pending')                    # the fine maps: first read here
        win0, win1 = self.fine_preprocess(data['featmap_f0'], data['featmap_f1'], tok0, tok1, data)
        if win0.size(0) != 0:
            win0, win1 = self.loftr_fine(win0, win1)
        self.fine_matching(win0, win1, data, train=train)
        # the coarse maps are REPLACED by the transformer outputs (N, HW, C): the head consumes these (:129-135)
        data.update(featmap0=tok0, featmap1=tok1, mask_c0=m0, mask_c1=m1, translation_scale=None)

    # -------------------------------------------------------------------------------------------------
    # stage 6a: solver pose -> the 13 normalised numbers the head receives, and their inverse-pose twin
    # (loftr.py:137-171).  Accepts loftr_rt (3, 4) [reference] or (B, 3, 4); counts (1,) or (B,).
    # -------------------------------------------------------------------------------------------------
    def preprocess_helper(self, data):
        f0, f1 = data['featmap0'], data['featmap1']
        preds = inv_preds = None
        if self.config['regress']['use_simple_moe']:
            dev = f0.device
            rt = data['loftr_rt'].detach().to(dev)
            rt = rt.unsqueeze(0) if rt.dim() == 2 else rt
            B = rt.shape[0]
            extra = []
            if self.config['regress']['regress_use_num_corres']:
                extra.append('num_correspondences')
            if self.config['use_many_ransac_thr']:
                extra += ['num_correspondences_before_ransac', 'inliers_best_tight', 'inliers_best_ultra_tight']
            if rt.is_cuda:                                                     # K11: the lines below in one launch
                cnts = []
                for k in extra:
                    c = data[k].detach().to(dev).reshape(B)
                    cnts.append(c if c.dtype in (torch.int32, torch.int64) else c.to(torch.int64))
                preds, inv_preds = ops.pose_features(rt.to(torch.float64), cnts)
                return f0, f1, data.get('mask_c0'), data.get('mask_c1'), preds, inv_preds
            last_row = torch.tensor([[[0, 0, 0, 1.]]], device=dev, dtype=rt.dtype).expand(B, -1, -1)
            rt_inv = torch.linalg.inv(torch.cat([rt, last_row], dim=1))[:, :3, :4]
            preds = compute_normalized_6d(rt.float())
            inv_preds = compute_normalized_6d(rt_inv).float()
            if extra:                                                          # counts / 500 (:158, :164-166)
                cnt = torch.cat([data[k].detach().float().to(dev).reshape(B, 1) / 500 for k in extra], -1)
                preds, inv_preds = torch.cat([preds, cnt], -1), torch.cat([inv_preds, cnt], -1)
        return f0, f1, data.get('mask_c0'), data.get('mask_c1'), preds, inv_preds

    # -------------------------------------------------------------------------------------------------
    # stage 6: regression head + solver/regressor blend; exports the prior for the next solver round
    # (loftr.py:173-192)
    # -------------------------------------------------------------------------------------------------
    def forward_rt_prediction(self, data):
        if not self.config['regress_rt']:
            return

        # (the re-run flag lives in a list, not on the function: `run.again` would make the closure refer to itself -- a reference
        # cycle that kept the whole `data` dict, gigabytes of device tensors, alive until Python's cyclic collector came by)
        again = [False]

        def run():
            data.pop(self._HEAD_KEY, None) if again[0] else None       # a re-run must not reuse features with inf / NaN in them
            again[0] = True
            self._rt_prediction(data)
        self._guarded(run, data['featmap0'].device)

    def _rt_prediction(self, data):
        f0, f1, m0, m1, preds, inv_preds = self.preprocess_helper(data)
        features = self._head_features(data, f0, f1, preds, inv_preds)
        pose, mlp_feats, gate = self.loftr_regress(f0, f1, mask0=m0, mask1=m1, loftr_preds=preds,
                                                   inv_loftr_preds=inv_preds, F=None, features=features)
        data.update(regressed_rt=pose, expec_rt=pose[0])
        rc = self.config['regress']
        if rc['save_mlp_feats']:
            data['mlp_feats'] = mlp_feats
        if rc['save_gating_weights']:
            data['gating_reg_weights'] = gate
        if self.config['solver'] == 'prior_ransac':
            # The reference hands the prior on as a numpy array (loftr.py:186-192).  Formed on the device and copied to pinned host
            # memory WITHOUT a synchronisation of its own: the copy is complete when forward_rt_prediction returns (the activation-range
            # read of _guarded synchronises behind it); the next solver round takes the device tensor (far_amd.supervision.spvs_RT).
            p = pose.detach().float()
            if p.is_cuda:
                mean, std = _pose_stats(p.device)
                prior_dev = ops.prior_from_pose(p, mean, std)                       # (B, 3, 4): one launch (K11)
                host = torch.empty(prior_dev.shape, dtype=torch.float32, pin_memory=True)
                host.copy_(prior_dev, non_blocking=True)
                prior = host.numpy()
                out = prior[0] if len(prior) == 1 else prior
                data['priorRT'] = out
                data['_priorRT_device'] = (out, prior_dev)
            else:
                R = rotation_6d_to_matrix(p[:, 3:] * pose_std_6d[3:] + pose_mean_6d[3:]).numpy()
                t = (p[:, :3] * pose_std_6d[:3] + pose_mean_6d[:3]).numpy()
                prior = np.concatenate([R, t[:, :, None]], axis=-1)                # (B, 3, 4)
                data['priorRT'] = prior[0] if len(prior) == 1 else prior
                data.pop('_priorRT_device', None)

    # The evaluation loop calls forward_rt_prediction FINE_PRED_STEPS times on one batch with only the 13 solver numbers
    # changing (lightning_loftr.py:338-343); the head's feature stage does not read them, so its result is kept IN THE
    # CALLER'S DATA DICT (one batch's lifetime; dropped by forward_correspondence_prediction) -- never in the module.
    # Valid while the same tensor objects sit in data['featmap0/1'] unmodified (version counters; far_amd.ops bumps
    # them on every `out=` write, torch does on its own in-place ops) and the head's weights / precision are
    # unchanged.  Inference tensors carry no version counter: under torch.inference_mode() they are immutable outside
    # inference mode and the dict scoping is what bounds the reuse.
    _HEAD_KEY = '_far_head_features'

    @staticmethod
    def _tensor_stamp(t):
        return (id(t), t.data_ptr(), tuple(t.shape), ops.tensor_version(t))

    def _join_side(self, *which):
        """The current stream waits for what is in flight on the side streams: the head's feature stage ('_side_pending'), the FPN's
        fine branch ('_fpn_pending'); no names = both."""
        for name in which or ('_fpn_pending', '_side_pending'):
            ev = self.__dict__.pop(name, None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    def _head_features(self, data, f0, f1, preds, inv_preds, join=True, reader=None):
        head = self.loftr_regress
        if join:
            self._join_side()
        if torch.is_grad_enabled() or not getattr(head, 'cache_features', True):
            return None                                    # training / caching disabled: the head computes them itself
        stamp = (self._tensor_stamp(f0), self._tensor_stamp(f1), head.feature_stamp())
        hit = data.get(self._HEAD_KEY)
        if hit is not None and hit[0] == stamp and hit[2] is f0 and hit[3] is f1:
            return hit[1]
        features = head.compute_features(f0, f1, preds, inv_preds)
        if not join:
            # allocated on the side stream, read on the caller's: the allocator must not hand the blocks out again (to side-stream
            # work of the next batch) before the reader's launches are done with them
            for t in ((features.feats, features.enc0, features.moe0) if hasattr(features, 'enc0') else (features,)):
                t.record_stream(reader)
        data[self._HEAD_KEY] = (stamp, features, f0, f1)
        return features

    @classmethod
    def invalidate_head_cache(cls, data):
        """For callers that rewrite data['featmap0/1'] through raw device pointers outside torch and far_amd.ops."""
        data.pop(cls._HEAD_KEY, None)

    # -------------------------------------------------------------------------------------------------
    # Activation range of the split-fp16 kernels.  The reference's fp32 convolutions and Linear layers accept any finite
    # activation (resnet_fpn.py:101-119); K9 / K13 / K14 / K1 / K2 split their operands around a power-of-two scale and
    # overflow to inf beyond 65504 / 2^e.  Every such launch ORs a device flag when that happens (ops.overflow_flag);
    # forward / forward_rt_prediction read it ONCE at their end and, if set, widen the range and run again:
    #   e -= 4 for K9 (16x the range per step, down to e = -24: |a| <= 1e12),
    #   the fused fine-level layers (K13 / K14, fixed exponent) run as K9 + K5 launches,
    #   K1 / K2 (fixed exponent) run on their exact-f32 MFMA variants.
    # The setting sticks to the module (a checkpoint with large activations pays the re-run once, not per batch).
    # -------------------------------------------------------------------------------------------------
    def _widen_activation_range(self):
        from .transformer import CrossAttention, LoFTREncoderLayer
        if self.act_exp - 4 < ops.ACT_EXP_MIN:
            raise ops.ActivationOverflow('activations beyond the widest split-fp16 range (|a| > 1e12 or non-finite inputs)')
        self.act_exp -= 4
        for m in self.modules():
            if isinstance(m, LoFTREncoderLayer):
                m.fused_attn = m.fused_mlp = False
            if isinstance(m, CrossAttention):
                m.exact_f32 = True
        if hasattr(self, 'coarse_matching'):
            self.coarse_matching.variant = 'f32'
        import warnings
        warnings.warn(f'far_amd: an activation exceeded the split-fp16 range; re-running with activation exponent {self.act_exp} '
                      f'(|a| <= {65504.0 / 2.0 ** self.act_exp:.3g}), unfused fine-level layers and the exact-f32 K1 / K2 variants')

    def _range_state(self):
        """What _widen_activation_range changes: (act_exp, per-module fused / variant settings)."""
        from .transformer import CrossAttention, LoFTREncoderLayer
        mods = []
        for m in self.modules():
            if isinstance(m, LoFTREncoderLayer):
                mods.append((m, 'layer', m.fused_attn, m.fused_mlp))
            elif isinstance(m, CrossAttention):
                mods.append((m, 'cross', m.exact_f32, None))
        cm = self.coarse_matching.variant if hasattr(self, 'coarse_matching') else None
        return self.act_exp, mods, cm

    def _restore_range_state(self, state):
        self.act_exp, mods, cm = state
        for m, kind, a, b in mods:
            if kind == 'layer':
                m.fused_attn, m.fused_mlp = a, b
            else:
                m.exact_f32 = a
        if cm is not None:
            self.coarse_matching.variant = cm

    def _guarded(self, fn, device, inputs=()):
        """Runs fn() under this module's activation exponent; on an overflow report widens the range and runs it again.
        The per-device flag is cleared on entry: backward launches (K9 dgrad, K16, the layer node) and other modules OR the same
        flag and nobody reads it after them, so a stale report would otherwise widen this module for good on its next clean call.
        A widening persists only if it produced a clean run: non-finite inputs (which raise the flag at every exponent) and
        activations beyond the widest range leave the module exactly as it was and raise ActivationOverflow."""
        if device.type != 'cuda':
            return fn()
        if getattr(self, '_in_sequence', False):           # inside guarded_sequence: ITS guard reads the flag, once, behind the whole sequence
            with ops.activation_exponent(self.act_exp):
                return fn()
        ops.overflow_flag(device).zero_()
        saved = None
        while True:
            try:
                with ops.activation_exponent(self.act_exp):
                    out = fn()
            except ops.ActivationOverflow:
                raise
            except Exception:
                self._join_side()
                # fn() may be a whole step (guarded_sequence): the stages behind an overflowed launch then ran on inf / NaN and may
                # have raised on them (non-finite keypoints, degenerate match counts, host-side conversions) before the flag was
                # read.  With the flag set that exception is a symptom of the overflow: widen and re-run; otherwise it is the caller's
                if not ops.activation_overflowed(device):
                    raise
            else:
                self._join_side()                              # the flag read below must see the side stream's launches too
                if not ops.activation_overflowed(device):      # one host read per call
                    return out
            if saved is None:
                saved = self._range_state()
                bad = [t for t in inputs if torch.is_tensor(t) and t.is_floating_point() and not bool(torch.isfinite(t).all())]
                if bad:
                    raise ops.ActivationOverflow('non-finite values in the inputs (image / feature tensors): the outputs of this call '
                                                 'contain inf / NaN; the activation range was left unchanged')
            try:
                self._widen_activation_range()
            except ops.ActivationOverflow:
                self._restore_range_state(saved)             # nothing that did not yield finite outputs is kept
                raise

    def guarded_sequence(self, fn, device, inputs=()):
        """fn() = a SEQUENCE of calls on one batch -- forward, forward_rt_prediction, the solver rounds in between (what
        far_amd.pipeline.test_step replays) -- under ONE activation-range guard: the calls inside do not read the overflow flag
        themselves (each read is a host synchronisation: the GPU then idles while the host enqueues the next stage, three times per
        evaluation step), the flag is read once behind the whole sequence, and an overflow widens the range and runs the WHOLE sequence
        again (fn must be re-runnable: it receives no arguments and rewrites every result it wrote).  Same guarantees as a guarded
        single call: results are never silently out of range when this returns."""
        if device.type != 'cuda' or getattr(self, '_in_sequence', False):
            return fn()

        def run():
            self._in_sequence = True
            try:
                return fn()
            finally:
                self._in_sequence = False
        return self._guarded(run, device, inputs)

    def check_activation_range(self, data):
        """For callers that drive forward_feature_extraction / forward_correspondence_prediction themselves: raises
        ops.ActivationOverflow if a launch since the last check left the split-fp16 range (outputs contain inf / NaN)."""
        self._join_side()
        ops.check_activation_range(data['image0'].device, 'LoFTR')

    def forward(self, data, train=False):
        """Results are side effects on `data` (loftr.py:194-205, which returns None).  The dict is also RETURNED: a
        wrapper that copies dict arguments on the way in (DistributedDataParallel built with device_ids) would otherwise
        leave the caller with a dict the module never wrote to; far_amd.pipeline merges a returned copy back."""
        def run():
            with ops.activation_exponent(self.act_exp):
                self._feature_extraction(data, side_ok=True)
            self.forward_correspondence_prediction(data, train=train)
        self._guarded(run, data['image0'].device, (data['image0'], data['image1']))
        return data

    def load_state_dict(self, state_dict, *args, **kwargs):
        # Lightning checkpoints carry the 'matcher.' prefix (lightning_loftr.py:58-75, loftr.py:207-211)
        stripped = {(k[len('matcher.'):] if k.startswith('matcher.') else k): v for k, v in state_dict.items()}
        return super().load_state_dict(stripped, *args, **kwargs)

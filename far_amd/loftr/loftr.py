"""The drop-in module: same constructor dict, methods, data-dict protocol and parameter names as
mp3d_loftr/src/loftr/loftr.py (LoFTR :14-211), with the hot operators on libfar_hip.so.

  forward(data)                          -> forward_feature_extraction + forward_correspondence_prediction
  forward_rt_prediction(data)            -> EMM head, writes regressed_rt / expec_rt / priorRT
All results are side effects on the caller's `data` dict.
Batched use: loftr_rt may be (3, 4) [reference, B = 1] or (B, 3, 4); count tensors (1,) or (B,).
"""
import numpy as np
import torch
import torch.nn as nn

from ..pose6d import compute_normalized_6d, pose_mean_6d, pose_std_6d, rotation_6d_to_matrix
from .backbone import build_backbone
from .coarse_matching import CoarseMatching
from .fine_matching import FineMatching
from .fine_preprocess import FinePreprocess
from .position_encoding import PositionEncodingSine
from .transformer import LocalFeatureTransformer, LocalFeatureTransformerRegressor


class LoFTR(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        save = config.get('save_preds')
        if config['from_saved_preds'] is None or (save is not None and 'ground_truth' in save):
            self.backbone = build_backbone(config)
            self.pos_encoding = PositionEncodingSine(config['coarse']['d_model'],
                                                     temp_bug_fix=config['coarse']['temp_bug_fix'])
            self.loftr_coarse = LocalFeatureTransformer(config['coarse'])
            self.coarse_matching = CoarseMatching(config['match_coarse'])
            self.fine_preprocess = FinePreprocess(config)
            self.loftr_fine = LocalFeatureTransformer(config["fine"])
            self.fine_matching = FineMatching(config)
            if config.get('predict_translation_scale'):
                raise NotImplementedError('predict_translation_scale is off in every FAR script (loftr.py:29-53)')
        if config['regress_rt']:
            self.loftr_regress = LocalFeatureTransformerRegressor(config)
        # bf16 + channels_last for the convolutional backbone only (vendor path); everything after it is fp32
        self.backbone_dtype = torch.float32

    # ---- 1. local feature CNN (loftr.py:56-89) ----
    def forward_feature_extraction(self, data):
        data.update({'bs': data['image0'].size(0),
                     'hw0_i': data['image0'].shape[2:], 'hw1_i': data['image1'].shape[2:]})
        bs = data['bs']
        if data['hw0_i'] == data['hw1_i']:
            x = torch.cat([data['image0'], data['image1']], dim=0)
            if self.backbone_dtype != torch.float32:
                x = x.to(self.backbone_dtype)
            if self.backbone_dtype != torch.float32:
                x = x.contiguous(memory_format=torch.channels_last)   # half-precision convs prefer NHWC
            with torch.autocast('cuda', dtype=self.backbone_dtype, enabled=self.backbone_dtype != torch.float32):
                feats_c, feats_f = self.backbone(x)
            feats_c, feats_f = feats_c.float(), feats_f.float()
            (feat_c0, feat_c1), (feat_f0, feat_f1) = feats_c.split(bs), feats_f.split(bs)
        else:
            (feat_c0, feat_f0), (feat_c1, feat_f1) = self.backbone(data['image0']), self.backbone(data['image1'])
            feats_c = None
        data.update({'hw0_c': feat_c0.shape[2:], 'hw1_c': feat_c1.shape[2:],
                     'hw0_f': feat_f0.shape[2:], 'hw1_f': feat_f1.shape[2:]})
        data.update({'featmap0': feat_c0, 'featmap1': feat_c1, 'featmap_f0': feat_f0, 'featmap_f1': feat_f1,
                     'feats_c': feats_c})

    # ---- 2-5. coarse transformer, coarse match, fine refinement (loftr.py:91-135) ----
    def forward_correspondence_prediction(self, data, train=False):
        feat_c0, feat_c1 = data['featmap0'], data['featmap1']
        feat_f0, feat_f1 = data['featmap_f0'], data['featmap_f1']
        # 'n c h w -> n (h w) c' after adding the positional encoding
        feat_c0 = self.pos_encoding(feat_c0).flatten(2).transpose(1, 2).contiguous()
        feat_c1 = self.pos_encoding(feat_c1).flatten(2).transpose(1, 2).contiguous()
        mask_c0 = mask_c1 = None
        if 'mask0' in data:
            mask_c0, mask_c1 = data['mask0'].flatten(-2), data['mask1'].flatten(-2)
        feat_c0, feat_c1 = self.loftr_coarse(feat_c0, feat_c1, mask_c0, mask_c1)
        self.coarse_matching(feat_c0, feat_c1, data, mask_c0=mask_c0, mask_c1=mask_c1)
        feat_f0_unfold, feat_f1_unfold = self.fine_preprocess(feat_f0, feat_f1, feat_c0, feat_c1, data)
        if feat_f0_unfold.size(0) != 0:
            feat_f0_unfold, feat_f1_unfold = self.loftr_fine(feat_f0_unfold, feat_f1_unfold)
        self.fine_matching(feat_f0_unfold, feat_f1_unfold, data, train=train)
        data.update({'featmap0': feat_c0, 'featmap1': feat_c1, 'mask_c0': mask_c0, 'mask_c1': mask_c1,
                     'translation_scale': None})

    # ---- 6a. solver pose -> normalised 6D inputs of the head (loftr.py:137-171) ----
    def preprocess_helper(self, data):
        feat_c0, feat_c1 = data['featmap0'], data['featmap1']
        mask_c0, mask_c1 = data.get('mask_c0'), data.get('mask_c1')
        preds = inv_preds = None
        if self.config['regress']['use_simple_moe']:
            dev = feat_c0.device
            rt = data['loftr_rt'].detach().to(dev)
            if rt.dim() == 2:
                rt = rt.unsqueeze(0)                                   # (B, 3, 4); reference squeezes to (3, 4)
            B = rt.shape[0]
            preds = compute_normalized_6d(rt.float())
            bottom = torch.tensor([[[0, 0, 0, 1.]]], device=dev, dtype=rt.dtype).expand(B, -1, -1)
            inv = torch.linalg.inv(torch.cat([rt, bottom], dim=1))[:, :3, :4]
            inv_preds = compute_normalized_6d(inv).float()

            def col(key):
                return data[key].detach().float().to(dev).reshape(B, 1) / 500
            if self.config['regress']['regress_use_num_corres']:
                n = col('num_correspondences')
                preds, inv_preds = torch.cat([preds, n], -1), torch.cat([inv_preds, n], -1)
            if self.config['use_many_ransac_thr']:
                n3 = torch.cat([col('num_correspondences_before_ransac'), col('inliers_best_tight'),
                                col('inliers_best_ultra_tight')], -1)
                preds, inv_preds = torch.cat([preds, n3], -1), torch.cat([inv_preds, n3], -1)
        return feat_c0, feat_c1, mask_c0, mask_c1, preds, inv_preds

    # ---- 6. regression head (loftr.py:173-192) ----
    def forward_rt_prediction(self, data):
        if not self.config['regress_rt']:
            return
        feat_c0, feat_c1, mask_c0, mask_c1, preds, inv_preds = self.preprocess_helper(data)
        pred_RT, mlp_features, gate = self.loftr_regress(feat_c0, feat_c1, mask0=mask_c0, mask1=mask_c1,
                                                         loftr_preds=preds, inv_loftr_preds=inv_preds, F=None)
        data.update({'regressed_rt': pred_RT, 'expec_rt': pred_RT[0]})
        if self.config['regress']['save_mlp_feats']:
            data.update({'mlp_feats': mlp_features})
        if self.config['regress']['save_gating_weights']:
            data.update({'gating_reg_weights': gate})
        if self.config['solver'] == 'prior_ransac':
            p = pred_RT.detach().float().cpu()
            R = rotation_6d_to_matrix(p[:, 3:] * pose_std_6d[3:] + pose_mean_6d[3:]).numpy()
            t = (p[:, :3] * pose_std_6d[:3] + pose_mean_6d[:3]).numpy()
            prior = np.concatenate([R, t[:, :, None]], axis=-1)      # (B, 3, 4)
            data.update({'priorRT': prior[0] if prior.shape[0] == 1 else prior})

    def forward(self, data, train=False):
        self.forward_feature_extraction(data)
        self.forward_correspondence_prediction(data, train=train)

    def load_state_dict(self, state_dict, *args, **kwargs):
        for k in list(state_dict.keys()):
            if k.startswith('matcher.'):
                state_dict[k.replace('matcher.', '', 1)] = state_dict.pop(k)
        return super().load_state_dict(state_dict, *args, **kwargs)

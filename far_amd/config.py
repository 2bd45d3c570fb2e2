"""Default model configuration: the dict the reference hands to LoFTR(config=...), i.e.
lower_config(cfg)['loftr'] (mp3d_loftr/src/lightning/lightning_loftr.py:40-50) for the FAR evaluation
setting (mp3d_loftr/scripts/eval_matterport.sh:27-37, demo.py:58-99).  Plain data, same keys/values."""
import copy

FAR_EVAL_CONFIG = {
    'backbone_type': 'ResNetFPN',
    'resolution': (8, 2),
    'fine_window_size': 5,
    'fine_concat_coarse_feat': True,
    'resnetfpn': {'initial_dim': 128, 'block_dims': [128, 196, 256]},
    'coarse': {'d_model': 256, 'd_ffn': 256, 'nhead': 8, 'layer_names': ['self', 'cross'] * 3,
               'attention': 'linear', 'temp_bug_fix': True},
    'match_coarse': {'thr': 0.2, 'border_rm': 2, 'match_type': 'dual_softmax', 'dsmax_temperature': 0.1,
                     'skh_iters': 3, 'skh_init_bin_score': 1.0, 'skh_prefilter': False,
                     'train_coarse_percent': 0.2, 'train_pad_num_gt_min': 200, 'sparse_spvs': True},
    'fine': {'d_model': 128, 'd_ffn': 128, 'nhead': 8, 'layer_names': ['self', 'cross'], 'attention': 'linear'},
    'regress': {'d_model': 256, 'd_ffn': 256, 'nhead': 8, 'layer_names': ['self', 'cross'], 'attention': 'linear',
                'temp_bug_fix': False, 'use_pos_embedding': True, 'regress_use_num_corres': True,
                'save_mlp_feats': False, 'use_simple_moe': True, 'use_2wt': True, 'use_5050_weight': False,
                'use_1wt': False, 'scale_8pt': True, 'save_gating_weights': False},
    'predict_translation_scale': False,
    'regress_rt': True,
    'regress_loftr_layers': 1,
    'from_saved_preds': None,
    'solver': 'prior_ransac',
    'use_many_ransac_thr': True,
    'fine_pred_steps': 2,
    'training': False,
}


def far_eval_config():
    return copy.deepcopy(FAR_EVAL_CONFIG)


def far_train_config():
    """The lower-cased config LoFTRLoss receives (lightning_loftr.py:40-53) for the last stage of
    mp3d_loftr/scripts/train_matterport.sh ("training FAR (full)": --rt_weight_rot/tr 0.01, --fine_weight/--coarse_weight 1,
    --use_l1_rt_loss, --solver prior_ransac, --fine_pred_steps 2) with the loss defaults of src/config/default.py:58-80."""
    loftr = far_eval_config()
    loftr['training'] = True
    loftr['loss'] = {'coarse_type': 'focal', 'coarse_weight': 1.0, 'focal_alpha': 0.25, 'focal_gamma': 2.0, 'pos_weight': 1.0,
                     'neg_weight': 1.0, 'fine_type': 'l2_with_std', 'fine_weight': 1.0, 'fine_correct_thr': 1.0,
                     'rt_weight_rot': 0.01, 'rt_weight_tr': 0.01, 'scale_weight': 0.0, 'use_l1_rt_loss': True,
                     'max_scale_loss': 100.0}
    return {'loftr': loftr, 'use_correspondence_transformer': False}


class TrainerCfg:
    """The two TRAINER fields spvs_RT reads (src/loftr/utils/supervision.py:192-193)."""
    RANSAC_PIXEL_THR = 0.5
    RANSAC_CONF = 0.99999


class RunCfg:
    """Minimal stand-in for the yacs node handed to compute_supervision_RT: config.TRAINER.*, config.LOFTR.SOLVER."""

    def __init__(self, solver='prior_ransac', fine_pred_steps=2, resolution=(8, 2), fine_window_size=5, minimal_solver=8):
        self.TRAINER = TrainerCfg()
        # MINIMAL_SOLVER (far_amd's own key, absent from the reference's config = 8): 8 = normalized 8-point hypotheses, the
        # solver north_star names; 5 = Nister's five-point solver for every pair (what the reference EXECUTES is a five-point
        # solver too -- OpenCV's; ransac.py:146-152).  Pairs with 5..7 correspondences use the five-point solver either way.
        self.LOFTR = type('L', (), {'SOLVER': solver, 'FINE_PRED_STEPS': fine_pred_steps, 'RESOLUTION': resolution,
                                    'FINE_WINDOW_SIZE': fine_window_size, 'MINIMAL_SOLVER': minimal_solver})()

    def __getitem__(self, key):                 # spvs_coarse / spvs_fine read config['LOFTR']['RESOLUTION'] (supervision.py:58, :151)
        node = getattr(self, key)
        return node if key != 'LOFTR' else {'RESOLUTION': node.RESOLUTION, 'FINE_WINDOW_SIZE': node.FINE_WINDOW_SIZE}

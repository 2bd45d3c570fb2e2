"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (crockwell/far @ /root/reference) on CPU.

Container-only: needs /root/reference and tools/ref_shim.py.  The fixtures are data (seeds, small inputs,
expected outputs); no reference source is copied.  Weights come from far_amd.synth (seeded), inputs from
seeds recorded in each file.  Run:  python tools/make_goldens.py
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True          # never leave __pycache__ files inside /root/reference
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import ref_shim  # noqa: E402

ref_shim.install()
OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)

from far_amd import synth  # noqa: E402
from tests.util import (GRAD_KEYS, correlated_features, loss_inputs, spvs_scene, train_inputs,  # noqa: E402
                        train_step, two_view_scene)

NOTE_KORNIA = ('uses tools/ref_shim.py restatements of kornia 0.7.1 (create_meshgrid / spatial_expectation2d / '
               'sampson_epipolar_distance): parity unpinned against kornia itself')


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def ref_model():
    from src.loftr import LoFTR
    cfg = ref_shim.far_eval_config()
    m = LoFTR(config=cfg).eval()
    synth.load_synthetic(m, seed=0)
    return m, cfg


def g1_coarse():
    """CoarseMatching.forward + get_coarse_match (coarse_matching.py:86-265) on correlated features."""
    spec = importlib.util.spec_from_file_location('ref_cm', ref_shim.REF_ROOT + '/src/loftr/utils/coarse_matching.py')
    cm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cm)
    cfg = ref_shim.far_eval_config()['match_coarse']
    mod = cm.CoarseMatching(cfg).eval()
    # small: full tensors
    f0, f1, _ = correlated_features(2, (12, 16), 64, seed=1, amp=2.0)
    data = {'hw0_i': (96, 128), 'hw1_i': (96, 128), 'hw0_c': (12, 16), 'hw1_c': (12, 16)}
    with torch.no_grad():
        mod(torch.from_numpy(f0), torch.from_numpy(f1), data)
    save('g1_coarse_small', seed=1, hw=(12, 16), C=64, amp=2.0, conf_matrix=data['conf_matrix'].numpy(),
         b_ids=data['b_ids'].numpy(), i_ids=data['i_ids'].numpy(), j_ids=data['j_ids'].numpy(),
         mconf=data['mconf'].numpy(), mkpts0_c=data['mkpts0_c'].numpy(), mkpts1_c=data['mkpts1_c'].numpy())
    # full grid: ids + conf of matches + row maxima only
    f0, f1, _ = correlated_features(1, (60, 80), 256, seed=3, amp=1.2, frac=0.8)
    data = {'hw0_i': (480, 640), 'hw1_i': (480, 640), 'hw0_c': (60, 80), 'hw1_c': (60, 80)}
    with torch.no_grad():
        mod(torch.from_numpy(f0), torch.from_numpy(f1), data)
    c = data['conf_matrix'][0]
    save('g1_coarse_full', seed=3, hw=(60, 80), C=256, amp=1.2, frac=0.8,
         b_ids=data['b_ids'].numpy(), i_ids=data['i_ids'].numpy(), j_ids=data['j_ids'].numpy(),
         mconf=data['mconf'].numpy(), mkpts0_c=data['mkpts0_c'].numpy(), mkpts1_c=data['mkpts1_c'].numpy(),
         rowmax=c.max(1)[0].numpy(), colmax=c.max(0)[0].numpy(), conf_sum=np.float64(c.double().sum().item()))


def g2_fine(m):
    """FinePreprocess (fine_preprocess.py:29-59) and FineMatching (fine_matching.py:15-76)."""
    rng = np.random.default_rng(12)
    N, C, Hf, Wf = 1, 128, 24, 32
    ff0 = rng.standard_normal((N, C, Hf, Wf)).astype(np.float32)
    ff1 = rng.standard_normal((N, C, Hf, Wf)).astype(np.float32)
    fc0 = rng.standard_normal((N, 48, 256)).astype(np.float32)
    fc1 = rng.standard_normal((N, 48, 256)).astype(np.float32)
    M = 24
    i_ids = rng.choice(48, M, replace=False).astype(np.int64)
    i_ids[:3] = [0, 7, 47]
    j_ids = rng.choice(48, M, replace=False).astype(np.int64)
    b_ids = np.zeros(M, np.int64)
    data = {'hw0_f': (Hf, Wf), 'hw0_c': (6, 8), 'b_ids': torch.from_numpy(b_ids), 'i_ids': torch.from_numpy(i_ids),
            'j_ids': torch.from_numpy(j_ids)}
    with torch.no_grad():
        w0, w1 = m.fine_preprocess(torch.from_numpy(ff0), torch.from_numpy(ff1), torch.from_numpy(fc0),
                                   torch.from_numpy(fc1), data)
    # fine matching on its own random windows
    f0 = rng.standard_normal((M, 25, 128)).astype(np.float32)
    f1 = rng.standard_normal((M, 25, 128)).astype(np.float32)
    f1[:10] = f0[:10, 12:13, :] * (rng.random((10, 25, 1)) > 0.8)
    mk = (rng.integers(0, 80, (M, 2)) * 8).astype(np.float32)
    d2 = {'hw0_i': (480, 640), 'hw0_f': (240, 320), 'mkpts0_c': torch.from_numpy(mk), 'mkpts1_c': torch.from_numpy(mk),
          'mconf': torch.ones(M), 'b_ids': torch.from_numpy(b_ids)}
    with torch.no_grad():
        m.fine_matching(torch.from_numpy(f0), torch.from_numpy(f1), d2)
    save('g2_fine', seed=12, note=NOTE_KORNIA, ff0=ff0, ff1=ff1, fc0=fc0, fc1=fc1, i_ids=i_ids, j_ids=j_ids,
         win0=w0.numpy(), win1=w1.numpy(), f0=f0, f1=f1, mk=mk, expec_f=d2['expec_f'].numpy(),
         mkpts1_f=d2['mkpts1_f'].numpy())


def g3_encoder(m):
    """LinearAttention (linear_attention.py:20-52) and LoFTREncoderLayer (transformer.py:44-67)."""
    from src.loftr.loftr_module.linear_attention import LinearAttention
    rng = np.random.default_rng(13)
    q = rng.standard_normal((2, 50, 8, 32)).astype(np.float32)
    k = rng.standard_normal((2, 70, 8, 32)).astype(np.float32)
    v = rng.standard_normal((2, 70, 8, 32)).astype(np.float32)
    with torch.no_grad():
        o = LinearAttention()(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v)).numpy()
        x = rng.standard_normal((2, 50, 256)).astype(np.float32)
        s = rng.standard_normal((2, 70, 256)).astype(np.float32)
        y = m.loftr_coarse.layers[1](torch.from_numpy(x), torch.from_numpy(s)).numpy()
        a, b = m.loftr_coarse(torch.from_numpy(x), torch.from_numpy(s))
    save('g3_encoder', seed=13, q=q, k=k, v=v, attn_out=o, x=x, src=s, layer1_out=y,
         stack_out0=a.numpy(), stack_out1=b.numpy())


def g4_head(m):
    """get_positional_encodings :183-248, CrossAttention :266-303, CrossBlock :335-348, forward_emm :423-483,
    LoFTR.preprocess_helper / forward_rt_prediction (loftr.py:137-192) at the only supported grid (60x80)."""
    from src.loftr.loftr_module.transformer import get_positional_encodings
    pos = get_positional_encodings(1, 4800, intrinsics=True).numpy()[0]
    rng = np.random.default_rng(14)
    f0 = rng.standard_normal((1, 4800, 256)).astype(np.float32)
    f1 = (0.5 * f0 + rng.standard_normal((1, 4800, 256))).astype(np.float32)
    emm = m.loftr_regress.emm
    with torch.no_grad():
        x1 = emm.norm1(torch.from_numpy(f0) + emm.pos_embed)
        x2 = emm.norm1(torch.from_numpy(f1) + emm.pos_embed)
        fa, fb = emm.cross_attn(x1, x2)
        blk = emm(torch.cat([torch.from_numpy(f0), torch.from_numpy(f1)], 0))
    # fake solver output
    ang = 0.3
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.6, -0.1, 0.79])
    rt = np.concatenate([R, t[:, None]], 1)
    data = {'featmap0': torch.from_numpy(f0), 'featmap1': torch.from_numpy(f1), 'loftr_rt': torch.from_numpy(rt),
            'num_correspondences': torch.tensor([731]), 'num_correspondences_before_ransac': torch.tensor([1500]),
            'inliers_best_tight': torch.tensor([410]), 'inliers_best_ultra_tight': torch.tensor([57])}
    with torch.no_grad():
        _, _, _, _, lp, ilp = m.preprocess_helper(data)
        m.forward_rt_prediction(data)
    save('g4_head', seed=14, pos6=pos, xattn_a=fa.numpy(), xattn_b=fb.numpy(), block_out=blk.numpy(), loftr_rt=rt,
         counts=np.array([731, 1500, 410, 57]), loftr_preds_6d=lp.numpy(), inv_loftr_preds_6d=ilp.numpy(),
         regressed_rt=data['regressed_rt'].numpy(), priorRT=data['priorRT'])


def g5_solver():
    """run_8point (cv_geometry.py:772-833), decompose_essential_matrix (essential.py:99-139),
    RANSAC.verify (ransac.py:256-292) and get_prior_estimate (:203-231) on seeded two-view data (float32)."""
    from cv_geometry import run_8point
    from essential import decompose_essential_matrix
    from ransac import RANSAC
    k0, k1, K, Rgt, tgt = two_view_scene(600, seed=41, outlier_frac=0.3)
    kn0 = ((k0.astype(np.float64) - K[[0, 1], [2, 2]]) / K[[0, 1], [0, 1]]).astype(np.float32)
    kn1 = ((k1.astype(np.float64) - K[[0, 1], [2, 2]]) / K[[0, 1], [0, 1]]).astype(np.float32)
    rng = np.random.default_rng(5)
    Hn = 256
    samples = np.stack([rng.choice(len(kn0), 8, replace=False) for _ in range(Hn)]).astype(np.int64)
    p1, p2 = torch.from_numpy(kn0)[samples], torch.from_numpy(kn1)[samples]
    with torch.no_grad():
        F = run_8point(p1, p2, torch.ones(Hn, 8))
        R1, R2, T = decompose_essential_matrix(F)
        pcl = np.random.RandomState(0).uniform(low=-3.0, high=3.0, size=(300, 3)).astype(np.float32)
        prior = np.concatenate([Rgt, (2.0 * tgt)[:, None]], 1)
        rs = RANSAC(model_type='essential_cv2', max_iter=1, inl_th=3e-7, batch_size=Hn, max_lo_iters=0,
                    prior_params={'rotation_pcl_error': True, 'rotation_error': False, 'K1': torch.eye(3),
                                  'K2': torch.eye(3), 'RT': torch.FloatTensor(prior), 'pcl': torch.FloatTensor(pcl),
                                  'lambda': 0.3, 'biased_sampling': 'biased'},
                    use_noexp_prior_scoring=True, use_linear_bias_sampling=True, bias_sigma_sq=0.1)
        perr = rs.get_prior_estimate(F)
        errs = rs.error_fn(torch.from_numpy(kn0)[None].expand(Hn, -1, 2), torch.from_numpy(kn1)[None].expand(Hn, -1, 2),
                           F, squared=True)
    save('g5_solver', seed=41, note=NOTE_KORNIA, kpts0=k0, kpts1=k1, K=K, R_gt=Rgt, t_gt=tgt, samples=samples,
         F=F.numpy(), R1=R1.numpy(), R2=R2.numpy(), T=T.numpy(), prior=prior, pcl=pcl, prior_err=perr.numpy(),
         sampson=errs.numpy().astype(np.float32), count=(errs <= 3e-7).sum(1).numpy())


def g6_pose6d():
    from src.losses.loftr_loss import compute_normalized_6d, rotation_6d_to_matrix, pose_mean_6d, pose_std_6d
    rng = np.random.default_rng(16)
    d6 = rng.standard_normal((20, 6)).astype(np.float32)
    R = rotation_6d_to_matrix(torch.from_numpy(d6)).numpy()
    rt = np.concatenate([R, rng.standard_normal((20, 3, 1)).astype(np.float32)], 2)
    n6 = compute_normalized_6d(torch.from_numpy(rt)).numpy()
    save('g6_pose6d', seed=16, d6=d6, R=R, rt=rt, n6=n6, mean=pose_mean_6d.numpy(), std=pose_std_6d.numpy())


def g7_full(m):
    """LoFTR.forward (loftr.py:194-205) at 640x480 on one synthetic pair with the synthetic checkpoint."""
    im0, im1 = synth.synth_image_pair(1, seed=0)
    data = {'image0': torch.from_numpy(im0), 'image1': torch.from_numpy(im1)}
    with torch.no_grad():
        m(data)
    c = data['conf_matrix'][0]
    rs = torch.sort(c, dim=1)[0]
    save('g7_full', seed=0, b_ids=data['b_ids'].numpy(), i_ids=data['i_ids'].numpy(), j_ids=data['j_ids'].numpy(),
         mconf=data['mconf'].numpy(), mkpts0_f=data['mkpts0_f'].numpy(), mkpts1_f=data['mkpts1_f'].numpy(),
         expec_f=data['expec_f'].numpy(), rowmax=rs[:, -1].numpy(), rowgap=(rs[:, -1] - rs[:, -2]).numpy(),
         featmap0_sample=data['featmap0'][0, ::97].numpy(), featmap1_sample=data['featmap1'][0, ::97].numpy(),
         feats_c_sample=data['feats_c'][:, ::16, ::7, ::9].numpy(),
         featmap_f0_sample=data['featmap_f0'][:, ::16, ::31, ::37].numpy())


def g9_metrics():
    """relative_pose_error (metrics.py:17-36) and error_auc (:307-324)."""
    from src.utils.metrics import relative_pose_error, error_auc
    rng = np.random.default_rng(19)
    B = 40
    Ts, Rs, ts, out = [], [], [], []
    for b in range(B):
        _, _, _, Rg, tg = two_view_scene(20, seed=100 + b)
        _, _, _, Re, te = two_view_scene(20, seed=100 + b + (b % 3))
        T = np.eye(4); T[:3, :3] = Rg; T[:3, 3] = tg * rng.uniform(0.5, 3)
        te = te * rng.uniform(0.5, 3) * (1 if b % 4 else -1)
        Ts.append(T); Rs.append(Re); ts.append(te)
        out.append(relative_pose_error(T, Re, te, ignore_gt_t_thr=0.0))
    out = np.array(out, np.float64)
    auc = error_auc(np.maximum(out[:, 0], out[:, 1]), [5, 10, 20])
    save('g9_metrics', T=np.stack(Ts), R=np.stack(Rs), t=np.stack(ts), errs=out,
         auc=np.array([auc['auc@5'], auc['auc@10'], auc['auc@20']]))


def g10_training(m):
    """Training-mode forward/backward through the reference: CoarseMatching train sampling (coarse_matching.py:199-240),
    differentiable conf_matrix / expec_f / regressed_rt, parameter gradients."""
    im0, im1, ii, jj, rt = train_inputs()
    data, losses = train_step(m, im0, im1, ii, jj, rt)
    P = dict(m.named_parameters())
    gn = np.array([P[k].grad.norm().item() for k in GRAD_KEYS])
    gs = np.stack([P[k].grad.reshape(-1)[:: max(1, P[k].numel() // 8)][:8].numpy() for k in GRAD_KEYS])
    save('g10_training', losses=np.array([l.item() for l in losses]), b_ids=data['b_ids'].numpy(), i_ids=data['i_ids'].numpy(),
         j_ids=data['j_ids'].numpy(), n_mconf=len(data['mconf']), expec_f_head=data['expec_f'][:64].detach().numpy(),
         regressed_rt=data['regressed_rt'].detach().numpy(), grad_norms=gn, grad_samples=gs)
    m.eval()


def g11_matcher_544x720(m):
    """LoFTR.forward (loftr.py:194-205) at 544x720 (coarse grid 68x90, L = S = 6120; fine grid 272x360): BASELINE
    configs[4]'s resolution on the mp3d_loftr matcher (the head is tied to 60x80 and is not part of this config)."""
    im0, im1 = synth.synth_image_pair(1, seed=5, hw=(544, 720))
    data = {'image0': torch.from_numpy(im0), 'image1': torch.from_numpy(im1)}
    with torch.no_grad():
        m(data)
    c = data['conf_matrix'][0]
    rs = torch.sort(c, dim=1)[0]
    save('g11_matcher_544x720', seed=5, hw=(544, 720), b_ids=data['b_ids'].numpy(), i_ids=data['i_ids'].numpy(),
         j_ids=data['j_ids'].numpy(), mconf=data['mconf'].numpy(), mkpts0_f=data['mkpts0_f'].numpy(),
         mkpts1_f=data['mkpts1_f'].numpy(), expec_f=data['expec_f'].numpy(), rowmax=rs[:, -1].numpy(),
         rowgap=(rs[:, -1] - rs[:, -2]).numpy(), featmap0_sample=data['featmap0'][0, ::97].numpy(),
         feats_c_sample=data['feats_c'][:, ::16, ::7, ::9].numpy(),
         featmap_f0_sample=data['featmap_f0'][:, ::16, ::31, ::37].numpy())


def g12_ransac_loop():
    """The reference's WHOLE hypothesise-and-verify loop, RANSAC.forward (ransac.py:340-442), run with committed
    sample indices: the object estimate_pose builds for 'prior_ransac' / 'prior_ransac_noprior' (metrics.py:104-147:
    model_type='essential_cv2', max_iter=1, inl_th=3e-7, lambda 0.3, linear bias sampling, sigma^2 0.1), with the
    minimal solver wired the way RANSAC(model_type='fundamental') wires it (ransac.py:140-145: run_8point, 8-point
    samples) -- the one north_star-sanctioned substitution (cv2's 5-point is absent).  `sample` (:161-175) is
    replaced by an observer that records the weight vector it is handed and returns the committed indices; `verify`
    and `remove_bad_models` are observed, not changed.  Pins: bias weights :358-371, remove_bad_models :303-308,
    prior error/score :203-231 + :395-398, Sampson counts + argmax :273-281, the masks at thr, thr/10, thr/100
    :284-287, the score floor :353/:409."""
    from cv_geometry import run_8point
    from ransac import RANSAC
    out = {}
    for tag, seed, with_prior in (('p', 77, True), ('n', 78, False)):
        k0, k1, K, Rgt, tgt = two_view_scene(500, seed=seed, outlier_frac=0.35)
        # metrics.py:88-89 on float32 pixel tensors and float64 intrinsics, then torch.FloatTensor (:124-125)
        kn0 = ((torch.from_numpy(k0) - torch.from_numpy(K)[[0, 1], [2, 2]][None]) / torch.from_numpy(K)[[0, 1], [0, 1]][None]).numpy()
        kn1 = ((torch.from_numpy(k1) - torch.from_numpy(K)[[0, 1], [2, 2]][None]) / torch.from_numpy(K)[[0, 1], [0, 1]][None]).numpy()
        kp1, kp2 = torch.FloatTensor(kn0), torch.FloatTensor(kn1)
        rng = np.random.default_rng(seed)
        Hn = 512
        samples = np.stack([rng.choice(len(k0), 8, replace=False) for _ in range(Hn)]).astype(np.int64)
        # a perturbed ground-truth pose as the prior (non-unit translation: setup_prior normalises it in place, :183)
        ang = 0.05
        dR = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
        prior = np.concatenate([dR @ Rgt, (1.7 * tgt + 0.03)[:, None]], 1).astype(np.float32)
        pcl = np.random.RandomState(0).uniform(low=-3.0, high=3.0, size=(300, 3)).astype(np.float32)
        pp = {'rotation_pcl_error': True, 'rotation_error': False, 'K1': torch.eye(3), 'K2': torch.eye(3),
              'RT': torch.FloatTensor(prior.copy()), 'pcl': torch.FloatTensor(pcl), 'lambda': 0.3,
              'biased_sampling': 'biased'} if with_prior else {}
        rs = RANSAC(model_type='essential_cv2', max_iter=1, inl_th=3e-7, prior_params=pp, max_lo_iters=0, batch_size=Hn,
                    use_noexp_prior_scoring=with_prior, use_linear_bias_sampling=with_prior,
                    **({'bias_sigma_sq': 0.1} if with_prior else {}))
        rs.minimal_solver = run_8point          # ransac.py:143
        rs.minimal_sample_size = 8              # ransac.py:144
        seen = {}

        def sample(sample_size, pop_size, batch_size, weight=None, device=None, _s=samples, _seen=seen):
            assert sample_size == 8 and batch_size == len(_s)
            _seen['weight'] = None if weight is None else weight.detach().clone()
            return torch.from_numpy(_s)
        rs.sample = sample
        rbm = rs.remove_bad_models

        def remove_bad_models(models, _seen=seen):
            _seen['models_all'] = models.detach().clone()
            kept = rbm(models)
            diag = torch.diagonal(models, dim1=1, dim2=2)
            _seen['keep'] = (diag.abs().min(dim=1)[0] > 1e-4)
            return kept
        rs.remove_bad_models = remove_bad_models
        ver = rs.verify

        def verify(kp1_, kp2_, models, inl_th, prior_score, _seen=seen, _rs=rs):
            _seen['prior_score'] = prior_score.detach().clone()
            errs = _rs.error_fn(kp1_[None].expand(len(models), -1, 2), kp2_[None].expand(len(models), -1, 2), models, squared=True)
            _seen['count'] = (errs <= inl_th).sum(1)
            tot = (errs <= inl_th).to(kp1_).sum(1) + prior_score.to(kp1_)
            _seen['best_kept'] = int(tot.argmax())
            _seen['score'] = tot.detach().clone()
            _seen['err_best'] = errs[_seen['best_kept']].detach().clone()
            return ver(kp1_, kp2_, models, inl_th, prior_score)
        rs.verify = verify
        with torch.no_grad():
            E, inl, tight, ultra = rs.forward(kp1=kp1, kp2=kp2)
        keep = seen['keep'].numpy()
        best_all = int(np.nonzero(keep)[0][seen['best_kept']])
        out.update({f'{tag}_kpts0': k0, f'{tag}_kpts1': k1, f'{tag}_K': K, f'{tag}_samples': samples,
                    f'{tag}_models': seen['models_all'].numpy(), f'{tag}_keep': keep,
                    f'{tag}_prior_score': seen['prior_score'].numpy().astype(np.float32),
                    f'{tag}_count': seen['count'].numpy(), f'{tag}_score': seen['score'].numpy(),
                    f'{tag}_best': best_all, f'{tag}_E': E.numpy(), f'{tag}_inliers': inl.numpy().reshape(-1),
                    f'{tag}_tight': tight.numpy().reshape(-1), f'{tag}_ultra': ultra.numpy().reshape(-1),
                    f'{tag}_err_best': seen['err_best'].numpy(), f'{tag}_R_gt': Rgt, f'{tag}_t_gt': tgt})
        if with_prior:
            out.update({'p_prior': prior, 'p_pcl': pcl, 'p_bias_weight': seen['weight'].numpy()})
        print(tag, 'kept', int(keep.sum()), 'best', best_all, 'count', int(seen['count'][seen['best_kept']]),
              'inliers', int(inl.sum()), int(tight.sum()), int(ultra.sum()))
    save('g12_ransac_loop', note=NOTE_KORNIA, **out)


def g13_mapfree_corr_volume_warp():
    """CorrelationVolumeWarping.forward (mapfree_6dreg/lib/models/regression/aggregator.py:44-115) in the FAR configuration
    (rot6d_trans_with_loftr.yaml: POSITION_ENCODER, MAX_SCORE_CHANNEL), run from the reference's own module: a small
    case stored in full and the 92 x 68 grid of the 360 x 270 regression images stored as samples + checksums."""
    import types
    sys.path.insert(0, '/root/reference/mapfree_6dreg')
    from lib.models.regression.aggregator import CorrelationVolumeWarping
    cfg = types.SimpleNamespace(POSITION_ENCODER=True, POSITION_ENCODER_IM1=None, MAX_SCORE_CHANNEL=True, CV_OUTLAYERS=0,
                                CV_HALF_CHANNELS=False, UPSAMPLE_POS_ENC=0, DUSTBIN=False, NORMALISE_DOT=False)
    mod = CorrelationVolumeWarping(cfg, 32).eval()
    assert mod.num_out_layers == 67
    rng = np.random.default_rng(33)
    out = {}
    for tag, (B, H, W, amp) in {'s': (2, 12, 9, 0.6), 'f': (1, 92, 68, 0.45)}.items():
        v0 = (amp * rng.standard_normal((B, 32, H, W))).astype(np.float32)
        v1 = (amp * rng.standard_normal((B, 32, H, W))).astype(np.float32)
        # correlated halves, so that some rows have a confident peak and others are diffuse
        v1[:, :, : H // 2] = 2.0 * v0[:, :, : H // 2][:, :, ::-1] + 0.3 * v1[:, :, : H // 2]
        with torch.no_grad():
            agg = mod(torch.from_numpy(v0), torch.from_numpy(v1)).numpy()
        out[tag + '_shape'] = np.array([B, H, W])
        out[tag + '_amp'] = amp
        if tag == 's':
            out.update(s_vol0=v0, s_vol1=v1, s_agg=agg)
        else:
            out.update(f_agg_sample=agg[:, :, ::7, ::5], f_agg_sum=agg.astype(np.float64).sum((2, 3)),
                       f_max_score=agg[:, 66])
    save('g13_mapfree_cvw', seed=33, **out)


def g14_spvs_coarse():
    """spvs_coarse (supervision.py:34-137) + warp_kpts (geometry.py:5-56) of the reference on synthetic planar scenes."""
    from src.loftr.utils.supervision import spvs_coarse
    d0, d1, T01, T10, K = spvs_scene()
    N = len(d0)
    data = {'image0': torch.zeros(N, 1, 480, 640), 'image1': torch.zeros(N, 1, 480, 640), 'depth0': torch.from_numpy(d0),
            'depth1': torch.from_numpy(d1), 'T_0to1': torch.from_numpy(T01), 'T_1to0': torch.from_numpy(T10),
            'K0': torch.from_numpy(K), 'K1': torch.from_numpy(K), 'pair_names': [('a', 'b')] * N}
    spvs_coarse(data, {'LOFTR': {'RESOLUTION': (8, 2)}})
    gt = data['conf_matrix_gt']
    save('g14_spvs_coarse', seed=61, b_ids=data['spv_b_ids'].numpy(), i_ids=data['spv_i_ids'].numpy(), j_ids=data['spv_j_ids'].numpy(),
         w_pt0_i_sample=data['spv_w_pt0_i'][:, ::37].numpy(), pt1_i_sample=data['spv_pt1_i'][:, ::37].numpy(),
         gt_sum=np.float64(gt.sum().item()), gt_rowsum=gt.sum(2).numpy().astype(np.int8))
    print('g14: GT matches', len(data['spv_b_ids']))


def g15_losses():
    """LoFTRLoss.forward (loftr_loss.py:294-356) and spvs_fine (supervision.py:142-166) of the reference."""
    from src.losses.loftr_loss import LoFTRLoss
    from src.loftr.utils.supervision import spvs_fine
    from far_amd.config import far_train_config
    x = loss_inputs()
    out = {}
    for tag, l1, none_correct in (('l1', True, False), ('l2', False, False), ('nocorrect', True, True)):
        cfg = far_train_config()
        cfg['loftr']['loss']['use_l1_rt_loss'] = l1
        lf = LoFTRLoss(cfg).train()
        gtf = x['expec_f_gt'] + (5.0 if none_correct else 0.0)
        data = {'conf_matrix': torch.from_numpy(x['conf']), 'conf_matrix_gt': torch.from_numpy(x['gt']),
                'expec_f': torch.from_numpy(x['expec_f']), 'expec_f_gt': torch.from_numpy(gtf),
                'expec_rt': torch.from_numpy(x['expec_rt']), 'T_0to1': torch.from_numpy(x['T']),
                'num_correspondences_after_ransac': 0, 'num_correspondences_before_ransac': 0}
        lf(data)
        out[f'loss_{tag}'] = data['loss'].numpy()
        for k in ('loss_c', 'loss_f', 'loss_rot', 'loss_tr'):
            out[f'{k}_{tag}'] = np.float64(data['loss_scalars'][k].item())
    # corner case loftr_loss.py:65-70: no ground-truth coarse match at all (conf_matrix_gt all zero)
    lf = LoFTRLoss(far_train_config()).train()
    data = {'conf_matrix': torch.from_numpy(x['conf']), 'conf_matrix_gt': torch.zeros_like(torch.from_numpy(x['gt'])),
            'expec_f': torch.from_numpy(x['expec_f']), 'expec_f_gt': torch.from_numpy(x['expec_f_gt']),
            'expec_rt': torch.from_numpy(x['expec_rt']), 'T_0to1': torch.from_numpy(x['T']),
            'num_correspondences_after_ransac': 0, 'num_correspondences_before_ransac': 0}
    lf(data)
    out['loss_nogt'] = data['loss'].numpy()
    out['loss_c_nogt'] = np.float64(data['loss_scalars']['loss_c'].item())
    data = {'spv_w_pt0_i': torch.from_numpy(x['w_pt0']), 'spv_pt1_i': torch.from_numpy(x['pt1']), 'b_ids': torch.from_numpy(x['b_ids']),
            'i_ids': torch.from_numpy(x['i_ids']), 'j_ids': torch.from_numpy(x['j_ids'])}
    spvs_fine(data, {'LOFTR': {'RESOLUTION': (8, 2), 'FINE_WINDOW_SIZE': 5}})
    save('g15_losses', seed=71, expec_f_gt=data['expec_f_gt'].numpy(), **out)
    print('g15:', {k: float(np.ravel(v)[0]) for k, v in out.items()})


def g16_eval_metrics():
    """compute_symmetrical_epipolar_errors (metrics.py:58-77), compute_pose_errors (:198-303; its solver call replaced by
    a stub that returns committed fits -- cv2 is absent -- so that the reference's BOOKKEEPING runs: successful_fits, the
    failed-fit convention, the counts, pred_R / pred_t), epidist_prec (:326-337) and aggregate_metrics (:339-377)."""
    import src.utils.metrics as rm
    from tests.util import eval_batch, eval_metrics_table
    x = eval_batch()
    B = len(x['T'])
    t = lambda a: torch.from_numpy(a)
    data = {'T_0to1': t(x['T']), 'K0': t(x['K0']), 'K1': t(x['K1']), 'm_bids': t(x['m_bids']),
            'mkpts0_f': t(x['mk0']), 'mkpts1_f': t(x['mk1'])}
    rm.compute_symmetrical_epipolar_errors(data)
    out = {'epi_errs': data['epi_errs'].numpy()}
    cfg = ref_shim._AttrDict(TRAINER=ref_shim._AttrDict(RANSAC_PIXEL_THR=0.5, RANSAC_CONF=0.99999),
                             LOFTR=ref_shim._AttrDict(SOLVER='prior_ransac'), SAVE_PREDS=None)
    # (1) the head branch, one pair per call as the reference evaluates (batch size 1)
    errs = []
    for b in range(B):
        d = {'T_0to1': t(x['T'][b:b + 1]), 'K0': t(x['K0'][b:b + 1]), 'K1': t(x['K1'][b:b + 1]),
             'regressed_rt': t(x['regressed_rt'][b:b + 1])}
        rm.compute_pose_errors(d, cfg)
        errs.append([d['R_errs'][0], d['t_errs'][0], d['t_errs_abs'][0], d['successful_fits'][0]])
        last = d
    out.update(head_errs=np.array(errs, np.float64), head_pred_R=np.asarray(last['pred_R']), head_pred_t=np.asarray(last['pred_t']))
    # (2) the solver branch with committed fits (pair 1 fails)
    calls = []

    def stub(k0, k1, K0, K1, thr, conf=None, translation_scale=None, solver=None, priorRT=None):
        b = len(calls)
        calls.append((len(k0), solver, None if priorRT is None else np.asarray(priorRT).copy()))
        if x['fit_ok'][b] == 0:
            return None, 0, 0, 0
        mask = x['fit_mask'][x['m_bids'] == b] > 0
        return (t(x['fit_R'][b]), t(x['fit_t'][b]), mask, t(np.eye(3))), torch.tensor(int(mask.sum())), 0, 0
    rm.estimate_pose = stub
    d = dict(data, translation_scale=None, priorRT=x['priorRT'])
    rm.compute_pose_errors(d, cfg)
    out.update(fit_errs=np.array([d['R_errs'], d['t_errs'], d['t_errs_abs'], d['successful_fits']], np.float64).T,
               fit_inlier_sums=np.array([np.asarray(i).sum() for i in d['inliers']], np.float64),
               fit_inlier_lens=np.array([len(i) for i in d['inliers']]),
               fit_before=np.array(d['num_correspondences_before_ransac']), fit_after=np.array([int(v) for v in d['num_correspondences_after_ransac']]),
               fit_pred_R=np.asarray(d['pred_R']), fit_pred_t=np.asarray(d['pred_t']),
               stub_saw_prior=np.array([c[2] is not None for c in calls]), stub_counts=np.array([c[0] for c in calls]))
    # (3) no correspondences in the dict at all
    d = {'T_0to1': t(x['T'][:1]), 'K0': t(x['K0'][:1]), 'K1': t(x['K1'][:1])}
    np.random.seed(0)
    rm.compute_pose_errors(d, cfg)
    out.update(none_errs=np.array([d['R_errs'][0], d['t_errs'][0], d['t_errs_abs'][0], d['successful_fits'][0]], np.float64))
    # (4) aggregation over a gathered table with DistributedSampler duplicates and a pair without matches
    m = eval_metrics_table()
    agg = rm.aggregate_metrics({k: (list(v) if isinstance(v, (list, np.ndarray)) else v) for k, v in m.items()}, 5e-4)
    out.update(agg_keys=np.array(list(agg.keys())), agg_vals=np.array([float(v) for v in agg.values()], np.float64))
    pr = rm.epidist_prec(np.array(m['epi_errs'], dtype=object), [1e-4, 5e-4, 1e-3], True)
    out.update(prec_keys=np.array(list(pr.keys())), prec_vals=np.array([float(v) for v in pr.values()], np.float64))
    save('g16_eval_metrics', seed=81, **out)
    print('g16:', dict(zip(out['agg_keys'].tolist(), out['agg_vals'].tolist())))


def five_point_samples(kind, n, seed):
    """n calibrated five-point samples (float64) of two-view scenes: 'general' depths, 'two_planes' (3 + 2 points on two planes),
    'plane' (five coplanar points).  Returns p1, p2 (n, 5, 2) and the true E (n, 3, 3), unit Frobenius norm."""
    rng = np.random.default_rng(seed)
    P1, P2, Et = [], [], []
    for _ in range(n):
        ang = rng.uniform(-0.4, 0.4, 3)
        cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
        R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
             @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
        t = rng.uniform(-1, 1, 3)
        X = np.stack([rng.uniform(-2, 2, 5), rng.uniform(-1.5, 1.5, 5), rng.uniform(3, 8, 5)], 1)
        if kind == 'plane':
            X[:, 2] = 5 + 0.2 * X[:, 0] - 0.1 * X[:, 1]
        elif kind == 'two_planes':
            X[:3, 2] = 5 + 0.3 * X[:3, 0]
            X[3:, 2] = 4 - 0.2 * X[3:, 1]
        X2 = X @ R.T + t
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        E = tx @ R
        P1.append(X[:, :2] / X[:, 2:]); P2.append(X2[:, :2] / X2[:, 2:]); Et.append(E / np.linalg.norm(E))
    return np.stack(P1), np.stack(P2), np.stack(Et)


def g17_fivepoint():
    """The reference's torch five-point solver and its loop.
    (a) run_5point_our_kornia (third_party/prior_ransac/cv_geometry.py:861-1043) on committed five-point samples -- general,
        two-plane and coplanar -- one sample per call (its singular_filter :959-961 drops samples from a batch without saying
        which), in float64 (the function is dtype-agnostic; RANSAC feeds it float32) and in float32: the ten models per sample
        (the real parts of ALL ten companion-matrix eigenvalues, :994: models of complex roots included, identity rows when a
        sample is dropped).
    (b) RANSAC(model_type='essential') (ransac.py:146-150: that solver, sample size 5, Sampson error) through RANSAC.forward
        (:340-442) with `sample` replaced by committed index sets, as G12 does for the 8-point: models, remove_bad_models mask,
        per-model inlier counts, best model, the three masks; once without and once with a prior.
    kornia.geometry.solvers: multiply_deg_one_poly / multiply_deg_two_one_poly are tools/ref_shim.py restatements (NOTE below),
    determinant_to_polynomial is the reference's own copy (cv_geometry.py:23-551)."""
    cvg = ref_shim.bind_reference_solvers()
    from ransac import RANSAC
    out = {}
    with torch.no_grad():
        for kind, n, seed in (('general', 40, 171), ('two_planes', 20, 172), ('plane', 20, 173)):
            p1, p2, Et = five_point_samples(kind, n, seed)
            m64 = np.zeros((n, 10, 3, 3))
            m32 = np.zeros((n, 10, 3, 3), np.float32)
            for i in range(n):
                a, b = torch.from_numpy(p1[i:i + 1]), torch.from_numpy(p2[i:i + 1])
                try:
                    m64[i] = cvg.run_5point_our_kornia(a, b, torch.ones(1, 5, dtype=torch.float64)).numpy().reshape(10, 3, 3)
                except Exception as e:             # every sample of the batch filtered out: torch.cat of an empty list
                    print('g17', kind, i, 'float64: no model:', type(e).__name__)
                    m64[i] = np.eye(3)
                try:
                    m32[i] = cvg.run_5point_our_kornia(a.float(), b.float(), torch.ones(1, 5)).numpy().reshape(10, 3, 3)
                except Exception as e:
                    print('g17', kind, i, 'float32: no model:', type(e).__name__)
                    m32[i] = np.eye(3)
            out.update({f'{kind}_p1': p1, f'{kind}_p2': p2, f'{kind}_E_true': Et, f'{kind}_models64': m64, f'{kind}_models32': m32})
            x1 = np.concatenate([p1, np.ones((n, 5, 1))], -1)
            x2 = np.concatenate([p2, np.ones((n, 5, 1))], -1)
            epi = np.abs(np.einsum('hsi,hkij,hsj->hks', x2, m64, x1)).max(-1)
            d = np.minimum(np.abs(m64 - Et[:, None]).max((-1, -2)), np.abs(m64 + Et[:, None]).max((-1, -2))).min(1)
            print(f'g17 {kind}: models with |x2^T E x1| < 1e-9 per sample: mean {float((epi < 1e-9).sum(1).mean()):.2f}; '
                  f'truth among the models (1e-6): {100 * float((d < 1e-6).mean()):.1f} %')
        # ---- (b) the loop
        for tag, seed, with_prior in (('p', 177, True), ('n', 178, False)):
            k0, k1, K, Rgt, tgt = two_view_scene(400, seed=seed, outlier_frac=0.3)
            kn0 = ((torch.from_numpy(k0) - torch.from_numpy(K)[[0, 1], [2, 2]][None]) / torch.from_numpy(K)[[0, 1], [0, 1]][None]).numpy()
            kn1 = ((torch.from_numpy(k1) - torch.from_numpy(K)[[0, 1], [2, 2]][None]) / torch.from_numpy(K)[[0, 1], [0, 1]][None]).numpy()
            kp1, kp2 = torch.FloatTensor(kn0), torch.FloatTensor(kn1)
            rng = np.random.default_rng(seed)
            Hn = 96
            samples = np.stack([rng.choice(len(k0), 5, replace=False) for _ in range(Hn)]).astype(np.int64)
            ang = 0.05
            dR = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
            prior = np.concatenate([dR @ Rgt, (1.7 * tgt + 0.03)[:, None]], 1).astype(np.float32)
            pcl = np.random.RandomState(0).uniform(low=-3.0, high=3.0, size=(300, 3)).astype(np.float32)
            pp = {'rotation_pcl_error': True, 'rotation_error': False, 'K1': torch.eye(3), 'K2': torch.eye(3),
                  'RT': torch.FloatTensor(prior.copy()), 'pcl': torch.FloatTensor(pcl), 'lambda': 0.3,
                  'biased_sampling': 'biased'} if with_prior else {}
            rs = RANSAC(model_type='essential', max_iter=1, inl_th=3e-7, prior_params=pp, max_lo_iters=0, batch_size=Hn,
                        use_noexp_prior_scoring=with_prior, use_linear_bias_sampling=with_prior,
                        **({'bias_sigma_sq': 0.1} if with_prior else {}))
            seen = {}

            def sample(sample_size, pop_size, batch_size, weight=None, device=None, _s=samples, _seen=seen):
                assert sample_size == 5 and batch_size == len(_s)
                _seen['weight'] = None if weight is None else weight.detach().clone()
                return torch.from_numpy(_s)
            rs.sample = sample
            rbm = rs.remove_bad_models

            def remove_bad_models(models, _seen=seen):
                _seen['models_all'] = models.detach().clone()
                diag = torch.diagonal(models, dim1=1, dim2=2)
                _seen['keep'] = (diag.abs().min(dim=1)[0] > 1e-4)
                return rbm(models)
            rs.remove_bad_models = remove_bad_models
            ver = rs.verify

            def verify(kp1_, kp2_, models, inl_th, prior_score, _seen=seen, _rs=rs):
                _seen['prior_score'] = prior_score.detach().clone()
                errs = _rs.error_fn(kp1_[None].expand(len(models), -1, 2), kp2_[None].expand(len(models), -1, 2), models, squared=True)
                _seen['count'] = (errs <= inl_th).sum(1)
                tot = (errs <= inl_th).to(kp1_).sum(1) + prior_score.to(kp1_)
                _seen['best_kept'] = int(tot.argmax())
                _seen['score'] = tot.detach().clone()
                _seen['err_best'] = errs[_seen['best_kept']].detach().clone()
                return ver(kp1_, kp2_, models, inl_th, prior_score)
            rs.verify = verify
            E, inl, tight, ultra = rs.forward(kp1=kp1, kp2=kp2)
            keep = seen['keep'].numpy()
            assert len(seen['models_all']) == 10 * Hn, 'a sample was dropped by singular_filter: model <-> sample bookkeeping lost'
            best_all = int(np.nonzero(keep)[0][seen['best_kept']])
            out.update({f'{tag}_kpts0': k0, f'{tag}_kpts1': k1, f'{tag}_K': K, f'{tag}_samples': samples,
                        f'{tag}_models': seen['models_all'].numpy(), f'{tag}_keep': keep,
                        f'{tag}_prior_score': seen['prior_score'].numpy().astype(np.float32),
                        f'{tag}_count': seen['count'].numpy(), f'{tag}_score': seen['score'].numpy(),
                        f'{tag}_best': best_all, f'{tag}_E': E.numpy(), f'{tag}_inliers': inl.numpy().reshape(-1),
                        f'{tag}_tight': tight.numpy().reshape(-1), f'{tag}_ultra': ultra.numpy().reshape(-1),
                        f'{tag}_err_best': seen['err_best'].numpy(), f'{tag}_R_gt': Rgt, f'{tag}_t_gt': tgt})
            if with_prior:
                out.update({'p_prior': prior, 'p_pcl': pcl, 'p_bias_weight': seen['weight'].numpy()})
            print(tag, 'models', len(keep), 'kept', int(keep.sum()), 'best', best_all, 'count', int(seen['count'][seen['best_kept']]),
                  'inliers', int(inl.sum()), int(tight.sum()), int(ultra.sum()))
    save('g17_fivepoint', note=NOTE_KORNIA + '; kornia.geometry.solvers.multiply_deg_one_poly / multiply_deg_two_one_poly restated in '
         'tools/ref_shim.py, determinant_to_polynomial = the reference\'s own cv_geometry.py:23-551', **out)


def g18_masked_training_coarse():
    """CoarseMatching.forward in TRAINING mode on a padded-mask batch (coarse_matching.py:86-147: masked_fill of the similarity,
    dual softmax; get_coarse_match :150-265 with mask_border_with_padding :28-43, compute_max_candidates :46-57 and the training-time
    sampling / GT padding :199-240), run from the reference's own module with a seeded generator; and its eval-mode output on the
    same tensors."""
    from src.loftr.utils.coarse_matching import CoarseMatching
    from tests.util import masked_coarse_inputs
    inp = masked_coarse_inputs()
    cfg = ref_shim.far_eval_config()['match_coarse']
    h, w = inp['h'], inp['w']
    out = {}
    for mode in ('train', 'eval'):
        cm = CoarseMatching(cfg)
        cm.train(mode == 'train')
        data = {'hw0_i': (8 * h, 8 * w), 'hw1_i': (8 * h, 8 * w), 'hw0_c': (h, w), 'hw1_c': (h, w),
                'mask0': torch.from_numpy(inp['mask0']), 'mask1': torch.from_numpy(inp['mask1']),
                'spv_b_ids': torch.from_numpy(inp['spv_b_ids']), 'spv_i_ids': torch.from_numpy(inp['spv_i_ids']),
                'spv_j_ids': torch.from_numpy(inp['spv_j_ids'])}
        f0 = torch.from_numpy(inp['f0']).requires_grad_(mode == 'train')
        f1 = torch.from_numpy(inp['f1']).requires_grad_(mode == 'train')
        torch.manual_seed(1234)
        cm(f0, f1, data, mask_c0=data['mask0'].flatten(-2), mask_c1=data['mask1'].flatten(-2))
        conf = data['conf_matrix']
        out.update({f'{mode}_b_ids': data['b_ids'].numpy(), f'{mode}_i_ids': data['i_ids'].numpy(), f'{mode}_j_ids': data['j_ids'].numpy(),
                    f'{mode}_gt_mask': data['gt_mask'].numpy(), f'{mode}_m_bids': data['m_bids'].numpy(),
                    f'{mode}_mkpts0_c': data['mkpts0_c'].numpy(), f'{mode}_mkpts1_c': data['mkpts1_c'].numpy(),
                    f'{mode}_mconf': data['mconf'].detach().numpy(), f'{mode}_conf_sum': conf.detach().sum((1, 2)).numpy(),
                    f'{mode}_conf_sample': conf.detach()[:, ::37, ::41].numpy()})
        if mode == 'train':                      # a gradient through the masked dual softmax: d sum(conf[gt]) / d feat
            pos = conf[data['spv_b_ids'], data['spv_i_ids'], data['spv_j_ids']]
            pos.sum().backward()
            out.update({'train_pos_conf': pos.detach().numpy(), 'train_df0_sample': f0.grad[:, ::53, ::17].numpy(),
                        'train_df0_norm': np.array(f0.grad.norm().item()), 'train_df1_norm': np.array(f1.grad.norm().item())})
        print('g18', mode, 'matches', len(data['b_ids']), 'kept', len(data['mconf']), 'gt', len(inp['spv_b_ids']))
    save('g18_masked_training_coarse', **out)


def g8_manifest(m):
    man = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(OUT, 'g8_state_dict_manifest.json'), 'w') as f:
        json.dump(man, f, indent=0)
    print('g8 manifest:', len(man), 'tensors', sum(int(np.prod(s)) if s else 1 for s in man.values()), 'elements')


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == 'g9':
        g9_metrics()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g10':
        g10_training(ref_model()[0])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g11':
        g11_matcher_544x720(ref_model()[0])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g16':
        g16_eval_metrics()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g15':
        g15_losses()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g14':
        g14_spvs_coarse()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g13':
        g13_mapfree_corr_volume_warp()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g12':
        g12_ransac_loop()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g18':
        g18_masked_training_coarse()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g17':
        g17_fivepoint()
        sys.exit(0)
    g1_coarse()
    g9_metrics()
    g5_solver()
    g6_pose6d()
    model, _ = ref_model()
    g8_manifest(model)
    g2_fine(model)
    g3_encoder(model)
    g4_head(model)
    g7_full(model)
    g10_training(model)
    g11_matcher_544x720(model)
    g12_ransac_loop()
    g13_mapfree_corr_volume_warp()
    g14_spvs_coarse()
    g15_losses()
    g16_eval_metrics()
    g17_fivepoint()
    g18_masked_training_coarse()

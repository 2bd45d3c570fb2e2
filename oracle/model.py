"""Oracle for the whole hot path: backbone -> coarse transformer -> K1 -> fine -> K4 -> EMM head
(test infrastructure only, see oracle/__init__.py).

A functional restatement driven by a {name: ndarray} state dict with the reference's parameter names
(SURVEY.md Appendix B).  Convolutions / Linear / LayerNorm use torch-CPU fp32 functional ops (they are the
same vendor-library ops on both sides of the parity check, not kernels of this build); the operators this
build replaces go through the numpy restatements in oracle/{coarse,fine,attention,head,solver}.py.

Follows mp3d_loftr/src/loftr/loftr.py:56-192, backbone/resnet_fpn.py:101-119, utils/position_encoding.py,
loftr_module/transformer.py:44-67, :90-112, :266-303, :335-348, :423-483, and the evaluation call order of
src/lightning/lightning_loftr.py:325-343.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import attention as oattn
from . import coarse as ocoarse
from . import fine as ofine
from . import head as ohead
from . import solver as osolver

POSE_MEAN = np.array([-0.34898765, 0.17085525, -0.87944315, 0.50275223, 0.03533648, -0.18179045,
                      -0.03533648, 0.98189617, 0.09313615], np.float32)       # loftr_loss.py:7
POSE_STD = np.array([1.94014405, 0.36770130, 1.88317520, 0.51837117, 0.12717603, 0.65426397,
                     0.12717603, 0.0188729, 0.09709263], np.float32)          # loftr_loss.py:8


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class Weights:
    def __init__(self, sd):
        self.sd = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items()}

    def t(self, name):
        return _t(self.sd[name])

    def has(self, name):
        return name in self.sd


# ---------------------------------------------------------------------------------------------------------
# backbone (resnet_fpn.py)
# ---------------------------------------------------------------------------------------------------------
def _bn(w, p, x):
    return F.batch_norm(x, w.t(p + '.running_mean'), w.t(p + '.running_var'), w.t(p + '.weight'), w.t(p + '.bias'),
                        training=False, eps=1e-5)


def _block(w, p, x, stride):
    y = F.relu(_bn(w, p + '.bn1', F.conv2d(x, w.t(p + '.conv1.weight'), stride=stride, padding=1)))
    y = _bn(w, p + '.bn2', F.conv2d(y, w.t(p + '.conv2.weight'), padding=1))
    if stride != 1:
        x = _bn(w, p + '.downsample.1', F.conv2d(x, w.t(p + '.downsample.0.weight'), stride=stride))
    return F.relu(x + y)


def backbone(w, img):
    """resnet_fpn.py:101-119.  img (N,1,H,W) -> feats_c (N,256,H/8,W/8), feats_f (N,128,H/2,W/2)."""
    p = 'backbone.'
    x0 = F.relu(_bn(w, p + 'bn1', F.conv2d(_t(img), w.t(p + 'conv1.weight'), stride=2, padding=3)))
    x1 = _block(w, p + 'layer1.1', _block(w, p + 'layer1.0', x0, 1), 1)
    x2 = _block(w, p + 'layer2.1', _block(w, p + 'layer2.0', x1, 2), 1)
    x3 = _block(w, p + 'layer3.1', _block(w, p + 'layer3.0', x2, 2), 1)
    x3o = F.conv2d(x3, w.t(p + 'layer3_outconv.weight'))
    up3 = F.interpolate(x3o, scale_factor=2., mode='bilinear', align_corners=True)
    x2o = F.conv2d(x2, w.t(p + 'layer2_outconv.weight')) + up3
    x2o = F.conv2d(x2o, w.t(p + 'layer2_outconv2.0.weight'), padding=1)
    x2o = F.leaky_relu(_bn(w, p + 'layer2_outconv2.1', x2o))
    x2o = F.conv2d(x2o, w.t(p + 'layer2_outconv2.3.weight'), padding=1)
    up2 = F.interpolate(x2o, scale_factor=2., mode='bilinear', align_corners=True)
    x1o = F.conv2d(x1, w.t(p + 'layer1_outconv.weight')) + up2
    x1o = F.conv2d(x1o, w.t(p + 'layer1_outconv2.0.weight'), padding=1)
    x1o = F.leaky_relu(_bn(w, p + 'layer1_outconv2.1', x1o))
    x1o = F.conv2d(x1o, w.t(p + 'layer1_outconv2.3.weight'), padding=1)
    return x3o.numpy(), x1o.numpy()


def position_encoding(d_model, h, w, temp_bug_fix=True):
    """position_encoding.py:21-33 cropped to (h, w): returns (d_model, h, w) float32."""
    ys = torch.ones(h, w).cumsum(0).float().unsqueeze(0)
    xs = torch.ones(h, w).cumsum(1).float().unsqueeze(0)
    k = torch.arange(0, d_model // 2, 2).float()
    if temp_bug_fix:
        div = torch.exp(k * (-math.log(10000.0) / (d_model // 2)))
    else:
        div = torch.exp(k * (-math.log(10000.0) / d_model // 2))
    div = div[:, None, None]
    pe = torch.zeros(d_model, h, w)
    pe[0::4] = torch.sin(xs * div)
    pe[1::4] = torch.cos(xs * div)
    pe[2::4] = torch.sin(ys * div)
    pe[3::4] = torch.cos(ys * div)
    return pe.numpy()


# ---------------------------------------------------------------------------------------------------------
# LoFTR encoder layers (transformer.py:44-67, :90-112)
# ---------------------------------------------------------------------------------------------------------
def encoder_layer(w, p, x, source, nhead):
    xt, st = _t(x), _t(source)
    q = F.linear(xt, w.t(p + '.q_proj.weight')).numpy()
    k = F.linear(st, w.t(p + '.k_proj.weight')).numpy()
    v = F.linear(st, w.t(p + '.v_proj.weight')).numpy()
    msg = oattn.linear_attention(q, k, v, nhead)                                     # linear_attention.py
    C = x.shape[-1]
    msg = F.linear(_t(msg), w.t(p + '.merge.weight'))
    msg = F.layer_norm(msg, (C,), w.t(p + '.norm1.weight'), w.t(p + '.norm1.bias'))
    msg = F.linear(F.relu(F.linear(torch.cat([xt, msg], dim=2), w.t(p + '.mlp.0.weight'))), w.t(p + '.mlp.2.weight'))
    msg = F.layer_norm(msg, (C,), w.t(p + '.norm2.weight'), w.t(p + '.norm2.bias'))
    return (xt + msg).numpy()


def feature_transformer(w, p, feat0, feat1, layer_names, nhead):
    for i, name in enumerate(layer_names):
        lp = f'{p}.layers.{i}'
        if name == 'self':
            feat0 = encoder_layer(w, lp, feat0, feat0, nhead)
            feat1 = encoder_layer(w, lp, feat1, feat1, nhead)
        else:
            feat0 = encoder_layer(w, lp, feat0, feat1, nhead)
            feat1 = encoder_layer(w, lp, feat1, feat0, nhead)                        # updated feat0 (:107-108)
    return feat0, feat1


# ---------------------------------------------------------------------------------------------------------
# matcher forward (loftr.py:56-135)
# ---------------------------------------------------------------------------------------------------------
def matcher_forward(w, cfg, image0, image1, from_featmaps=None):
    """Returns the data dict (numpy) the reference's LoFTR.forward would have produced (eval path)."""
    data = {}
    if from_featmaps is None:
        N = image0.shape[0]
        fc, ff = backbone(w, np.concatenate([image0, image1], 0))
        fc0, fc1, ff0, ff1 = fc[:N], fc[N:], ff[:N], ff[N:]
        hw_i = image0.shape[2:]
    else:
        fc0, fc1, ff0, ff1, hw_i = from_featmaps
        N = fc0.shape[0]
    hw_c, hw_f = fc0.shape[2:], ff0.shape[2:]
    C = fc0.shape[1]
    pe = position_encoding(C, hw_c[0], hw_c[1], cfg['coarse']['temp_bug_fix'])
    f0 = (fc0 + pe[None]).reshape(N, C, -1).transpose(0, 2, 1).copy()
    f1 = (fc1 + pe[None]).reshape(N, C, -1).transpose(0, 2, 1).copy()
    f0, f1 = feature_transformer(w, 'loftr_coarse', f0, f1, cfg['coarse']['layer_names'], cfg['coarse']['nhead'])
    cm = ocoarse.coarse_matching(f0, f1, cfg['match_coarse'], hw_c, hw_c, hw_i)
    data.update(cm)
    b, i, j = cm['b_ids'], cm['i_ids'], cm['j_ids']
    W = cfg['fine_window_size']
    stride = hw_f[0] // hw_c[0]
    M = len(b)
    if M:
        w0 = ofine.unfold_windows(ff0, b, i, hw_c[1], W, stride)                     # fine_preprocess.py:40-47
        w1 = ofine.unfold_windows(ff1, b, j, hw_c[1], W, stride)
        cwin = F.linear(_t(np.concatenate([f0[b, i], f1[b, j]], 0)), w.t('fine_preprocess.down_proj.weight'),
                        w.t('fine_preprocess.down_proj.bias'))                        # :50-52
        both = torch.cat([_t(np.concatenate([w0, w1], 0)), cwin.unsqueeze(1).expand(-1, W * W, -1)], -1)
        both = F.linear(both, w.t('fine_preprocess.merge_feat.weight'), w.t('fine_preprocess.merge_feat.bias'))
        w0, w1 = both[:M].numpy(), both[M:].numpy()
        w0, w1 = feature_transformer(w, 'loftr_fine', w0, w1, cfg['fine']['layer_names'], cfg['fine']['nhead'])
        scale = hw_i[0] / hw_f[0]
        expec, mk1 = ofine.fine_matching(w0, w1, cm['mkpts1_c'], (W // 2) * scale)
        data.update({'expec_f': expec, 'mkpts0_f': cm['mkpts0_c'], 'mkpts1_f': mk1})
    else:
        data.update({'expec_f': np.zeros((0, 3), np.float32), 'mkpts0_f': cm['mkpts0_c'], 'mkpts1_f': cm['mkpts1_c']})
    data.update({'featmap0': f0, 'featmap1': f1, 'hw0_c': hw_c, 'hw0_f': hw_f, 'hw0_i': hw_i, 'bs': N})
    return data


# ---------------------------------------------------------------------------------------------------------
# head (loftr.py:137-192, transformer.py:266-303, :335-348, :423-483)
# ---------------------------------------------------------------------------------------------------------
def normalized_6d(rt):
    """loftr_loss.py:31-39: rt (3,4) -> (9,)."""
    v = np.concatenate([rt[:3, 3], rt[:2, :3].reshape(-1)])
    return (v - POSE_MEAN) / POSE_STD


def rotation_6d_to_matrix(d6):
    """loftr_loss.py:10-29, d6 (6,)."""
    a1, a2 = d6[:3], d6[3:]
    b1 = a1 / max(np.linalg.norm(a1), 1e-12)
    b2 = a2 - (b1 * a2).sum() * b1
    b2 = b2 / max(np.linalg.norm(b2), 1e-12)
    return np.stack([b1, b2, np.cross(b1, b2)], 0)


def preprocess_helper(cfg, loftr_rt, num_corr, num_before, tight, ultra):
    """loftr.py:137-171 for one pair: -> loftr_preds_6d (1,13), inv (1,13) float32."""
    rt = np.asarray(loftr_rt, np.float64)
    preds = normalized_6d(rt.astype(np.float32)).astype(np.float32)
    rt44 = np.concatenate([rt, [[0, 0, 0, 1.]]], 0)
    inv = normalized_6d(np.linalg.inv(rt44)[:3, :4]).astype(np.float32)
    if cfg['regress']['regress_use_num_corres']:
        n = np.float32(num_corr) / np.float32(500)
        preds, inv = np.append(preds, n), np.append(inv, n)
    if cfg['use_many_ransac_thr']:
        n3 = np.array([num_before, tight, ultra], np.float32) / np.float32(500)
        preds, inv = np.concatenate([preds, n3]), np.concatenate([inv, n3])
    return preds[None].astype(np.float32), inv[None].astype(np.float32)


def cross_attention(w, x1, x2, pos, num_heads=4, prefix='loftr_regress.emm.cross_attn.'):
    """transformer.py:266-303 for B = 1.  x (1,N,C) -> (fundamental_2, fundamental_1), each (1,70,256).
    The 8-Point-ViT's CrossAttention (vision_transformer.py:160-208) is the same arithmetic at C = 192, 3 heads, N = 576."""
    p = prefix
    B, N, C = x1.shape
    d = C // num_heads

    def qkv(x):
        y = F.linear(_t(x), w.t(p + 'qkv.weight'), w.t(p + 'qkv.bias')).numpy()
        y = y.reshape(B, N, 3, num_heads, d).transpose(2, 0, 3, 1, 4)
        return y[0], y[1], y[2]
    q1, k1, v1 = qkv(x1)
    q2, k2, v2 = qkv(x2)
    scale = d ** -0.5
    posb = np.broadcast_to(pos[None, None], (B, num_heads, N, 6))
    v1t = np.concatenate([v1, posb], 3)
    v2t = np.concatenate([v2, posb], 3)
    f1, _ = ohead.bilinear_attention(q2, k1, v1t, scale)                              # :275,:281,:291
    f2, _ = ohead.bilinear_attention(q1, k2, v2t, scale)                              # :276,:282,:292
    ch = C + 6 * num_heads
    f1 = f1.reshape(B, ch, ch // num_heads).transpose(0, 2, 1)                        # :294-295
    f2 = f2.reshape(B, ch, ch // num_heads).transpose(0, 2, 1)
    pf = lambda f: F.linear(_t(np.ascontiguousarray(f)), w.t(p + 'proj_fundamental.weight'),
                            w.t(p + 'proj_fundamental.bias')).numpy()
    return pf(f2), pf(f1)


def head_forward(w, cfg, feat0, feat1, loftr_preds, inv_loftr_preds, pos=None):
    """LocalFeatureTransformerRegressor.forward for one pair (B = 1).  feat (1,4800,256)."""
    if cfg['regress_loftr_layers'] > 0:
        feat0, feat1 = feature_transformer(w, 'loftr_regress.loftr', feat0, feat1, cfg['regress']['layer_names'],
                                           cfg['regress']['nhead'])
    e = 'loftr_regress.emm.'
    x = np.concatenate([feat0, feat1], 0)
    if w.has(e + 'pos_embed'):
        x = x + w.sd[e + 'pos_embed']
    C = x.shape[-1]
    ln = lambda a, n, eps=1e-5: F.layer_norm(_t(a), (C,), w.t(n + '.weight'), w.t(n + '.bias'), eps).numpy()
    if pos is None:
        pos = ohead.positional_encodings()
    fa, fb = cross_attention(w, ln(x[0:1], e + 'norm1'), ln(x[1:2], e + 'norm1'), pos)          # :342
    f = np.concatenate([fa[:, None], fb[:, None]], 1).reshape(2, -1, C)                          # :344-345
    h = F.linear(F.gelu(F.linear(_t(ln(f, e + 'norm2')), w.t(e + 'mlp.fc1.weight'), w.t(e + 'mlp.fc1.bias'))),
                 w.t(e + 'mlp.fc2.weight'), w.t(e + 'mlp.fc2.bias')).numpy()
    f = f + h                                                                                    # :346
    feats = ln(f, 'loftr_regress.norm', 1e-6).reshape(1, -1)                                     # :426
    r = 'loftr_regress.'
    lin = lambda a, n: F.linear(a, w.t(r + n + '.weight'), w.t(r + n + '.bias'))
    ft = _t(feats)
    enc = lin(F.relu(lin(ft, 'encoder.0')), 'encoder.2')
    reg = lin(F.relu(lin(enc, 'pose_regressor_simple_moe.0')), 'pose_regressor_simple_moe.2').numpy()   # (1,9)
    lp = np.asarray(loftr_preds, np.float32)
    reg_t, sol_t = reg[..., :3], lp[..., :3]
    if cfg['regress']['scale_8pt']:                                                              # :436-446
        su = sol_t * POSE_STD[:3] + POSE_MEAN[:3]
        ru = reg_t * POSE_STD[:3] + POSE_MEAN[:3]
        su = su * np.linalg.norm(ru, axis=-1) / np.clip(np.linalg.norm(su, axis=-1), 1e-3, 100)
        sol_t = ((su - POSE_MEAN[:3]) / POSE_STD[:3]).astype(np.float32)
    extra = lp.shape[-1] - 9
    sol_R = lp[..., 3:-extra] if extra > 0 else lp[..., 3:]                                      # :452-455
    g = torch.cat([ft, _t(reg), _t(lp)], -1)
    g = lin(F.relu(lin(F.relu(lin(g, 'moe_predictor.0')), 'moe_predictor.2')), 'moe_predictor.4')
    gate = torch.sigmoid(g).numpy()                                                              # (1,2)
    T = gate[..., 0:1] * reg_t + (1 - gate[..., 0:1]) * sol_t                                    # :466
    R = gate[..., 1:2] * reg[..., 3:] + (1 - gate[..., 1:2]) * sol_R                             # :467
    return np.concatenate([T, R], -1).astype(np.float32), gate, feats


def prior_from_regressed(reg):
    """loftr.py:187-192: regressed_rt (1,9) -> priorRT (3,4)."""
    R6 = reg[0, 3:] * POSE_STD[3:] + POSE_MEAN[3:]
    t = reg[0, :3] * POSE_STD[:3] + POSE_MEAN[:3]
    return np.concatenate([rotation_6d_to_matrix(R6), t[:, None]], -1)


# ---------------------------------------------------------------------------------------------------------
# evaluation step (lightning_loftr.py:325-343), pair by pair as the reference runs it (batch size 1)
# ---------------------------------------------------------------------------------------------------------
def solve_pair(data, b, K0, K1, solver, priorRT, seed, H, pcl):
    sel = data['m_bids'] == b
    ret, nafter, tight, ultra, _ = osolver.estimate_pose(data['mkpts0_f'][sel], data['mkpts1_f'][sel], K0, K1, 0.5,
                                                          solver=solver, priorRT=priorRT, seed=seed, pair=b, H=H,
                                                          pcl=pcl)
    if ret is None:
        rt = np.concatenate([np.eye(3), np.zeros((3, 1))], 1)                        # supervision.py:221-224
        mask = np.zeros(int(sel.sum()), bool)
    else:
        rt = np.concatenate([ret[0], ret[1][:, None]], 1)
        mask = ret[2]
    return rt, int(sel.sum()), nafter, tight, ultra, mask


def test_step(w, cfg, image0, image1, K0, K1, seed=0, H=2048, steps=2, pcl=None):
    """Returns per-pair dicts with loftr_rt, regressed_rt, gate, counts -- the quantities the parity tests use."""
    if pcl is None:
        pcl = np.random.RandomState(0).uniform(low=-3.0, high=3.0, size=(300, 3)).astype(np.float32)
    data = matcher_forward(w, cfg, image0, image1)
    pos = ohead.positional_encodings()
    out = []
    for b in range(image0.shape[0]):
        prior = None
        res = {}
        rt, nb, na, ti, ul, mask = solve_pair(data, b, K0[b], K1[b], cfg['solver'], None, seed, H, pcl)
        for i in range(steps):
            lp, ilp = preprocess_helper(cfg, rt, na, nb, ti, ul)
            reg, gate, _ = head_forward(w, cfg, data['featmap0'][b:b + 1], data['featmap1'][b:b + 1], lp, ilp, pos)
            prior = prior_from_regressed(reg)
            res[f'loftr_rt_{i}'] = rt
            res[f'counts_{i}'] = (nb, na, ti, ul)
            res[f'mask_{i}'] = mask
            res[f'regressed_rt_{i}'] = reg
            res[f'gate_{i}'] = gate
            if i < steps - 1 and 'prior_ransac' in cfg['solver']:
                rt, nb, na, ti, ul, mask = solve_pair(data, b, K0[b], K1[b], cfg['solver'], prior, seed, H, pcl)
        res['priorRT'] = prior
        out.append(res)
    return data, out

"""Every switch and tuning key of far_amd/flags.py on a one-pair step of the headline path (matcher -> solver -> head -> solver(prior)
-> head at 640x480), held to the neutrality class the registry states: 'bitwise' = identical bits in every output, 'parity' = the
bars of DESIGN.md section 5 (another kernel / summation order of the same operator)."""
import numpy as np
import pytest
import torch

from far_amd import flags

pytestmark = pytest.mark.gpu
KEYS = ['b_ids', 'i_ids', 'j_ids', 'mconf', 'mkpts0_f', 'mkpts1_f', 'expec_f', 'loftr_rt', 'regressed_rt', 'featmap0', 'featmap1']
_BASE = {}


def _step():
    from far_amd import synth
    from far_amd.config import far_eval_config
    from far_amd.loftr import LoFTR
    from far_amd.pipeline import test_step
    m = LoFTR(far_eval_config()).eval()                       # a fresh model: the packed weight images follow the switches
    synth.load_synthetic(m, seed=0)
    m = m.cuda()
    im0, im1 = synth.synth_image_pair(1, seed=7)
    K = torch.from_numpy(synth.MP3D_K[None]).cuda()
    batch = {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
    test_step(m, batch, H=512, seed=1)
    torch.cuda.synchronize()
    return {k: batch[k].detach().cpu().numpy().copy() for k in KEYS}


def _base():
    if not _BASE:
        _BASE.update(_step())
        again = _step()
        for k in KEYS:                                        # the step itself is deterministic: the yardstick of 'bitwise'
            assert np.array_equal(_BASE[k], again[k]), k
    return _BASE


def _hold(got, neutral, what):
    base = _base()
    if neutral == 'bitwise':
        for k in KEYS:
            assert got[k].shape == base[k].shape and np.array_equal(got[k], base[k]), (what, k)
        return
    a = set(zip(base['i_ids'].tolist(), base['j_ids'].tolist()))
    b = set(zip(got['i_ids'].tolist(), got['j_ids'].tolist()))
    assert len(a & b) >= 0.995 * len(a | b), (what, len(a), len(b), len(a & b))      # (entries on the decision margin may flip)
    sc = np.abs(base['featmap0']).max()
    assert np.abs(got['featmap0'] - base['featmap0']).max() < 1e-4 * sc, what
    common = sorted(a & b)
    ia = {m: i for i, m in enumerate(zip(base['i_ids'].tolist(), base['j_ids'].tolist()))}
    ib = {m: i for i, m in enumerate(zip(got['i_ids'].tolist(), got['j_ids'].tolist()))}
    ka, kb = [ia[m] for m in common], [ib[m] for m in common]
    assert np.abs(base['mconf'][ka] - got['mconf'][kb]).max() < 1e-4, what
    assert np.abs(base['mkpts1_f'][ka] - got['mkpts1_f'][kb]).max() < 2e-2, what          # pixels
    # past the matcher the solver runs RANSAC on a slightly different correspondence set (hypotheses are drawn by index: one match more or
    # less re-deals them) and the synthetic pair constrains the pose weakly: its outputs are held to validity, not to closeness
    R = got['loftr_rt'][..., :3].reshape(3, 3)
    assert abs(np.linalg.det(R) - 1) < 1e-6 and np.isfinite(got['regressed_rt']).all(), what


@pytest.mark.parametrize('sw', [s for s in flags.SWITCHES if s.scope == 'inference'], ids=lambda s: s.env)
def test_switch_is_result_neutral(sw):
    obj, attr = flags.target(sw)
    _base()
    old = getattr(obj, attr)
    setattr(obj, attr, sw.off_value)
    try:
        got = _step()
    finally:
        setattr(obj, attr, old)
    _hold(got, sw.neutral, sw.env)


@pytest.mark.parametrize('t,v', [(t, v) for t in flags.TUNING for v in t.values], ids=lambda x: str(getattr(x, 'key', x)))
def test_tuning_key_is_result_neutral(t, v):
    from far_amd import _lib
    lib = _lib.load()
    _base()
    assert lib.far_set_tuning(t.key, v) == 0
    try:
        got = _step()
    finally:
        lib.far_set_tuning(t.key, t.default)
    _hold(got, t.neutral, f'tuning {t.key}={v}')

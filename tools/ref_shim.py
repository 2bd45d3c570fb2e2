"""Import shim for running the reference (crockwell/far, mp3d_loftr) on CPU in THIS container.

Container-only tooling: used by tools/make_goldens.py to produce tests/golden/*.npz.
Nothing under tools/ is imported by the product (far_amd/) or shipped to the GPU box's
test/bench path.  The reference tree never travels; only the generated vectors do.

What is stubbed (absent packages):
  loguru, yacs, cv2, kornia, pytorch_lightning.utilities.rank_zero_only
Non-reference arithmetic supplied by this shim (kornia 0.7.1 is not installed; these are
restatements of its *published* definitions, so anything that depends on them is
"parity unpinned" against kornia itself and says so in the fixture metadata):
  kornia.utils.grid.create_meshgrid
  kornia.geometry.subpix.dsnt.spatial_expectation2d
  kornia.geometry.epipolar.{sampson_epipolar_distance, symmetrical_epipolar_distance,
                            essential_from_Rt}
  kornia.geometry.conversions.convert_points_to_homogeneous
  kornia.geometry.solvers.{multiply_deg_one_poly, multiply_deg_two_one_poly}   (the monomial bookkeeping of Nister's five-point
      solver; kornia.geometry.solvers.determinant_to_polynomial is bound to the reference's OWN copy of that function,
      third_party/prior_ransac/cv_geometry.py:23-551, by bind_reference_solvers() -- not restated)
"""
import sys
import types

import torch

REF_ROOT = '/root/reference/mp3d_loftr'


class _AttrDict(dict):
    """Minimal stand-in for yacs CfgNode: attribute access, clone, merge no-ops."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        import copy
        return copy.deepcopy(self)


def _mk(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def _create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=None):
    # kornia 0.7.1 kornia/utils/grid.py: x fastest, linspace(-1, 1) when normalized,
    # shape (1, H, W, 2) with [..., 0] = x, [..., 1] = y
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    if normalized_coordinates:
        xs = (xs / (width - 1) - 0.5) * 2
        ys = (ys / (height - 1) - 0.5) * 2
    base_grid = torch.stack(torch.meshgrid([xs, ys], indexing="ij"), dim=-1)  # WxHx2
    return base_grid.permute(1, 0, 2).unsqueeze(0)


def _spatial_expectation2d(input, normalized_coordinates=True):
    # kornia 0.7.1 kornia/geometry/subpix/dsnt.py
    batch_size, channels, height, width = input.shape
    grid = _create_meshgrid(height, width, normalized_coordinates, input.device).to(input.dtype)
    pos_x = grid[..., 0].reshape(-1)
    pos_y = grid[..., 1].reshape(-1)
    input_flat = input.view(batch_size, channels, -1)
    expected_y = torch.sum(pos_y * input_flat, -1, keepdim=True)
    expected_x = torch.sum(pos_x * input_flat, -1, keepdim=True)
    output = torch.cat([expected_x, expected_y], -1)
    return output.view(batch_size, channels, 2)


def _to_h(points):
    return torch.nn.functional.pad(points, [0, 1], "constant", 1.0)


def _sampson_epipolar_distance(pts1, pts2, Fm, squared=True, eps=1e-8):
    # kornia 0.7.1 kornia/geometry/epipolar/_metrics.py
    if pts1.shape[-1] == 2:
        pts1 = _to_h(pts1)
    if pts2.shape[-1] == 2:
        pts2 = _to_h(pts2)
    F_t = Fm.transpose(dim0=-2, dim1=-1)
    line1_in_2 = pts1 @ F_t
    line2_in_1 = pts2 @ Fm
    numerator = (pts2 * line1_in_2).sum(dim=-1).pow(2)
    denominator = line1_in_2[..., :2].norm(2, dim=-1).pow(2) + line2_in_1[..., :2].norm(2, dim=-1).pow(2)
    out = numerator / denominator
    if squared:
        return out
    return (out + eps).sqrt()


def _symmetrical_epipolar_distance(pts1, pts2, Fm, squared=True, eps=1e-8):
    if pts1.shape[-1] == 2:
        pts1 = _to_h(pts1)
    if pts2.shape[-1] == 2:
        pts2 = _to_h(pts2)
    F_t = Fm.transpose(dim0=-2, dim1=-1)
    line1_in_2 = pts1 @ F_t
    line2_in_1 = pts2 @ Fm
    numerator = (pts2 * line1_in_2).sum(dim=-1).pow(2)
    denominator_inv = 1.0 / (line1_in_2[..., :2].norm(2, dim=-1).pow(2)) + 1.0 / (
        line2_in_1[..., :2].norm(2, dim=-1).pow(2))
    out = numerator * denominator_inv
    if squared:
        return out
    return (out + eps).sqrt()


def _cross_product_matrix(x):
    x0, x1, x2 = x[..., 0], x[..., 1], x[..., 2]
    z = torch.zeros_like(x0)
    return torch.stack([z, -x2, x1, x2, z, -x0, -x1, x0, z], dim=-1).view(*x.shape[:-1], 3, 3)


def _essential_from_Rt(R1, t1, R2, t2):
    # kornia: relative_camera_motion then [t]x R
    R = R2 @ R1.transpose(-2, -1)
    t = t2 - R @ t1
    return _cross_product_matrix(t[..., 0]) @ R


def _multiply_deg_one_poly(a, b):
    # kornia 0.7.1 kornia/geometry/solvers/polynomial_solver.py: two linear polynomials in (x, y, z, 1) ->
    # the quadratic in (x^2, xy, xz, x, y^2, yz, y, z^2, z, 1)
    return torch.stack([
        a[:, 0] * b[:, 0],
        a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0],
        a[:, 0] * b[:, 2] + a[:, 2] * b[:, 0],
        a[:, 0] * b[:, 3] + a[:, 3] * b[:, 0],
        a[:, 1] * b[:, 1],
        a[:, 1] * b[:, 2] + a[:, 2] * b[:, 1],
        a[:, 1] * b[:, 3] + a[:, 3] * b[:, 1],
        a[:, 2] * b[:, 2],
        a[:, 2] * b[:, 3] + a[:, 3] * b[:, 2],
        a[:, 3] * b[:, 3],
    ], dim=-1)


def _multiply_deg_two_one_poly(a, b):
    # kornia 0.7.1, same file: a quadratic (order above) times a linear polynomial -> the cubic in Nister's order
    # (x^3, y^3, x^2 y, x y^2, x^2 z, x^2, y^2 z, y^2, xyz, xy, x z^2, xz, x, y z^2, yz, y, z^3, z^2, z, 1)
    return torch.stack([
        a[:, 0] * b[:, 0],
        a[:, 4] * b[:, 1],
        a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0],
        a[:, 1] * b[:, 1] + a[:, 4] * b[:, 0],
        a[:, 0] * b[:, 2] + a[:, 2] * b[:, 0],
        a[:, 0] * b[:, 3] + a[:, 3] * b[:, 0],
        a[:, 4] * b[:, 2] + a[:, 5] * b[:, 1],
        a[:, 4] * b[:, 3] + a[:, 6] * b[:, 1],
        a[:, 1] * b[:, 2] + a[:, 2] * b[:, 1] + a[:, 5] * b[:, 0],
        a[:, 1] * b[:, 3] + a[:, 3] * b[:, 1] + a[:, 6] * b[:, 0],
        a[:, 2] * b[:, 2] + a[:, 7] * b[:, 0],
        a[:, 2] * b[:, 3] + a[:, 3] * b[:, 2] + a[:, 8] * b[:, 0],
        a[:, 3] * b[:, 3] + a[:, 9] * b[:, 0],
        a[:, 5] * b[:, 2] + a[:, 7] * b[:, 1],
        a[:, 5] * b[:, 3] + a[:, 6] * b[:, 2] + a[:, 8] * b[:, 1],
        a[:, 6] * b[:, 3] + a[:, 9] * b[:, 1],
        a[:, 7] * b[:, 2],
        a[:, 7] * b[:, 3] + a[:, 8] * b[:, 2],
        a[:, 8] * b[:, 3] + a[:, 9] * b[:, 2],
        a[:, 9] * b[:, 3],
    ], dim=-1)


def bind_reference_solvers():
    """kornia.geometry.solvers.determinant_to_polynomial := the reference's own copy (cv_geometry.py:23-551; its call site
    :977 uses the kornia name, the local one is commented out at :978).  Call after install(); imports cv_geometry."""
    import cv_geometry
    sys.modules['kornia.geometry'].solvers.determinant_to_polynomial = cv_geometry.determinant_to_polynomial
    cv_geometry.solvers.determinant_to_polynomial = cv_geometry.determinant_to_polynomial
    return cv_geometry


_installed = False


def install():
    """Install the stubs and put the reference on sys.path. Idempotent."""
    global _installed
    if _installed:
        return
    _installed = True

    # loguru
    lg = _mk('loguru')

    class _L:
        def __getattr__(self, k):
            return lambda *a, **k2: None
    lg.logger = _L()
    lg._Logger = _L

    # yacs
    _mk('yacs')
    yc = _mk('yacs.config')

    class CfgNode(_AttrDict):
        pass
    yc.CfgNode = CfgNode

    # cv2 (empty; any call raises AttributeError -> that branch is not runnable here)
    _mk('cv2')

    # pytorch_lightning bits used by src/utils/misc.py
    _mk('pytorch_lightning')
    plu = _mk('pytorch_lightning.utilities')

    def rank_zero_only(fn):
        return fn
    rank_zero_only.rank = 0
    plu.rank_zero_only = rank_zero_only

    # kornia
    k = _mk('kornia')
    kc = _mk('kornia.core')
    kc.Device = object
    kc.Module = torch.nn.Module
    kc.Tensor = torch.Tensor
    kc.zeros = torch.zeros
    kcc = _mk('kornia.core.check')
    kcc.KORNIA_CHECK_SHAPE = lambda *a, **k2: None
    kcc.KORNIA_CHECK = lambda *a, **k2: None
    kcc.KORNIA_CHECK_SAME_SHAPE = lambda *a, **k2: None
    kg = _mk('kornia.geometry')
    for n in ['find_fundamental', 'find_homography_dlt', 'find_homography_dlt_iterated',
              'find_homography_lines_dlt', 'find_homography_lines_dlt_iterated']:
        setattr(kg, n, None)
    kg.symmetrical_epipolar_distance = _symmetrical_epipolar_distance
    kg.solvers = types.SimpleNamespace(multiply_deg_one_poly=_multiply_deg_one_poly,
                                       multiply_deg_two_one_poly=_multiply_deg_two_one_poly)
    kge = _mk('kornia.geometry.epipolar')
    kge.sampson_epipolar_distance = _sampson_epipolar_distance
    kge.symmetrical_epipolar_distance = _symmetrical_epipolar_distance
    kge.essential_from_Rt = _essential_from_Rt
    kge.numeric = types.SimpleNamespace(cross_product_matrix=_cross_product_matrix)
    kg.epipolar = kge
    kgef = _mk('kornia.geometry.epipolar.fundamental')
    kgef.fundamental_from_essential = lambda E, K1, K2: torch.inverse(K2).transpose(-2, -1) @ E @ torch.inverse(K1)
    kgh = _mk('kornia.geometry.homography')
    for n in ['line_segment_transfer_error_one_way', 'oneway_transfer_error', 'sample_is_valid_for_homography']:
        setattr(kgh, n, None)
    kgc = _mk('kornia.geometry.conversions')
    kgc.convert_points_to_homogeneous = _to_h
    kgs = _mk('kornia.geometry.subpix')
    kgd = _mk('kornia.geometry.subpix.dsnt')
    kgd.spatial_expectation2d = _spatial_expectation2d
    kgs.dsnt = kgd
    ku = _mk('kornia.utils')
    kug = _mk('kornia.utils.grid')
    kug.create_meshgrid = _create_meshgrid
    ku.create_meshgrid = _create_meshgrid
    ku.grid = kug
    k.geometry = kg
    k.utils = ku
    k.core = kc

    # .cuda() -> identity (hard-coded .cuda() calls on the hot path)
    torch.Tensor.cuda = lambda self, *a, **k2: self

    sys.path.insert(0, REF_ROOT)
    sys.path.insert(0, REF_ROOT + '/third_party/prior_ransac')


def far_eval_config():
    """lower_config(cfg)['loftr'] for the FAR eval setting (demo.py:58-99 / eval_matterport.sh:27-37)."""
    install()
    from src.config.default import get_cfg_defaults
    from src.utils.misc import lower_config
    cfg = get_cfg_defaults()
    L = cfg.LOFTR
    L.PREDICT_TRANSLATION_SCALE = False
    L.REGRESS_RT = True
    L.REGRESS_LOFTR_LAYERS = 1
    L.REGRESS.USE_POS_EMBEDDING = True
    L.REGRESS.REGRESS_USE_NUM_CORRES = True
    L.COARSE.LAYER_NAMES = ['self', 'cross'] * 3
    L.FROM_SAVED_PREDS = None
    L.SOLVER = "prior_ransac"
    L.USE_MANY_RANSAC_THR = True
    L.FINE_PRED_STEPS = 2
    L.REGRESS.SAVE_MLP_FEATS = False
    L.REGRESS.USE_SIMPLE_MOE = True
    L.REGRESS.USE_2WT = True
    L.REGRESS.USE_5050_WEIGHT = False
    L.REGRESS.USE_1WT = False
    L.REGRESS.SCALE_8PT = True
    L.REGRESS.SAVE_GATING_WEIGHTS = False
    L.TRAINING = False
    return lower_config(cfg)['loftr']

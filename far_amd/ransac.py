"""The function-level API of the pose solver, on kernel K4's stages:

    RANSAC(model_type, inl_th, batch_size, max_iter, ...).forward(kp1, kp2)   third_party/prior_ransac/ransac.py:74-159, :340-442
    run_8point(points1, points2, weights)                                     third_party/prior_ransac/cv_geometry.py:772-833
    decompose_essential_matrix(E_mat)                                         third_party/prior_ransac/essential.py:99-139

Same names, argument meaning, shapes and error behaviour (AssertionError on malformed inputs, NotImplementedError on an unknown
model type) as the reference's, so that a caller importing them from `third_party.prior_ransac` can import them from here.  The
arithmetic runs in float64 on the GPU (far_ransac_f64 / far_eightpoint_f64 / far_decompose_essential_f64, include/far_hip.h);
results come back in the dtype of the inputs.  There is no CPU path: CPU tensors raise.

What RANSAC.forward covers is what K4 implements, i.e. the configurations estimate_pose constructs (metrics.py:100-153:
`essential_cv2`, max_iter = 1, max_lo_iters = 0, batch_size = 2048, with or without a prior): `essential*` model types, the squared
Sampson error (ransac.py:151-157), one batch of `batch_size * max_iter` models, NO local optimisation, the prior as biased sampling
(exp(-d / 0.1)) + the no-exp prior score (-err^2 / lambda).  Anything else raises NotImplementedError instead of silently computing
something different:
  * model_type = 'fundamental' -- the reference verifies it with symmetrical_epipolar_distance (ransac.py:141), K4 with Sampson:
    other inlier sets at the same inl_th.  The 8-point minimal solver itself is available as `essential` / `essential_cv2` with
    minimal = 8, and stand-alone as run_8point;
  * max_lo_iters > 0 -- the reference polishes the best model with find_fundamental on its inliers (:413-424); K4 does not.  The
    constructor default stays the reference's 5, so the argument must be passed as 0, as metrics.py does;
  * 'homography*' models, early stopping, the exp prior score, a prior without linear biased sampling.
prior_params['K1'] / ['K2'] are accepted and NOT used -- exactly as in the reference: fundamental_from_RT computes F from them and
then returns E (ransac.py:63-71), so the bias weights are the symmetric epipolar distance to the prior's ESSENTIAL matrix on the
(already K-normalised) points, which is what k_prepare evaluates.  Documented deviations: the sampling hash
replaces numpy / torch RNG draws; 'essential_cv2' (OpenCV's five-point on 6 points, not in this image) maps to the minimal solver
named by `minimal` (default 8 = the normalized 8-point on 8 samples; 5 = Nister's five-point).
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops

_P = ops._p


def _need_gpu(t, what):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise _lib.FarHipError(f'{what}: far_amd.ransac needs tensors on the GPU (no CPU fallback exists)')


def run_8point(points1, points2, weights=None):
    """cv_geometry.py:772-833.  points1, points2 (B, N, 2), N >= 8; weights (B, N) or None -> F (B, 3, 3)."""
    if points1.shape != points2.shape:
        raise AssertionError(points1.shape, points2.shape)
    if len(points1.shape) != 3 or points1.shape[-1] != 2:
        raise AssertionError(points1.shape)
    if points1.shape[1] < 8:
        raise AssertionError(points1.shape)
    if weights is not None and not (len(weights.shape) == 2 and weights.shape[1] == points1.shape[1]):
        raise AssertionError(weights.shape)
    _need_gpu(points1, 'run_8point')
    lib = _lib.load()
    B, N = int(points1.shape[0]), int(points1.shape[1])
    f64 = torch.float64
    p1 = points1.detach().to(f64).contiguous()
    p2 = points2.detach().to(f64).contiguous()
    w = None if weights is None else weights.detach().to(f64).contiguous()
    F = torch.empty(B, 3, 3, dtype=f64, device=points1.device)
    _lib.check(lib.far_eightpoint_f64(_P(p1), _P(p2), _P(w), B, N, _P(F), ops._stream()), 'far_eightpoint_f64')
    return F.to(points1.dtype)


def decompose_essential_matrix(E_mat):
    """essential.py:99-139.  E_mat (*, 3, 3) -> (R1 (*, 3, 3), R2 (*, 3, 3), T (*, 3, 1))."""
    if not (len(E_mat.shape) >= 2 and tuple(E_mat.shape[-2:]) == (3, 3)):
        raise AssertionError(E_mat.shape)
    _need_gpu(E_mat, 'decompose_essential_matrix')
    lib = _lib.load()
    lead = tuple(E_mat.shape[:-2])
    f64 = torch.float64
    E = E_mat.detach().to(f64).reshape(-1, 3, 3).contiguous()
    n = int(E.shape[0])
    R1 = torch.empty(n, 3, 3, dtype=f64, device=E.device)
    R2 = torch.empty(n, 3, 3, dtype=f64, device=E.device)
    t = torch.empty(n, 3, dtype=f64, device=E.device)
    _lib.check(lib.far_decompose_essential_f64(_P(E), n, _P(R1), _P(R2), _P(t), ops._stream()), 'far_decompose_essential_f64')
    dt = E_mat.dtype
    return R1.reshape(lead + (3, 3)).to(dt), R2.reshape(lead + (3, 3)).to(dt), t.reshape(lead + (3, 1)).to(dt)


class RANSAC(torch.nn.Module):
    """ransac.py:74-159 (constructor: same parameter names and defaults) and :340-442 (forward)."""

    _MODELS = {'essential': 5, 'essential_cv2': None}

    def __init__(self, model_type='homography', inl_th=2.0, batch_size=2048, max_iter=10, confidence=0.99, max_lo_iters=5,
                 prior_params={}, use_noexp_prior_scoring=False, use_linear_bias_sampling=False, bias_sigma_sq=1.0,
                 compute_stopping_inlier_only=False, perform_early_stopping=False, l1_dist=False, use_epipolar_error=False,
                 K=None, normalize=False, minimal=8, seed=0):
        super().__init__()
        self.supported_models = ['essential', 'essential_cv2']
        if model_type == 'fundamental':
            raise NotImplementedError("model_type='fundamental' is verified with the symmetric epipolar distance in the reference "
                                      "(ransac.py:141); kernel K4 scores with the squared Sampson distance.  Use 'essential' / 'essential_cv2' "
                                      "with minimal=8 for 8-point hypotheses, or run_8point for the solver alone.")
        if model_type not in self._MODELS:
            # the reference knows 'homography' / 'homography_from_linesegments' too; they are not on FAR's path
            raise NotImplementedError(f'{model_type} is unknown. Try one of {self.supported_models}')
        self.model_type, self.inl_th, self.batch_size, self.max_iter = model_type, float(inl_th), int(batch_size), int(max_iter)
        self.confidence, self.max_lo_iters = confidence, int(max_lo_iters)
        self.use_noexp_prior_scoring, self.use_linear_bias_sampling, self.bias_sigma_sq = use_noexp_prior_scoring, use_linear_bias_sampling, bias_sigma_sq
        self.compute_stopping_inlier_only, self.perform_early_stopping = compute_stopping_inlier_only, perform_early_stopping
        self.l1_dist, self.use_epipolar_error, self.K, self.normalize = l1_dist, use_epipolar_error, K, normalize
        self.minimal = self._MODELS[model_type] or int(minimal)
        self.minimal_sample_size = {'essential': 5, 'essential_cv2': 6}[model_type]
        self.seed = int(seed)
        self.prior_params = prior_params
        self.setup_prior(prior_params)
        unsupported = []
        if self.minimal not in (5, 8):
            unsupported.append(f'minimal={minimal}')
        if self.max_lo_iters > 0:
            unsupported.append(f'local optimisation (max_lo_iters={self.max_lo_iters}; pass max_lo_iters=0 as metrics.py:118 does)')
        if use_epipolar_error or l1_dist:
            unsupported.append('use_epipolar_error / l1_dist (K4 scores with the squared Sampson distance)')
        if perform_early_stopping or compute_stopping_inlier_only:
            unsupported.append('early stopping (K4 verifies one batch of batch_size * max_iter models)')
        if self.use_prior:
            if not use_noexp_prior_scoring:
                unsupported.append('the exp prior score (use_noexp_prior_scoring=False)')
            if not prior_params.get('biased_sampling') or not use_linear_bias_sampling or abs(float(bias_sigma_sq) - 0.1) > 1e-12:
                unsupported.append('a prior without linear biased sampling at bias_sigma_sq = 0.1')
            if not prior_params.get('rotation_pcl_error', True) or 'pcl' not in prior_params:
                unsupported.append('a prior without the point-cloud error (prior_params["pcl"])')
            if self.max_iter != 1:
                unsupported.append('a prior with max_iter != 1 (the reference alternates biased and unbiased batches)')
        if self.minimal == 5 and self.batch_size * self.max_iter < 10:
            unsupported.append('fewer than 10 models for the five-point solver')
        if unsupported:
            raise NotImplementedError('far_amd.ransac.RANSAC (kernel K4) does not implement: ' + '; '.join(unsupported))

    def setup_prior(self, prior_params):
        """ransac.py:177-186 (normalises the prior translation IN PLACE, as the reference does)."""
        if prior_params:
            self.use_prior = True
            self.prior_lambda = prior_params['lambda']
            RT = prior_params['RT']
            RT[:, 3] /= torch.linalg.norm(RT[:, 3])
        else:
            self.use_prior = False
            self.prior_lambda = 1.0

    def validate_inputs(self, kp1, kp2, weights=None):
        """ransac.py:310-338 for the point models."""
        if not (torch.is_tensor(kp1) and torch.is_tensor(kp2)):
            raise AssertionError('kp1 / kp2 must be tensors')
        if len(kp1.shape) != 2 or kp1.shape[-1] != 2 or len(kp2.shape) != 2 or kp2.shape[-1] != 2:
            raise AssertionError(kp1.shape, kp2.shape)
        if not (kp1.shape[0] == kp2.shape[0]) or (kp1.shape[0] < self.minimal_sample_size):
            raise ValueError(f'kp1 and kp2 should be equal shape at at least [{self.minimal_sample_size}, 2], got {kp1.shape}, {kp2.shape}')

    def forward(self, kp1, kp2, weights=None, samples=None):
        """kp1, kp2 (N, 2) -> (model (3, 3), inliers (N,) bool, inliers_best_tight (N,) bool, inliers_best_ultra_tight (N,) bool)
        -- the four values ransac.py:442 returns.  No model above the score floor: zeros(3, 3) and all-False masks (:354-355).
        samples: optional explicit minimal samples ((H, 8) or (H // 10, 5) int): the hook the parity tests use."""
        self.validate_inputs(kp1, kp2, weights)
        _need_gpu(kp1, 'RANSAC.forward')
        lib = _lib.load()
        dev = kp1.device
        N = int(kp1.shape[0])
        H = self.batch_size * self.max_iter
        a = kp1.detach().float().contiguous()
        b = kp2.detach().float().contiguous()
        offs = torch.tensor([0, N], dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
        th = torch.full((1,), self.inl_th, dtype=torch.float64, device=dev)
        prior = pcl = None
        P = 0
        if self.use_prior:
            prior = self.prior_params['RT'].detach().to(device=dev, dtype=torch.float32).reshape(1, 3, 4).contiguous()
            pcl = self.prior_params['pcl'].detach().to(device=dev, dtype=torch.float32).contiguous()
            P = int(pcl.shape[0])
        smp = None if samples is None else torch.as_tensor(samples).to(device=dev, dtype=torch.int32).contiguous()
        ws = ops._ws(lib.far_solver_workspace_bytes(1, N, H, P), dev)
        E = torch.empty(3, 3, dtype=torch.float64, device=dev)
        mask = torch.empty(N, dtype=torch.uint8, device=dev)
        cnt = torch.empty(4, dtype=torch.int32, device=dev)
        cp = cnt.data_ptr()
        rc = lib.far_ransac_f64(_P(a), _P(b), _P(offs), 1, N, N, _P(th), _P(prior), _P(pcl), P, float(self.prior_lambda), H, self.minimal,
                                self.seed & 0xffffffff, _P(smp), _P(E), _P(mask), ctypes.c_void_p(cp), ctypes.c_void_p(cp + 4),
                                ctypes.c_void_p(cp + 8), ctypes.c_void_p(cp + 12), _P(ws), ops._stream())
        _lib.check(rc, 'far_ransac_f64')
        self.last_best = cnt[3]                          # index of the winning model (-1: none), on the device
        return E.to(kp1.dtype), (mask & 1).bool(), (mask & 2).bool(), (mask & 4).bool()

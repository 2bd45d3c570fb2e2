"""CPU tests of host-side logic that carries no kernel: solver dispatch, config, fail-loud behaviour."""
import os

import numpy as np
import pytest
import torch

from far_amd import _lib, ops, solver
from far_amd.config import RunCfg, far_eval_config


def test_solver_branch_mirrors_reference_dispatch():
    # metrics.py:100, :130, :153
    assert solver._branch('prior_ransac', True) == 'prior'
    assert solver._branch('prior_ransac', False) == 'ransac'       # falls through to the cv2.RANSAC branch
    assert solver._branch('prior_ransac_noprior', False) == 'noprior'
    assert solver._branch('ransac', False) == 'ransac'


def test_prior_point_cloud_is_the_reference_draw():
    # np.random.seed(0) (supervision.py:207) then np.random.uniform(-3, 3, (300, 3)) (metrics.py:103)
    np.random.seed(0)
    ref = np.random.uniform(low=-3.0, high=3.0, size=(300, 3)).astype(np.float32)
    np.testing.assert_array_equal(solver.prior_point_cloud('cpu').numpy(), ref)


def test_fewer_than_five_keypoints_returns_none():
    k = torch.zeros(4, 2)
    assert solver.estimate_pose(k, k, torch.eye(3), torch.eye(3), 0.5) == (None, 0, 0, 0)     # metrics.py:83-85


def test_ops_fail_loudly_without_gpu_tensors():
    x = torch.zeros(1, 32, 32)
    with pytest.raises(_lib.FarHipError):
        ops.dual_softmax_stats(x, x)
    with pytest.raises(_lib.FarHipError):
        ops.linear_attention(x, x, x, 1)


def test_config_matches_far_eval_setting():
    c = far_eval_config()
    assert c['coarse']['layer_names'] == ['self', 'cross'] * 3 and c['solver'] == 'prior_ransac'
    assert c['match_coarse']['thr'] == 0.2 and c['match_coarse']['border_rm'] == 2
    r = RunCfg()
    assert r.TRAINER.RANSAC_PIXEL_THR == 0.5 and r.LOFTR.SOLVER == 'prior_ransac'


def test_inference_on_cpu_raises_instead_of_falling_back():
    # no grad, eval mode, CPU tensors: the kernel path is the only path and it needs the GPU
    from far_amd.loftr import CoarseMatching
    m = CoarseMatching(far_eval_config()['match_coarse']).eval()
    with torch.no_grad(), pytest.raises(_lib.FarHipError):
        m(torch.zeros(1, 4, 32), torch.zeros(1, 4, 32), {'hw0_c': (2, 2), 'hw1_c': (2, 2), 'hw0_i': (16, 16)})


def test_cached_prediction_format_roundtrip(tmp_path):
    from far_amd import cache_io
    g = torch.Generator().manual_seed(0)
    B = 3
    data = {'loftr_rt': torch.randn(B, 3, 4, generator=g, dtype=torch.float64), 'num_correspondences': torch.tensor([5, 700, 0]),
            'featmap0': torch.randn(B, 4800, 256, generator=g), 'featmap1': torch.randn(B, 4800, 256, generator=g)}
    cache_io.save_batch(str(tmp_path), 'test', [7, 8, 11], data)
    # the reference's reader side: one .pt per quantity per pair id (interiornet_streetlearn.py:108-118)
    one = torch.load(str(tmp_path / 'test' / 'coarse_features' / '8.pt'))
    assert one.shape == (2, 4800, 256) and torch.equal(one[1], data['featmap1'][1])
    assert torch.load(str(tmp_path / 'test' / 'loftr_preds' / '11.pt')).shape == (3, 4)
    back = cache_io.load_batch(str(tmp_path), 'test', [11, 7])
    assert torch.equal(back['loftr_rt'], data['loftr_rt'][[2, 0]]) and back['num_correspondences'].tolist() == [0, 5]
    assert torch.equal(back['featmap0'], data['featmap0'][[2, 0]]) and back['inliers_best_tight'].tolist() == [0, 0]


# ---------------------------------------------------------------------------------------------------------------------
# property tests (hypothesis) of the host-side logic that no golden vector exercises exhaustively
# ---------------------------------------------------------------------------------------------------------------------
from hypothesis import given, settings, strategies as st   # noqa: E402


@settings(max_examples=200, deadline=None)
@given(n=st.integers(0, 500), world=st.integers(1, 16))
def test_shard_indices_partition_the_pairs(n, world):
    """Every pair goes to exactly one rank, in rank::world order (DistributedSampler(shuffle=False), data.py:115-117)."""
    from far_amd.parallel import shard_indices
    shards = [list(shard_indices(n, r, world)) for r in range(world)]
    assert sorted(i for s in shards for i in s) == list(range(n))
    for r, s in enumerate(shards):
        assert s == list(range(r, n, world))
        assert abs(len(s) - n / world) < 1                           # balanced to within one pair


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 10_000), m=st.integers(1, 60), shift=st.floats(0.0, 3.0))
def test_fine_loss_is_a_function_of_the_match_set(seed, m, shift):
    """loftr_loss.py:151-183: permuting the matches leaves the loss unchanged; it is never negative; the std column only
    re-weights (weights sum to the number of matches); eval mode without a correct match returns None."""
    from far_amd.losses import fine_loss_l2_std
    g = torch.Generator().manual_seed(seed)
    ef = torch.cat([torch.randn(m, 2, generator=g), torch.rand(m, 1, generator=g) + 0.05], 1)
    gt = torch.randn(m, 2, generator=g) * 0.6 + shift
    perm = torch.randperm(m, generator=g)
    a = fine_loss_l2_std(ef, gt, 1.0, True)
    b = fine_loss_l2_std(ef[perm], gt[perm], 1.0, True)
    assert float(a) >= 0 and abs(float(a) - float(b)) <= 1e-5 * max(1.0, float(a))
    if not bool((gt.abs().amax(1) < 1.0).any()):
        assert fine_loss_l2_std(ef, gt, 1.0, False) is None and float(a) == 0.0     # the dummy entry carries weight 0


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 10_000), b=st.integers(1, 5), scramble=st.booleans())
def test_spvs_rt_segments_are_order_independent_on_the_host(seed, b, scramble):
    """The host-side half of spvs_RT's batching (the solver itself is a GPU kernel): counts from m_bids and the stable
    sort that makes per-pair segments contiguous reproduce the reference's `mask = m_bids == bs` selection
    (supervision.py:209-210) for any order of the matches."""
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, 9, (b,), generator=g)
    bids = torch.repeat_interleave(torch.arange(b), counts)
    pts = torch.arange(len(bids), dtype=torch.float32)[:, None].repeat(1, 2)
    if scramble and len(bids):
        p = torch.randperm(len(bids), generator=g)
        bids, pts = bids[p], pts[p]
    got_counts = torch.bincount(bids, minlength=b)
    assert torch.equal(got_counts, counts)
    order = torch.sort(bids, stable=True)[1]
    seg = pts[order]
    offs = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(got_counts, 0)])
    for bs in range(b):
        ref = pts[bids == bs]                                      # the reference's selection, in the caller's order
        assert torch.equal(seg[offs[bs]:offs[bs + 1]], ref)
    back = torch.empty_like(order)
    back[order] = torch.arange(len(order))
    assert torch.equal(seg[back], pts)                             # and the mask scatter-back is the inverse permutation


def test_trainval_inference_merges_a_copied_dict_and_checks_the_supervision_contract():
    """pipeline._trainval_inference with a stand-in matcher: (1) a forward wrapper that hands the module a COPY of the
    batch (what DistributedDataParallel(device_ids=[...]) does to dict inputs) -- the returned dict is merged back into the
    caller's; (2) a depth-supervised source without depth or labels is refused up front; (3) interiornet_streetlearn
    skips both supervision calls (lightning_loftr.py:131-140)."""
    import pytest
    import torch
    from far_amd.config import RunCfg
    from far_amd.pipeline import _trainval_inference

    class Matcher:
        config = {'regress_rt': False}

        def __call__(self, data, train=False):
            data.update(b_ids=torch.tensor([0, 0]), i_ids=torch.tensor([1, 2]), j_ids=torch.tensor([3, 4]),
                        mkpts0_f=torch.zeros(2, 2), seen_train=train)
            return data
    m = Matcher()
    seen = {}
    loss_fn = lambda d: seen.update(keys=set(d))
    labels = dict(spv_b_ids=torch.zeros(1, dtype=torch.long), spv_i_ids=torch.zeros(1, dtype=torch.long),
                  spv_j_ids=torch.zeros(1, dtype=torch.long), spv_w_pt0_i=torch.zeros(1, 8, 2), spv_pt1_i=torch.ones(1, 8, 2))
    batch = dict(labels, dataset_name=['mp3d'])
    copying = lambda d, train=False: m(dict(d), train=train)
    _trainval_inference(m, batch, loss_fn, RunCfg(), True, 16, 0, forward=copying)
    assert batch['seen_train'] is True and 'b_ids' in batch and batch['expec_f_gt'].shape == (2, 2)
    assert 'expec_f_gt' in seen['keys']
    with pytest.raises(KeyError, match='depth-supervised'):
        _trainval_inference(m, {'dataset_name': ['mp3d'], 'spv_b_ids': labels['spv_b_ids']}, loss_fn, RunCfg(), False, 16, 0)
    b3 = {'dataset_name': ['interiornet_streetlearn']}
    _trainval_inference(m, b3, loss_fn, RunCfg(), False, 16, 0)
    assert 'expec_f_gt' not in b3 and b3['seen_train'] is False
    silent = lambda d, train=False: None if m(dict(d), train=train) else None       # a wrapper that swallows the dict
    with pytest.raises(AssertionError, match='different dict'):
        _trainval_inference(m, dict(labels, dataset_name=['mp3d']), loss_fn, RunCfg(), True, 16, 0, forward=silent)


def test_kv_interleaved_weight_layout_is_what_far_linear_kv_expects():
    """ops.kv_interleaved_weight: rows 64 j + [0, 32) = Wk's rows of head j, 64 j + [32, 64) = Wv's (include/far_hip.h,
    far_linear_kv_f16s).  With it, the LinearAttention state (linear_attention.py:38-45) read head by head from the fused
    projection's columns is the one computed from separate k / v projections (oracle/attention.py)."""
    import numpy as np
    import torch
    from far_amd import ops
    from oracle import attention as oa
    rng = np.random.default_rng(3)
    H, D, K, S = 8, 32, 24, 40
    wk, wv = (torch.from_numpy(rng.standard_normal((H * D, K))) for _ in range(2))
    w = ops.kv_interleaved_weight(wk, wv, H)
    assert w.shape == (2 * H * D, K)
    for j in range(H):
        assert torch.equal(w[64 * j:64 * j + 32], wk[32 * j:32 * j + 32]) and torch.equal(w[64 * j + 32:64 * j + 64], wv[32 * j:32 * j + 32])
    x = rng.standard_normal((1, S, K))
    kv = x @ w.numpy().T                                                  # the fused launch's columns
    kf = np.where(kv > 0, kv, np.expm1(kv)) + 1
    q = rng.standard_normal((1, 5, H * D))
    state = np.stack([kf[0, :, 64 * j:64 * j + 32].T @ (kv[0, :, 64 * j + 32:64 * j + 64] / S) for j in range(H)])     # [H][d][v]
    ksum = np.stack([kf[0, :, 64 * j:64 * j + 32].sum(0) for j in range(H)])
    qf = (np.where(q > 0, q, np.expm1(q)) + 1).reshape(5, H, D)
    out = np.einsum('lhd,hdv->lhv', qf, state) / (np.einsum('lhd,hd->lh', qf, ksum)[..., None] + 1e-6) * S
    ref = oa.linear_attention(q, x @ wk.numpy().T, x @ wv.numpy().T, H, dtype=np.float64)
    np.testing.assert_allclose(out.reshape(1, 5, H * D), ref, rtol=1e-10, atol=1e-12)
    import pytest
    from far_amd._lib import FarHipError
    with pytest.raises(FarHipError):
        ops.kv_interleaved_weight(wk, wv[:64], H)


def test_demo_without_a_gpu_exits_2_and_computes_nothing():
    """BASELINE configs[0] says 'demo on CPU'; the product has no CPU path, so demo.py says so and exits 2 (tests/test_demo_gpu.py
    runs it for real on the GPU box)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('a GPU is visible')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'demo.py'), '--synthetic'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'no GPU visible' in r.stderr and 'predicted pose' not in r.stdout


def test_ransac_api_argument_checks_need_no_gpu():
    """far_amd/ransac.py mirrors the reference's error behaviour (cv_geometry.py:786-794, essential.py:110-111, ransac.py:158-159,
    :310-338) before anything touches the library, and refuses CPU tensors loudly (no CPU fallback exists)."""
    from far_amd.ransac import RANSAC, decompose_essential_matrix, run_8point
    p = torch.zeros(2, 9, 2)
    with pytest.raises(AssertionError):
        run_8point(p, torch.zeros(2, 8, 2))
    with pytest.raises(AssertionError):
        run_8point(p[:, :7], p[:, :7])
    with pytest.raises(AssertionError):
        run_8point(p, p, torch.ones(2, 5))
    with pytest.raises(AssertionError):
        decompose_essential_matrix(torch.zeros(4, 3, 2))
    with pytest.raises(_lib.FarHipError, match='no CPU fallback'):
        run_8point(p, p)
    with pytest.raises(_lib.FarHipError, match='no CPU fallback'):
        decompose_essential_matrix(torch.eye(3)[None])
    with pytest.raises(NotImplementedError):
        RANSAC(model_type='homography')
    with pytest.raises(NotImplementedError, match='early stopping'):
        RANSAC(model_type='essential', perform_early_stopping=True, max_lo_iters=0)
    with pytest.raises(NotImplementedError, match='symmetric epipolar'):      # the reference verifies it with another error function
        RANSAC(model_type='fundamental', max_lo_iters=0)
    with pytest.raises(NotImplementedError, match='local optimisation'):      # the reference's default of 5 polishing rounds is not silently dropped
        RANSAC(model_type='essential')
    with pytest.raises(NotImplementedError, match='exp prior score'):
        RANSAC(model_type='essential_cv2', prior_params={'RT': torch.eye(3, 4), 'pcl': torch.zeros(3, 3), 'lambda': 0.3, 'biased_sampling': 'biased'},
               use_linear_bias_sampling=True, bias_sigma_sq=0.1, max_iter=1, max_lo_iters=0)
    prior = {'RT': torch.tensor([[1., 0, 0, 2.0], [0, 1, 0, 0], [0, 0, 1, 0]]), 'pcl': torch.zeros(3, 3), 'lambda': 0.3, 'biased_sampling': 'biased'}
    m = RANSAC(model_type='essential_cv2', prior_params=prior, use_noexp_prior_scoring=True, use_linear_bias_sampling=True, bias_sigma_sq=0.1,
               max_iter=1, max_lo_iters=0, inl_th=3e-7)
    assert m.use_prior and m.minimal == 8 and m.minimal_sample_size == 6
    assert abs(float(torch.linalg.norm(prior['RT'][:, 3])) - 1.0) < 1e-6            # ransac.py:183 normalises the caller's tensor in place
    with pytest.raises(ValueError):
        m.forward(torch.zeros(5, 2), torch.zeros(5, 2))
    with pytest.raises(_lib.FarHipError, match='no CPU fallback'):
        m.forward(torch.zeros(20, 2), torch.zeros(20, 2))


def test_build_id_ties_the_library_to_its_sources(tmp_path, monkeypatch):
    """far_amd/_lib.py refuses a library whose far_build_id() differs from the id of far_amd/csrc next to it (the .so travels
    outside git); FAR_HIP_LIB (another build, on purpose) skips the check."""
    from far_amd import build as fb
    lib = _lib.load()
    assert lib.far_build_id().decode() == fb.source_id() and len(fb.source_id()) == 16
    monkeypatch.setattr(fb, 'source_id', lambda: '0' * 16)
    monkeypatch.setattr(_lib, '_lib', None)
    with pytest.raises(_lib.FarHipError, match='built from other sources'):
        _lib.load()
    monkeypatch.setenv('FAR_HIP_LIB', _lib.LIB_PATH)
    monkeypatch.setattr(_lib, '_lib', None)
    assert _lib.load().far_abi_version() == _lib.EXPECTED_ABI
    monkeypatch.setattr(_lib, '_lib', lib)


def test_build_asm_scan_catches_a_staged_register_touched_before_its_wait(tmp_path):
    """far_amd/build.py scans K9's generated code for the property its asm pixel loads rely on; the scan itself on hand-made code:
    a clean sequence, a copy of the staged register above the counted wait, and a wait whose count does not cover the load."""
    from far_amd import build
    head = '_ZN1x6k_convILi1EEEvv:\n'
    load = '\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v[2:3], off\n\t;;#ASMEND\n'
    dma = '\tglobal_load_lds_dwordx4 v[4:5], off\n'
    wait = lambda k: f'\t;;#ASMSTART\n\ts_waitcnt vmcnt({k})\n\t;;#ASMEND\n'
    use = '\tv_cvt_pk_f16_f32 v20, v10, v11\n'
    tail = '\ts_endpgm\n'
    cases = {'clean': (head + load + dma * 4 + '\tv_add_f32 v30, v31, v32\n' + wait(4) + use + tail, 0),
             'copy above the wait': (head + load + dma * 4 + '\tv_mov_b32 v40, v12\n' + wait(4) + use + tail, 1),
             'wait that does not cover the load': (head + load + dma * 2 + wait(4) + use + tail, 1)}
    for name, (text, nbad) in cases.items():
        p = tmp_path / 'k.s'
        p.write_text(text)
        nfn, nld, bad = build.asm_check(str(p), 'k_conv')
        assert (nfn, nld, len(bad)) == (1, 1, nbad), (name, bad)


def test_build_scan_flags_a_half_register_write_read_too_soon(tmp_path):
    """far_amd/build.py: partial_write_check -- the gate behind round 6's second root cause (common.h: split2; docs/rounds/r06.md 2g).
    On hand-made code: `mixhi, one instruction, mfma` (what made K9's k | v-state epilogue differ from launch to launch) is flagged, also
    when it sits in a cold block BEHIND the function's s_endpgm (where hipcc had put that epilogue); the asm statement with its own
    s_nop 1 is clean; an LDS store may read the register at once (K17's transform); a half-register write hipcc emitted itself (no asm
    markers) is the compiler's business."""
    from far_amd import build
    head = '_ZN1x6k_convILi1EEEvv:\n'
    mix = lambda tail='': ('\t;;#ASMSTART\n\tv_fma_mixlo_f16 v39, v35, -1.0, v89 op_sel_hi:[1,0,0]\n\t;;#ASMEND\n'
                           '\t;;#ASMSTART\n\tv_fma_mixhi_f16 v39, v35, -1.0, v1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n' + tail + '\t;;#ASMEND\n')
    mfma = '\tv_mfma_f32_32x32x16_f16 v[66:81], v[38:41], v[50:53], v[66:81]\n'
    end = '\ts_endpgm\n'
    fend = '.Lfunc_end0:\n'
    cases = {
        'one instruction in between': (head + mix() + '\ts_nop 0\n' + mfma + end + fend, 1),
        'the same in a cold block behind s_endpgm': (head + end + '.LBB0_7:\n' + mix() + '\ts_nop 0\n' + mfma + '\ts_branch .LBB0_1\n' + fend, 1),
        'the next issue slot': (head + mix() + mfma + end + fend, 1),
        'the asm carries s_nop 1': (head + mix('\ts_nop 1\n') + mfma + end + fend, 0),
        'two other instructions in between': (head + mix() + '\tv_add_u32_e32 v1, v2, v3\n\tv_mul_f32_e32 v4, v5, v6\n' + mfma + end + fend, 0),
        'an LDS store reads it at once': (head + mix() + '\tds_write_b128 v138, v[38:41] offset:10240\n' + end + fend, 0),
        'not from asm': (head + '\tv_fma_mixhi_f16 v39, v35, -1.0, v1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n' + mfma + end + fend, 0),
    }
    for name, (text, nbad) in cases.items():
        p = tmp_path / 'h.s'
        p.write_text(text)
        n, bad = build.partial_write_check(str(p))
        assert len(bad) == nbad and n == (0 if name == 'not from asm' else 1), (name, n, bad)


def test_build_ring_scan_flags_a_barrier_crossed_with_lds_reads_outstanding(tmp_path):
    """far_amd/build.py: lds_ring_check -- the build gate behind the round-6 root cause (common.h: ring_barrier).  On hand-made code:
    the rounds-3..5 shape of K13 / K14 (two fragment reads, vmcnt wait, barrier, THEN the lgkmcnt wait hipcc sank below it) is flagged;
    the same with s_waitcnt lgkmcnt(0) in front of the barrier is clean; a partial wait that leaves a read in flight is flagged; a
    kernel without LDS-DMA is not this scan's business; the staged-store ADVICE case of _asm_scan (a store that reads a staged register
    before its wait) is a violation too."""
    from far_amd import build
    head = '_ZN1x5k_ringILi4EEEvv:\n'
    dma = '\tglobal_load_lds_dwordx4 v[4:5], off\n'
    reads = '\tds_read_b128 v[6:9], v137 offset:15360\n\tds_read_b128 v[2:5], v137 offset:14336\n'
    mfma = '\tv_mfma_f32_32x32x16_f16 v[66:81], v[178:181], v[2:5], v[66:81]\n'
    tail = '\ts_endpgm\n'
    cases = {
        'rounds 3-5: the wait sunk below the barrier': (head + dma + reads + '\ts_waitcnt vmcnt(8)\n\ts_barrier\n' + mfma + '\ts_waitcnt lgkmcnt(0)\n' + mfma + dma + tail, 1),
        'round 6: lgkmcnt(0) + barrier': (head + dma + reads + '\ts_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n' + mfma + dma + tail, 0),
        'combined wait': (head + dma + reads + '\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n' + dma + tail, 0),
        'partial wait leaves one read in flight': (head + dma + reads + '\ts_waitcnt lgkmcnt(1)\n\ts_barrier\n' + dma + tail, 1),
        'no LDS-DMA in the kernel': (head + reads + '\ts_barrier\n' + tail, 0),
    }
    for name, (text, nbad) in cases.items():
        p = tmp_path / 'r.s'
        p.write_text(text)
        nfn, nbar, bad = build.lds_ring_check(str(p))
        assert len(bad) == nbad, (name, bad)
        assert nbar == (0 if 'no LDS-DMA' in name else 1), name
    # _asm_scan: a store that READS a staged register in front of the counted wait is a violation, not "a younger request"
    k9 = ('_ZN1x6k_convILi1EEEvv:\n\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v[2:3], off\n\t;;#ASMEND\n' + dma * 2 +
          '\tscratch_store_dword off, v11, off offset:16\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(3)\n\t;;#ASMEND\n\tv_cvt_pk_f16_f32 v20, v10, v11\n' + tail)
    p = tmp_path / 'k.s'
    p.write_text(k9)
    nfn, nld, bad = build.asm_check(str(p), 'k_conv')
    assert (nfn, nld) == (1, 1) and len(bad) == 1 and 'scratch_store' in bad[0], bad


def test_package_raises_on_cpu_tensors_without_the_test_side_helper():
    """far_amd has no CPU / eager / vendor path of its own (far_amd/_vendor.py): the torch compositions live in tests/vendor_ops.py and
    are installed by conftest.py.  With the helper uninstalled -- the product's state -- CPU tensors raise FarHipError."""
    import pytest
    import torch
    from far_amd import _vendor
    from far_amd._lib import FarHipError
    from far_amd.loftr.transformer import LoFTREncoderLayer, Mlp
    keep = _vendor._impl
    _vendor.install(None)
    try:
        x = torch.zeros(1, 4, 256)
        with pytest.raises(FarHipError, match='tests/vendor_ops.py'):
            LoFTREncoderLayer(256, 8)(x, x)
        with pytest.raises(FarHipError):
            Mlp(256, 512)(x)
    finally:
        _vendor.install(keep)
    assert _vendor.installed()


def test_no_vendor_execution_inside_the_package():
    """VERDICT r5 item 4's criterion: no autocast region, no functional convolution / linear / softmax / unfold call under far_amd/."""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'far_amd')
    bad = []
    for base, _, files in os.walk(root):
        for f in files:
            if f.endswith('.py'):
                for i, ln in enumerate(open(os.path.join(base, f)), 1):
                    if re.search(r'autocast|F\.conv2d|nn\.functional\.(linear|softmax|unfold)', ln):
                        bad.append(f'{f}:{i}: {ln.strip()}')
    assert not bad, bad
    assert not os.path.exists(os.path.join(root, 'autograd_ops.py'))

"""Training path (BASELINE configs[2]) on CPU: the differentiable vendor-op path of far_amd vs golden G10, which
tools/make_goldens.py produced by running the REFERENCE's training-mode forward + backward.  Also a 2-rank gloo DDP
step (gradient all-reduce = the one exchange step of the path).  On CPU tensors the modules run the differentiable
vendor-op forms (tests/vendor_ops.py); on the GPU the same step runs K1 / K5 / K9-linear / K2 forward and backward
kernels and is held to the same golden (tests/test_pipeline_gpu.py::test_training_step_on_gpu)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from far_amd import synth
from far_amd.config import far_eval_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
G = os.path.join(ROOT, 'tests', 'golden')


def _train_helpers():
    """train_inputs / train_step / GRAD_KEYS: the builders the golden generator used (tests/util.py)."""
    from tests import util
    return util


@pytest.mark.timeout(900)
def test_training_forward_backward_matches_reference():
    from far_amd.loftr import LoFTR
    h = _train_helpers()
    g = np.load(os.path.join(G, 'g10_training.npz'))
    m = LoFTR(far_eval_config())
    synth.load_synthetic(m, seed=0)
    im0, im1, ii, jj, rt = h.train_inputs()
    data, losses = h.train_step(m, im0, im1, ii, jj, rt)
    # the training-time sampling (two randint draws) reproduces the reference's choice under the same seed
    for k in ['b_ids', 'i_ids', 'j_ids']:
        np.testing.assert_array_equal(data[k].numpy(), g[k])
    assert len(data['mconf']) == int(g['n_mconf'])
    np.testing.assert_allclose([l.item() for l in losses], g['losses'], rtol=2e-4)
    np.testing.assert_allclose(data['expec_f'][:64].detach().numpy(), g['expec_f_head'], atol=2e-4)
    np.testing.assert_allclose(data['regressed_rt'].detach().numpy(), g['regressed_rt'],
                               atol=1e-3 * np.abs(g['regressed_rt']).max(), rtol=1e-3)
    P = dict(m.named_parameters())
    for n, k in enumerate(h.GRAD_KEYS):
        gr = P[k].grad
        assert gr is not None, k
        np.testing.assert_allclose(gr.norm().item(), g['grad_norms'][n], rtol=5e-3, err_msg=k)
        s = gr.reshape(-1)[:: max(1, gr.numel() // 8)][:8].numpy()
        np.testing.assert_allclose(s, g['grad_samples'][n], rtol=2e-2, atol=2e-3 * np.abs(g['grad_samples'][n]).max() + 1e-7, err_msg=k)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from far_amd import _vendor
    from far_amd.loftr import LoFTR
    from tests import vendor_ops
    _vendor.install(vendor_ops)                        # a spawned rank does not run conftest.py: the CPU compositions are test infrastructure
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        cfg = far_eval_config()
        cfg['regress_rt'] = False                      # matcher only: keeps the CPU test small
        cfg['coarse']['layer_names'] = ['self', 'cross']
        m = LoFTR(cfg)
        synth.load_synthetic(m, seed=0)
        m.train()
        ddp = DDP(m)                                   # gradient all-reduce over the process group
        im0, im1 = synth.synth_image_pair(1, seed=50 + rank, hw=(96, 128), disparities=(8,))   # a DIFFERENT pair per rank
        ii = torch.arange(12 * 16)
        data = {'image0': torch.from_numpy(im0), 'image1': torch.from_numpy(im1),
                'spv_b_ids': torch.zeros(len(ii), dtype=torch.int64), 'spv_i_ids': ii, 'spv_j_ids': ii}
        m.coarse_matching.train_pad_num_gt_min = 8
        ddp(data, train=True)
        loss = -torch.log(data['conf_matrix'][0, ii, ii] + 1e-6).mean() + data['expec_f'].pow(2).mean()
        loss.backward()
        g = m.loftr_coarse.layers[0].q_proj.weight.grad
        gb = m.backbone.conv1.weight.grad
        q.put((rank, float(loss), float(g.double().sum()), float(g.norm()), float(gb.norm())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_two_ranks_gloo_gradient_allreduce():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, s0, n0, b0), (_, l1, s1, n1, b1) = res
    assert l0 != l1                                      # different data per rank ...
    assert s0 == pytest.approx(s1, rel=1e-6) and n0 == pytest.approx(n1, rel=1e-6)   # ... identical averaged grads
    assert b0 == pytest.approx(b1, rel=1e-6) and n0 > 0


@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_g18_masked_coarse_matching_matches_reference(mode):
    """Padded-mask batches through far_amd.loftr.stages.CoarseMatching on CPU tensors (the dense differentiable form, which is
    also what GPU training uses for masked batches) against the reference's own module on the same tensors and seed (golden
    G18): masked dual softmax, mask_border_with_padding, compute_max_candidates, the training-time sampling / GT padding
    (coarse_matching.py:28-57, 110-117, 199-240) -- ids bit-identical, confidences and the gradient through the masked
    softmax to fp32 round-off."""
    from far_amd.loftr.stages import CoarseMatching
    from tests.util import masked_coarse_inputs
    g = np.load(os.path.join(G, 'g18_masked_training_coarse.npz'))
    inp = masked_coarse_inputs()
    h, w = inp['h'], inp['w']
    cm = CoarseMatching(far_eval_config()['match_coarse'])
    cm.train(mode == 'train')
    cm.materialize_conf = True
    data = {'hw0_i': (8 * h, 8 * w), 'hw1_i': (8 * h, 8 * w), 'hw0_c': (h, w), 'hw1_c': (h, w),
            'mask0': torch.from_numpy(inp['mask0']), 'mask1': torch.from_numpy(inp['mask1']),
            'spv_b_ids': torch.from_numpy(inp['spv_b_ids']), 'spv_i_ids': torch.from_numpy(inp['spv_i_ids']),
            'spv_j_ids': torch.from_numpy(inp['spv_j_ids'])}
    f0 = torch.from_numpy(inp['f0']).requires_grad_(mode == 'train')
    f1 = torch.from_numpy(inp['f1']).requires_grad_(mode == 'train')
    torch.manual_seed(1234)
    if mode == 'train':
        cm(f0, f1, data, mask_c0=data['mask0'].flatten(-2), mask_c1=data['mask1'].flatten(-2))
    else:
        with torch.no_grad():
            cm._forward_train(f0, f1, data, data['mask0'].flatten(-2), data['mask1'].flatten(-2))    # the dense form, eval-mode selection
    for k in ('b_ids', 'i_ids', 'j_ids', 'gt_mask', 'm_bids'):
        np.testing.assert_array_equal(data[k].numpy(), g[f'{mode}_{k}'], err_msg=k)
    np.testing.assert_allclose(data['mkpts0_c'].numpy(), g[f'{mode}_mkpts0_c'])
    np.testing.assert_allclose(data['mkpts1_c'].numpy(), g[f'{mode}_mkpts1_c'])
    np.testing.assert_allclose(data['mconf'].detach().numpy(), g[f'{mode}_mconf'], rtol=2e-5, atol=1e-7)
    conf = data['conf_matrix']
    np.testing.assert_allclose(conf.detach().sum((1, 2)).numpy(), g[f'{mode}_conf_sum'], rtol=1e-5)
    np.testing.assert_allclose(conf.detach()[:, ::37, ::41].numpy(), g[f'{mode}_conf_sample'], rtol=2e-5, atol=1e-8)
    if mode == 'train':
        pos = conf[data['spv_b_ids'], data['spv_i_ids'], data['spv_j_ids']]
        pos.sum().backward()
        np.testing.assert_allclose(pos.detach().numpy(), g['train_pos_conf'], rtol=2e-5, atol=1e-8)
        np.testing.assert_allclose(f0.grad.norm().item(), float(g['train_df0_norm']), rtol=1e-4)
        np.testing.assert_allclose(f1.grad.norm().item(), float(g['train_df1_norm']), rtol=1e-4)
        np.testing.assert_allclose(f0.grad[:, ::53, ::17].numpy(), g['train_df0_sample'], rtol=1e-3, atol=1e-6)


def test_sync_batchnorm_modules_are_not_taken_by_the_batch_statistics_kernel():
    """The reference's multi-GPU training converts every BatchNorm2d to SyncBatchNorm (train.py:342): K19 computes LOCAL batch statistics
    and must leave those modules to torch (the predicate is a plain type test, checked here without a GPU)."""
    from far_amd.loftr import backbone as bb
    x = torch.zeros(1, 4, 2, 2)
    plain, sync = torch.nn.BatchNorm2d(4).train(), torch.nn.SyncBatchNorm(4).train()

    class _Cuda:                      # the predicate only looks at is_cuda / dtype
        is_cuda, dtype = True, torch.float32
    assert bb._bn_hip(plain, _Cuda) and not bb._bn_hip(sync, _Cuda)
    assert not bb._bn_hip(plain.eval(), _Cuda) and not bb._bn_hip(torch.nn.BatchNorm2d(4).train(), x)

"""K11 (far_pose_pack_f64 / far_pose_features_f32) against the torch statements of the same reference lines
(supervision.py:218-233, loftr.py:137-171); the reference golden for preprocess_helper is checked through the model in
test_pipeline_gpu.py::test_head_vs_reference_golden."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from far_amd import ops
    return ops


def test_pose_pack_matches_the_packaging_lines():
    ops = _ops()
    from far_amd.pose6d import compute_normalized_6d
    g = torch.Generator(device='cuda').manual_seed(3)
    B = 37
    R = torch.linalg.qr(torch.randn(B, 3, 3, device='cuda', dtype=torch.float64, generator=g))[0].contiguous()
    t = torch.randn(B, 3, device='cuda', dtype=torch.float64, generator=g)
    E = torch.randn(B, 3, 3, device='cuda', dtype=torch.float64, generator=g)
    status = (torch.rand(B, device='cuda', generator=g) > 0.3).to(torch.int32)
    counts = torch.randint(0, 12, (B,), device='cuda', generator=g, dtype=torch.int32)
    offs = torch.cat([torch.zeros(1, dtype=torch.int32, device='cuda'), counts.cumsum(0).to(torch.int32)])
    after, tight, ultra = (torch.randint(0, 2000, (B,), device='cuda', generator=g, dtype=torch.int32) for _ in range(3))
    sol = dict(R=R, t=t, E=E, status=status, num_after=after, tight=tight, ultra=ultra)
    rt, E2, before, a2, t2, u2 = ops.pose_pack(sol, offs)
    ok = status.bool()
    eye34 = torch.cat([torch.eye(3), torch.zeros(3, 1)], 1).to('cuda', torch.float64)
    ref_rt = torch.where(ok[:, None, None], torch.cat([R, t.unsqueeze(-1)], -1), eye34)
    assert torch.equal(rt, ref_rt)
    assert torch.equal(E2, torch.where(ok[:, None, None], E, torch.eye(3, device='cuda', dtype=torch.float64)))
    few = counts < 5
    assert before.dtype == torch.int64 and torch.equal(before, counts.long())
    for got, src in ((a2, after), (t2, tight), (u2, ultra)):
        assert got.dtype == torch.int32 and torch.equal(got, torch.where(few, torch.zeros_like(src), src))

    # pose features: rigid poses (the solver's) and a general affine one, int32 and int64 counts
    rt_any = rt.clone()
    rt_any[0, :, :3] += 0.3 * torch.randn(3, 3, device='cuda', dtype=torch.float64, generator=g)
    cnts = [a2, before, t2, u2]
    preds, inv = ops.pose_features(rt_any, cnts)
    last = torch.tensor([[[0, 0, 0, 1.]]], device='cuda', dtype=torch.float64).expand(B, -1, -1)
    rt_inv = torch.linalg.inv(torch.cat([rt_any, last], 1))[:, :3, :4]
    cnt = torch.cat([c.float().reshape(B, 1) / 500 for c in cnts], -1)
    ref_p = torch.cat([compute_normalized_6d(rt_any.float()), cnt], -1)
    ref_i = torch.cat([compute_normalized_6d(rt_inv).float(), cnt], -1)
    assert preds.shape == (B, 13)
    torch.testing.assert_close(preds, ref_p, rtol=1e-6, atol=1e-6)            # fp32 (v - mean) / std (IEEE division here)
    torch.testing.assert_close(inv, ref_i, rtol=2e-6, atol=2e-6)              # LU vs cofactor inverse, both float64
    p9, i9 = ops.pose_features(rt_any, [])
    assert p9.shape == (B, 9) and torch.equal(p9, preds[:, :9]) and torch.equal(i9, inv[:, :9])


def test_pose_features_rejects_bad_counts():
    ops = _ops()
    from far_amd._lib import FarHipError
    rt = torch.eye(3, 4, dtype=torch.float64, device='cuda')[None].repeat(4, 1, 1)
    with pytest.raises(FarHipError):
        ops.pose_features(rt, [torch.zeros(4, device='cuda')])                # float counts
    with pytest.raises(FarHipError):
        ops.pose_features(rt, [torch.zeros(3, dtype=torch.int32, device='cuda')])

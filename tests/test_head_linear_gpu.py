"""K15 (head_linear_f32.hip): the regression head's small dense layers as exact-fp32, row-independent kernels -- against
float64, and the property they exist for: a row's result does not depend on the batch it is computed in
(transformer.py:294-295, :423-431, :448-458)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,K,N', [(1, 35840, 1024), (32, 35840, 1024), (5, 22, 512), (32, 512, 512), (3, 512, 9), (70, 35862, 512), (2, 7, 2)])
def test_rows_linear_matches_float64_and_is_batch_invariant(B, K, N):
    from far_amd import ops
    g = torch.Generator(device='cuda').manual_seed(B + K + N)
    x = torch.randn(B, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    b = torch.randn(N, device='cuda', generator=g)
    add = torch.randn(B, N, device='cuda', generator=g)
    pr = ops.PackedRows(w, b)
    for act, fn in (('none', lambda t: t), ('relu', torch.relu), ('sigmoid', torch.sigmoid), ('gelu', torch.nn.functional.gelu)):
        y = ops.rows_linear(x, pr, act=act, add=add)
        ref = fn(x.double() @ w.double().t() + b.double() + add.double())
        err = float((y.double() - ref).abs().max())
        assert err < 2e-5, (act, err)                                    # fp32 chains of length K; measured ~2e-6
    y = ops.rows_linear(x, pr)
    for r in (0, B // 2, B - 1):                                          # a row alone == the row inside the batch, bit for bit
        assert torch.equal(ops.rows_linear(x[r:r + 1], pr)[0], y[r]), r
    xs = torch.randn(B, K + 5, device='cuda', generator=g)[:, :K]         # strided rows
    assert torch.equal(ops.rows_linear(xs, pr), ops.rows_linear(xs.contiguous(), pr))
    half = ops.PackedRows(w, None, cols=(0, K // 2))                       # a column range of the weight (moe_predictor's feature part)
    if K >= 2:
        ref = x[:, :K // 2].double() @ w[:, :K // 2].double().t()
        assert float((ops.rows_linear(x[:, :K // 2], half).double() - ref).abs().max()) < 2e-5


@pytest.mark.parametrize('Z,N,heads', [(8, 4800, 4), (3, 221, 1), (256, 4800, 4), (2, 64, 2)])
def test_emm_contract_matches_float64_and_is_problem_count_invariant(Z, N, heads):
    from far_amd import ops
    g = torch.Generator(device='cuda').manual_seed(Z + N)
    P = Z // heads
    v = torch.randn(heads, P, N, 64, device='cuda', generator=g)          # (head, problem) planes as the fused q | k | v projection writes them
    pos = torch.rand(N, 6, device='cuda', generator=g)
    T = torch.randn(Z, N, 70, device='cuda', generator=g) / N ** 0.5
    F = ops.emm_contract(ops._p(v), heads, P * N * 64, N * 64, pos, T)
    vz = v.permute(1, 0, 2, 3).reshape(Z, N, 64)                          # z = p * heads + hh
    vt = torch.cat([vz, pos[None].expand(Z, -1, -1)], 2).double()
    ref = vt.transpose(1, 2) @ T.double()
    assert float((F.double() - ref).abs().max()) < 5e-5 * max(1.0, float(ref.abs().max()))
    one = ops.emm_contract(ops._p(vz[Z - 1:].contiguous()), 1, 0, N * 64, pos, T[Z - 1:].contiguous())
    assert torch.equal(one[0], F[Z - 1])                                   # a problem alone == the problem in the batch

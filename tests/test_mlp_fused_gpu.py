"""K13 (far_mlp_fused_f16s): the MLP block of a LoFTR encoder layer at d_model = 128 in one launch -- against float64
(transformer.py:64-67 restated), against the two K9 launches it replaces, inside LoFTREncoderLayer, and its edge cases."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = 128


def _weights(seed, w_amp=1.0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    w0 = torch.randn(2 * D, 2 * D, device='cuda', generator=g) * (w_amp / (2 * D) ** 0.5)
    w2 = torch.randn(D, 2 * D, device='cuda', generator=g) * (w_amp / (2 * D) ** 0.5)
    gam = torch.rand(D, device='cuda', generator=g) + 0.5
    bet = torch.randn(D, device='cuda', generator=g)
    return w0, w2, gam, bet


def _ref64(x, m, w0, w2, gam, bet, eps):
    xd, md = x.double(), m.double()
    hid = torch.relu(torch.cat([xd, md], -1) @ w0.double().t()) @ w2.double().t()
    return xd + torch.nn.functional.layer_norm(hid, (D,), gam.double(), bet.double(), eps)


@pytest.mark.parametrize('rows', [1, 31, 128, 129, 25 * 777, 4000])
def test_mlp_fused_matches_float64_and_the_two_launch_path(rows):
    from far_amd import ops
    w0, w2, gam, bet = _weights(rows)
    g = torch.Generator(device='cuda').manual_seed(rows + 1)
    x = torch.randn(1, rows, D, device='cuda', generator=g)
    m = torch.randn(1, rows, D, device='cuda', generator=g)
    y = ops.mlp_fused(x, m, ops.PackedMlp(w0, w2), gam, bet, 1e-5)
    ref = _ref64(x, m, w0, w2, gam, bet, 1e-5)
    hid = ops.linear_f16s(x, ops.PackedConv(w0), act='relu', x2=m)
    two = ops.linear_f16s(hid, ops.PackedConv(w2), ln=(gam, bet, 1e-5), post_residual=x)
    sc = float(ref.abs().max())
    e1, e2 = float((y.double() - ref).abs().max()) / sc, float((two.double() - ref).abs().max()) / sc
    print(f'[k13] rows={rows}: fused vs float64 {e1:.2e}; two K9 launches vs float64 {e2:.2e}')
    assert e1 < 2e-6 and e2 < 2e-6                                   # measured 2.3e-7 .. 3.5e-7 for both
    assert torch.isfinite(y).all() and y.shape == x.shape


@pytest.mark.parametrize('rows', [31, 129, 25 * 777])
def test_mlp_fused_plain_fp16_operands(rows):
    """far_mlp_fused_f16 (LoFTR.set_precision('fp16')): plain fp16 operands; 16-bit-operand bar against float64, not the parity kernel."""
    from far_amd import ops
    w0, w2, gam, bet = _weights(rows)
    g = torch.Generator(device='cuda').manual_seed(rows + 1)
    x = torch.randn(1, rows, D, device='cuda', generator=g)
    m = torch.randn(1, rows, D, device='cuda', generator=g)
    pm = ops.PackedMlp(w0, w2)
    y = ops.mlp_fused(x, m, pm, gam, bet, 1e-5, plain16=True)
    ys = ops.mlp_fused(x, m, pm, gam, bet, 1e-5)
    ref = _ref64(x, m, w0, w2, gam, bet, 1e-5)
    sc = float(ref.abs().max())
    e, es = float((y.double() - ref).abs().max()) / sc, float((ys.double() - ref).abs().max()) / sc
    print(f'[k13 plain] rows={rows}: plain vs float64 {e:.2e} (split {es:.2e})')
    assert torch.isfinite(y).all() and es < e < 3e-3
    assert torch.equal(y, ops.mlp_fused(x, m, pm, gam, bet, 1e-5, plain16=True))


def test_mlp_fused_weight_and_activation_scales():
    """Tiny / large weights (power-of-two pre-scaling per tensor), small activations, an all-negative pre-activation
    (ReLU -> zero hidden -> LayerNorm of a zero row = beta)."""
    from far_amd import ops
    g = torch.Generator(device='cuda').manual_seed(5)
    for w_amp, x_amp in ((1e-3, 1.0), (200.0, 1.0), (1.0, 1e-2), (1.0, 30.0)):
        w0, w2, gam, bet = _weights(7, w_amp)
        x = torch.randn(1, 300, D, device='cuda', generator=g) * x_amp
        m = torch.randn(1, 300, D, device='cuda', generator=g) * x_amp
        y = ops.mlp_fused(x, m, ops.PackedMlp(w0, w2), gam, bet, 1e-5)
        ref = _ref64(x, m, w0, w2, gam, bet, 1e-5)
        assert float((y.double() - ref).abs().max()) < 3e-6 * float(ref.abs().max()), (w_amp, x_amp)
    w0, w2, gam, bet = _weights(9)
    w0 = -w0.abs()
    x = torch.rand(1, 64, D, device='cuda', generator=g)
    m = torch.rand(1, 64, D, device='cuda', generator=g)
    y = ops.mlp_fused(x, m, ops.PackedMlp(w0, w2), gam, bet, 1e-5)
    torch.testing.assert_close(y, x + bet, rtol=0, atol=1e-6)


def test_mlp_fused_rejects_bad_arguments():
    from far_amd import _lib, ops
    w0, w2, gam, bet = _weights(3)
    pm = ops.PackedMlp(w0, w2)
    x = torch.randn(1, 10, D, device='cuda')
    with pytest.raises(_lib.FarHipError):
        ops.mlp_fused(x, x, pm, gam, bet, 1e-5, out=x)               # out may not alias the inputs
    with pytest.raises(_lib.FarHipError):
        ops.PackedMlp(torch.randn(512, 512, device='cuda'), torch.randn(256, 512, device='cuda'))   # d_model 256: not built
    assert ops.mlp_fused(x[:, :0], x[:, :0], pm, gam, bet, 1e-5).shape == (1, 0, D)                  # no rows: no launch


def test_encoder_layer_with_fused_mlp_equals_the_two_launch_layer():
    """LoFTREncoderLayer(128, 8) in inference: fused_mlp = True vs False on fine-level-shaped windows (self and cross)."""
    from far_amd.loftr.transformer import LoFTREncoderLayer
    torch.manual_seed(0)
    layer = LoFTREncoderLayer(128, 8).cuda().eval()
    for p in layer.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_uniform_(p)
    x = torch.randn(900, 25, 128, device='cuda')
    s = torch.randn(900, 25, 128, device='cuda')
    with torch.no_grad():
        for src in (x, s):
            layer.fused_mlp = True
            a = layer(x, src)
            layer.fused_mlp = False
            b = layer(x, src)
            d = float((a - b).abs().max()) / float(b.abs().max())
            print(f'[k13 in layer] max relative difference {d:.2e}')
            assert d < 2e-6


def test_pack_follows_weight_updates():
    """The layer's PackCache rebuilds the K13 image when mlp weights change in place (optimizer step, load_state_dict)."""
    from far_amd.loftr.transformer import LoFTREncoderLayer
    torch.manual_seed(1)
    layer = LoFTREncoderLayer(128, 8).cuda().eval()
    x = torch.randn(40, 25, 128, device='cuda')
    with torch.no_grad():
        a = layer(x, x)
        layer.mlp[2].weight.mul_(0.5)
        b = layer(x, x)
        layer.fused_mlp = False
        c = layer(x, x)
    assert float((a - b).abs().max()) > 1e-3
    assert float((b - c).abs().max()) < 2e-6 * float(c.abs().max())


def test_mlp_fused_overflow_is_reported():
    """K13's operands are 2^4-scaled activations AND the 2^4-scaled hidden tensor: an input beyond 4094, or a hidden value
    beyond 4094 from moderate inputs (large weights), makes the result inf / NaN and sets the device flag; in range the flag
    stays clear."""
    from far_amd import ops
    g = torch.Generator(device='cuda').manual_seed(21)
    w0, w2, gam, bet = _weights(3)
    x = torch.randn(1, 500, D, device='cuda', generator=g)
    m = torch.randn(1, 500, D, device='cuda', generator=g)
    ops.overflow_flag('cuda').zero_()
    y = ops.mlp_fused(x, m, ops.PackedMlp(w0, w2), gam, bet, 1e-5)
    assert torch.isfinite(y).all() and not ops.activation_overflowed('cuda')
    xb = x.clone()
    xb[0, 77, 5] = 6000.0
    y = ops.mlp_fused(xb, m, ops.PackedMlp(w0, w2), gam, bet, 1e-5)
    # (the row's hidden values are NaN = inf - inf, which ReLU's max turns into 0: the OUTPUT row is finite and wrong -- which
    #  is why the hidden accumulators are part of the test the kernel makes, not only its outputs)
    assert torch.isfinite(y[0, :64]).all()
    assert ops.activation_overflowed('cuda')
    w0b = w0 * 400.0                                                   # hidden = relu(W0 [x | msg]) reaches ~ 400 * 4 sigma > 4094
    y = ops.mlp_fused(x * 4, m * 4, ops.PackedMlp(w0b, w2), gam, bet, 1e-5)
    assert ops.activation_overflowed('cuda') and not torch.isfinite(y).all()

"""K9 (split-fp16 implicit-GEMM convolution / linear) and K10 (stem) against float64 torch convolutions.

The bar: K9's split mode is an fp32-grade kernel -- its error against the float64 result must be of the size of an fp32
convolution's own rounding error (we allow 4x the fp32 direct-sum bound), far below the 1e-3 / 1e-5 tolerances the
pipeline parity tests put on the quantities computed from these features; plain mode (fp16 operands) is held to 2e-3.
Reference: mp3d_loftr/src/loftr/backbone/resnet_fpn.py:5-43, 101-119.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from far_amd import ops
    return ops


def _rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max()), float(
        ((a.double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())


@pytest.mark.parametrize('N,H,W,Cin,Cout,ks', [
    (2, 24, 40, 32, 64, 3),          # whole tiles
    (3, 30, 37, 196, 196, 3),        # ragged tiles, padded channel chunk, padded output tile
    (1, 17, 16, 128, 128, 3),        # 4 x 1 wave layout, ragged rows
    (2, 9, 50, 256, 196, 3),
    (1, 5, 7, 196, 128, 3),
    (2, 13, 21, 128, 196, 1),        # 1x1 convolution
    (1, 1, 1000, 256, 256, 1),       # linear layer shape
    (1, 1, 77, 512, 512, 1),         # two output-channel blocks
    (1, 1, 300, 256, 768, 1),
])
def test_conv_split_matches_float64(N, H, W, Cin, Cout, ks):
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(N * 1000 + Cin + Cout + ks)
    x = (torch.randn(N, H, W, Cin, device='cuda', generator=g) * 1.5).relu_()
    w = torch.randn(Cout, Cin, ks, ks, device='cuda', generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5
    scale = torch.rand(Cout, device='cuda', generator=g) + 0.5
    shift = torch.randn(Cout, device='cuda', generator=g) * 0.1
    res = torch.randn(N, H, W, Cout, device='cuda', generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=ks // 2).permute(0, 2, 3, 1)
    ref_bn = ref * scale.double() + shift.double()
    for split, tol_max, tol_rms in ((True, 4e-6, 1.5e-6), (False, 4e-3, 2e-3)):
        pc = ops.PackedConv(w, scale, shift, split=split)
        for act, residual in (('none', None), ('relu', res), ('leaky', None)):
            y = ops.conv_nhwc(x, pc, residual=residual, act=act, slope=0.01)
            r = ref_bn + (residual.double() if residual is not None else 0)
            r = {'none': lambda t: t, 'relu': torch.relu, 'leaky': lambda t: F.leaky_relu(t, 0.01)}[act](r)
            emax, erms = _rel(y, r)
            assert emax < tol_max and erms < tol_rms, (split, act, emax, erms)


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(2, 48, 64, 128, 196), (1, 31, 45, 196, 256), (1, 16, 16, 64, 64), (2, 9, 7, 32, 40)])
def test_conv_stride2_matches_float64(N, H, W, Cin, Cout):
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(Cin + Cout + H)
    x = torch.randn(N, H, W, Cin, device='cuda', generator=g).relu_()
    w = torch.randn(Cout, Cin, 3, 3, device='cuda', generator=g) * (2.0 / (Cin * 9)) ** 0.5
    scale = torch.rand(Cout, device='cuda', generator=g) + 0.5
    shift = torch.randn(Cout, device='cuda', generator=g) * 0.1
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), stride=2, padding=1).permute(0, 2, 3, 1)
    ref = torch.relu(ref * scale.double() + shift.double())
    y = ops.conv_nhwc(x, ops.PackedConv(w, scale, shift, stride=2), act='relu')
    assert y.shape == ref.shape
    emax, erms = _rel(y, ref)
    assert emax < 4e-6 and erms < 1.5e-6, (emax, erms)


def test_conv_split_is_as_accurate_as_fp32_direct():
    """The split product's error is of the order of an fp32 direct convolution's (both measured against float64)."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(2, 40, 48, 256, device='cuda', generator=g).relu_()
    w = torch.randn(256, 256, 3, 3, device='cuda', generator=g) * 0.03
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1)
    direct = F.conv2d(x.permute(0, 3, 1, 2), w, padding=1)
    y = ops.conv_nhwc(x, ops.PackedConv(w)).permute(0, 3, 1, 2)
    e_mine, e_direct = _rel(y, ref)[1], _rel(direct, ref)[1]
    assert e_mine < 3 * e_direct + 1e-7, (e_mine, e_direct)


def test_conv_weight_scale_extremes():
    """Tiny and large weights: the power-of-two pre-scaling keeps the fp16 lo parts normal."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(11)
    x = torch.randn(1, 16, 16, 64, device='cuda', generator=g)
    for mag in (1e-4, 1.0, 300.0):
        w = torch.randn(64, 64, 3, 3, device='cuda', generator=g) * mag
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
        emax, erms = _rel(ops.conv_nhwc(x, ops.PackedConv(w)), ref)
        assert emax < 4e-6 and erms < 1.5e-6, (mag, emax, erms)


def test_conv_small_activations_keep_precision():
    """Activations 1e-3 of the usual scale lose only the documented absolute floor (2^-29 of unit scale)."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(12)
    x = torch.randn(1, 16, 16, 64, device='cuda', generator=g) * 1e-3
    w = torch.randn(64, 64, 3, 3, device='cuda', generator=g) * 0.05
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    y = ops.conv_nhwc(x, ops.PackedConv(w))
    assert float((y.double() - ref).abs().max()) < 5e-8


def test_conv_activation_overflow_is_loud():
    """K9 splits activations after a fixed 2^4 scale: |a| > 4094 leaves fp16's range.  That must never produce a
    plausible finite number: every output that reads such an activation is inf / NaN, every other output is still right."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(13)
    x = torch.randn(1, 16, 16, 64, device='cuda', generator=g)
    w = torch.randn(64, 64, 3, 3, device='cuda', generator=g) * 0.05
    ok = ops.conv_nhwc(x * 200.0, ops.PackedConv(w))                  # |a| up to ~900: inside the window
    ref = F.conv2d((x * 200.0).permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    assert torch.isfinite(ok).all() and _rel(ok, ref)[0] < 4e-6
    xb = x.clone()
    xb[0, 5, 7, 3] = 5000.0
    y = ops.conv_nhwc(xb, ops.PackedConv(w))
    hit = torch.zeros(16, 16, dtype=torch.bool, device='cuda')
    hit[4:7, 6:9] = True                                              # the 3x3 outputs that read pixel (5, 7)
    assert not torch.isfinite(y[0][hit]).any()                        # all 9 x 64 affected outputs are inf / NaN
    good = ops.conv_nhwc(x, ops.PackedConv(w))
    assert torch.equal(y[0][~hit], good[0][~hit])                     # and nothing else moved


def test_conv_activation_range_flag_and_recovery():
    """The recovery path behind the loud overflow: every K9 launch reports a non-finite accumulator through the device flag
    (ops.overflow_flag); under a lower activation exponent the same tensor -- values up to 1e5 next to values of 1e-4 --
    convolves at fp32-grade accuracy (error relative to the tensor's scale, as for an fp32 convolution), for every epilogue
    form (3x3 wide / narrow, 1x1 linear with fused LayerNorm)."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(14)
    x = torch.randn(2, 20, 24, 64, device='cuda', generator=g)
    x[0, 3, 4, 5] = 1.0e5
    x[1, 10, 11, 12] = -7.0e4
    x[:, :, :, 32:] *= 1e-4                                           # half of the channels at the small end
    flag = ops.overflow_flag('cuda')
    flag.zero_()
    for Cout, ks in ((64, 3), (196, 3), (40, 1)):
        w = torch.randn(Cout, 64, ks, ks, device='cuda', generator=g) * 0.05
        pc = ops.PackedConv(w)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=ks // 2).permute(0, 2, 3, 1)
        y = ops.conv_nhwc(x, pc)                                      # default exponent 4: 1e5 * 16 is beyond fp16
        assert not torch.isfinite(y).all()
        assert ops.activation_overflowed('cuda') and int(flag.item()) == 0      # reported, and reset by the read
        with pytest.raises(ops.ActivationOverflow):
            ops.conv_nhwc(x, pc)
            ops.check_activation_range('cuda', 'test')
        with ops.activation_exponent(-4):                             # |a| <= 65504 * 16 ~ 1e6
            y = ops.conv_nhwc(x, pc)
        assert torch.isfinite(y).all() and not ops.activation_overflowed('cuda')
        emax, erms = _rel(y, ref)
        assert emax < 4e-6, (Cout, ks, emax, erms)
        # the small-end channels alone: their contribution is resolved to the fp16 subnormal step of the split, 2^-24 / 2^-4
        xs = x.clone()
        xs[:, :, :, :32] = 0
        refs = F.conv2d(xs.permute(0, 3, 1, 2).double(), w.double(), padding=ks // 2).permute(0, 2, 3, 1)
        with ops.activation_exponent(-4):
            ys = ops.conv_nhwc(xs, pc)
        assert float((ys.double() - refs).abs().max()) < 64 * ks * ks * 0.2 * 2.0 ** -20      # sum |w| x half a step of 2^-20
    # linear layer with the fused LayerNorm epilogue
    xl = torch.randn(300, 256, device='cuda', generator=g) * 3.0e4
    wl = torch.randn(256, 256, device='cuda', generator=g) * 0.06
    gam, bet = torch.rand(256, device='cuda', generator=g) + 0.5, torch.randn(256, device='cuda', generator=g)
    pl = ops.PackedConv(wl)
    y = ops.linear_f16s(xl, pl, ln=(gam, bet, 1e-5))
    assert ops.activation_overflowed('cuda')
    with ops.activation_exponent(-4):                                 # |x| reaches ~1.3e5: beyond exponent 0's 65504
        y = ops.linear_f16s(xl, pl, ln=(gam, bet, 1e-5))
    assert not ops.activation_overflowed('cuda')
    ref = F.layer_norm(xl.double() @ wl.double().t(), (256,), gam.double(), bet.double(), 1e-5)
    assert float((y.double() - ref).abs().max()) < 2e-5


def test_linear_wrapper_matches_addmm():
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(2, 333, 256, device='cuda', generator=g)
    w = torch.randn(512, 256, device='cuda', generator=g) * 0.06
    b = torch.randn(512, device='cuda', generator=g)
    pc = ops.PackedConv(w, None, b)
    y = ops.linear_f16s(x, pc, act='relu')
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    assert y.shape == (2, 333, 512)
    emax, erms = _rel(y, ref)
    assert emax < 4e-6 and erms < 1.5e-6


def test_conv_rejects_bad_arguments():
    ops = _ops()
    from far_amd._lib import FarHipError
    w = torch.randn(64, 30, 3, 3, device='cuda')
    pc = ops.PackedConv(w)                                    # packing any Cin is fine ...
    with pytest.raises(FarHipError):                          # ... but the kernel reads 16-byte channel groups
        ops.conv_nhwc(torch.randn(1, 8, 8, 30, device='cuda'), pc)
    with pytest.raises(FarHipError):
        ops.PackedConv(torch.randn(8, 8, 5, 5, device='cuda'))
    with pytest.raises(FarHipError):
        ops.conv_nhwc(torch.randn(1, 8, 8, 32, device='cuda'), ops.PackedConv(torch.randn(8, 64, 3, 3, device='cuda')))
    with pytest.raises(FarHipError):                          # LeakyReLU slope outside [0, 1] (the epilogue takes max(v, v * slope))
        ops.conv_nhwc(torch.randn(1, 8, 8, 32, device='cuda'), ops.PackedConv(torch.randn(8, 32, 3, 3, device='cuda')),
                      act='leaky', slope=1.5)


@pytest.mark.parametrize('N,H,W,Cout', [(2, 48, 64, 128), (1, 37, 51, 128), (1, 480, 640, 128), (2, 20, 24, 64)])
def test_stem_matches_float64(N, H, W, Cout):
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(H + W)
    img = torch.rand(N, 1, H, W, device='cuda', generator=g)
    w = torch.randn(Cout, 1, 7, 7, device='cuda', generator=g) * 0.2
    scale = torch.rand(Cout, device='cuda', generator=g) + 0.5
    shift = torch.randn(Cout, device='cuda', generator=g) * 0.1
    y = ops.stem7x7(img, w, scale, shift)
    ref = torch.relu(F.conv2d(img.double(), w.double(), stride=2, padding=3) * scale.double()[None, :, None, None]
                     + shift.double()[None, :, None, None]).permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    emax, erms = _rel(y, ref)
    assert emax < 2e-6 and erms < 5e-7, (emax, erms)
    # round 4 (split-fp16 inference form): images in 0..255 and very dark images keep the accuracy; a non-finite pixel stays visible
    for amp in (255.0, 1e-3):
        ya = ops.stem7x7(img * amp, w, scale, shift)
        ra = torch.relu(F.conv2d((img * amp).double(), w.double(), stride=2, padding=3) * scale.double()[None, :, None, None]
                        + shift.double()[None, :, None, None]).permute(0, 2, 3, 1)
        assert _rel(ya, ra)[0] < 2e-6, (amp, _rel(ya, ra))
    bad = img.clone()
    bad[0, 0, H // 2, W // 2] = float('inf')
    assert not torch.isfinite(ops.stem7x7(bad, w, scale, shift)).all()


@pytest.mark.parametrize('H,W', [(96, 128), (136, 184), (64, 96)])      # whole tiles; ragged tiles at every level (BASELINE C5 is 544x720); a level narrower than 32 (merge through K8)
def test_fused_backbone_matches_reference_modules(H, W):
    """The NHWC kernel path of the backbone against its own reference-style torch modules in float64."""
    from far_amd.config import far_eval_config
    from far_amd.loftr.backbone import build_backbone
    torch.manual_seed(0)
    cfg = far_eval_config()
    bb = build_backbone(cfg).cuda().eval()
    with torch.no_grad():
        for m in bb.modules():                               # non-trivial BatchNorm statistics
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1)
        x = torch.rand(2, 1, H, W, device='cuda')
        c_f, f_f = bb(x)                                      # fused path
        bb64 = build_backbone(cfg).cuda().eval().double()
        bb64.load_state_dict({k: v.double() for k, v in bb.state_dict().items()})
        c_r, f_r = bb64(x.double())
        assert c_f.shape == c_r.shape and f_f.shape == f_r.shape
        assert _rel(c_f, c_r)[0] < 2e-5 and _rel(f_f, f_r)[0] < 2e-5
        # round 4: the 196-channel maps are stored with 208 channels (zero weights for the extra ones) -- bit-identical to the
        # unpadded layout; and K17 vs K9 on the 3x3 layers -- both within the bar above
        from far_amd.loftr import backbone as bbm
        ops = _ops()
        bbm.PAD_CHANNELS = False
        try:
            c_u, f_u = bb(x)
        finally:
            bbm.PAD_CHANNELS = True
        assert torch.equal(c_u, c_f) and torch.equal(f_u, f_f)
        ops.USE_WINO = False
        try:
            c_9, f_9 = bb(x)
        finally:
            ops.USE_WINO = True
        assert _rel(c_9, c_r)[0] < 2e-5 and _rel(f_9, f_r)[0] < 2e-5
        print(f'[deviation] backbone {H}x{W} vs float64 modules: K17 path {_rel(c_f, c_r)[0]:.2e} / {_rel(f_f, f_r)[0]:.2e}, '
              f'K9 path {_rel(c_9, c_r)[0]:.2e} / {_rel(f_9, f_r)[0]:.2e} (coarse / fine, of the map maximum)')
        bb.trunk_split = bb.fpn_split = False                 # plain fp16 operands
        c_h, f_h = bb(x)
        assert _rel(c_h, c_r)[1] < 5e-3 and _rel(f_h, f_r)[1] < 5e-3


def test_conv_fused_concat_input():
    """x2: the convolution input is cat([x, x2], -1) read in place (transformer.py:64)."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(3, 210, 256, device='cuda', generator=g)
    m = torch.randn(3, 210, 256, device='cuda', generator=g)
    w = torch.randn(512, 512, device='cuda', generator=g) * 0.04
    y = ops.linear_f16s(x, ops.PackedConv(w), act='relu', x2=m)
    ref = torch.relu(torch.cat([x, m], -1).double() @ w.double().t())
    emax, erms = _rel(y, ref)
    assert emax < 4e-6 and erms < 1.5e-6
    xs = torch.randn(1, 9, 11, 64, device='cuda', generator=g)
    ms = torch.randn(1, 9, 11, 32, device='cuda', generator=g)
    w3 = torch.randn(40, 96, 3, 3, device='cuda', generator=g) * 0.05
    y3 = ops.conv_nhwc(xs, ops.PackedConv(w3), x2=ms)
    ref3 = F.conv2d(torch.cat([xs, ms], -1).permute(0, 3, 1, 2).double(), w3.double(), padding=1).permute(0, 2, 3, 1)
    assert _rel(y3, ref3)[0] < 4e-6


def test_linear_output_planes():
    """out_planes: fused q | k | v projections land in separate contiguous tensors."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(8)
    for K, P, Csub in ((256, 3, 256), (128, 3, 128), (256, 2, 256)):
        x = torch.randn(2, 301, K, device='cuda', generator=g)
        w = torch.randn(P * Csub, K, device='cuda', generator=g) * 0.05
        y = ops.linear_f16s(x, ops.PackedConv(w), out_planes=P)
        assert y.shape == (P, 2, 301, Csub) and y.is_contiguous()
        ref = (x.double() @ w.double().t()).view(2, 301, P, Csub).permute(2, 0, 1, 3)
        assert _rel(y, ref)[0] < 4e-6


def test_pack_cache_follows_weight_updates():
    """The packed K9 images are rebuilt when a parameter or BatchNorm buffer changes in place (optimizer step,
    load_state_dict): the fused backbone must track its reference-style modules after every kind of update."""
    from far_amd.config import far_eval_config
    from far_amd.loftr.backbone import build_backbone
    torch.manual_seed(1)
    bb = build_backbone(far_eval_config()).cuda().eval()
    ref = build_backbone(far_eval_config()).cuda().eval().double()
    x = torch.rand(1, 1, 64, 96, device='cuda')

    def check():
        ref.load_state_dict({k: v.double() for k, v in bb.state_dict().items()})
        with torch.no_grad():
            c, f = bb(x)
            cr, fr = ref(x.double())
        assert _rel(c, cr)[0] < 2e-5 and _rel(f, fr)[0] < 2e-5

    check()
    with torch.no_grad():
        c0 = bb(x)[0].clone()
        bb.layer1[0].conv1.weight.mul_(1.5)                     # in-place parameter update
        bb.layer2[1].bn2.running_var.add_(0.3)                   # buffer update (folded into the epilogue vectors)
    check()
    with torch.no_grad():
        assert not torch.equal(bb(x)[0], c0)
        sd = {k: v * (1.1 if k.endswith('layer3_outconv.weight') else 1.0) for k, v in bb.state_dict().items()}
    bb.load_state_dict(sd)                                       # copy_ into the same storage: version bump
    check()


def test_linear_grouped_residual():
    """res_group: one residual row per group of consecutive rows (the repeated coarse feature of FinePreprocess)."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(14)
    M, G, K, Co = 333, 25, 128, 128
    x = torch.randn(M, G, K, device='cuda', generator=g)
    r = torch.randn(M, Co, device='cuda', generator=g)
    w = torch.randn(Co, K, device='cuda', generator=g) * 0.08
    y = ops.linear_f16s(x, ops.PackedConv(w), residual=r, res_group=G)
    ref = x.double() @ w.double().t() + r.double()[:, None, :]
    assert y.shape == (M, G, Co) and _rel(y, ref)[0] < 4e-6
    w2 = torch.randn(196, K, device='cuda', generator=g) * 0.08          # narrow (per-register) epilogue path: Cout 196 -> planes
    r2 = torch.randn(M, 196, device='cuda', generator=g)
    y2 = ops.linear_f16s(x, ops.PackedConv(w2), residual=r2, res_group=G)
    assert _rel(y2, x.double() @ w2.double().t() + r2.double()[:, None, :])[0] < 4e-6


@pytest.mark.parametrize('rows,K,Co', [(700, 256, 256), (1000, 128, 128), (333, 512, 256), (77, 256, 128)])
def test_linear_fused_layernorm(rows, K, Co):
    """ln / post_residual: LayerNorm over the output channels (+ residual) in the epilogue == Linear then nn.LayerNorm."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(rows)
    x = torch.randn(3, rows, K, device='cuda', generator=g)
    w = torch.randn(Co, K, device='cuda', generator=g) * 0.06
    gamma = torch.rand(Co, device='cuda', generator=g) + 0.5
    beta = torch.randn(Co, device='cuda', generator=g) * 0.2
    r = torch.randn(3, rows, Co, device='cuda', generator=g)
    out = torch.empty(3, rows, Co, device='cuda')
    y = ops.linear_f16s(x, ops.PackedConv(w), ln=(gamma, beta, 1e-5), post_residual=r, out=out)
    assert y.data_ptr() == out.data_ptr()
    lin = x.double() @ w.double().t()
    ref = F.layer_norm(lin, (Co,), gamma.double(), beta.double(), 1e-5) + r.double()
    assert float((y.double() - ref).abs().max()) < 2e-5
    y2 = ops.linear_f16s(x, ops.PackedConv(w), ln=(gamma, beta, 1e-5))
    assert float((y2.double() - (ref - r.double())).abs().max()) < 2e-5


def test_conv_random_shapes_sweep():
    """Seeded sweep over odd geometries (ragged tiles, channel counts that are not multiples of 16 / 32, both kernel
    sizes, both strides, both operand modes): every case against the float64 convolution."""
    ops = _ops()
    rng = __import__('random').Random(2024)
    g = torch.Generator(device='cuda').manual_seed(99)
    for case in range(40):
        ks = rng.choice([1, 3, 3])
        stride = rng.choice([1, 1, 2]) if ks == 3 else 1
        N = rng.randint(1, 3)
        H, W = rng.randint(1, 37), rng.randint(1, 45)
        Cin = 4 * rng.randint(1, 80)
        Cout = rng.choice([1, 7, 32, 33, 64, 100, 128, 129, 196, 256, 300])
        split = rng.random() < 0.75
        x = torch.randn(N, H, W, Cin, device='cuda', generator=g)
        w = torch.randn(Cout, Cin, ks, ks, device='cuda', generator=g) * (1.5 / (Cin * ks * ks)) ** 0.5
        b = torch.randn(Cout, device='cuda', generator=g) * 0.1
        y = ops.conv_nhwc(x, ops.PackedConv(w, None, b, split=split, stride=stride), act='leaky', slope=0.2)
        ref = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=stride, padding=ks // 2),
                           0.2).permute(0, 2, 3, 1)
        assert y.shape == ref.shape, (case, y.shape, ref.shape)
        emax = float((y.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
        assert emax < (5e-6 if split else 5e-3), (case, ks, stride, N, H, W, Cin, Cout, split, emax)


def test_linear_three_planes_of_128_and_residual_activations():
    """Cout = 384 runs as three 128-channel blocks (the fused q | k | v of the d_model-128 layers); and the three
    activation modes of the wide epilogue with a residual, on ragged row counts, against float64."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(5)
    rows = 3 * 25 * 37 + 11
    x = torch.randn(rows, 128, device='cuda', generator=g)
    w = torch.randn(384, 128, device='cuda', generator=g) * 0.09
    q, k, v = ops.linear_f16s(x, ops.PackedConv(w), out_planes=3)
    ref = x.double() @ w.double().t()
    for i, t in enumerate((q, k, v)):
        emax, _ = _rel(t, ref[:, 128 * i:128 * (i + 1)])
        assert emax < 5e-6, (i, emax)
    xi = torch.randn(2, 19, 23, 64, device='cuda', generator=g)
    wc = torch.randn(128, 64, 3, 3, device='cuda', generator=g) * 0.05
    res = torch.randn(2, 19, 23, 128, device='cuda', generator=g)
    base = F.conv2d(xi.permute(0, 3, 1, 2).double(), wc.double(), padding=1).permute(0, 2, 3, 1) + res.double()
    for act, fn in (('none', lambda t: t), ('relu', torch.relu), ('leaky', lambda t: F.leaky_relu(t, 0.01))):
        y = ops.conv_nhwc(xi, ops.PackedConv(wc), residual=res, act=act, slope=0.01)
        emax, _ = _rel(y, fn(base))
        assert emax < 5e-6, (act, emax)


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(2, 24, 32, 128, 196), (1, 30, 44, 196, 256), (3, 16, 64, 64, 128), (2, 60, 160, 196, 256),
                                            (1, 10, 320, 128, 208), (2, 6, 96, 32, 100)])
def test_conv_fused_upsample_merge_is_bit_identical_to_conv_plus_k8(N, H, W, Cin, Cout):
    """The FPN merge in the 1x1 convolution's epilogue (up=) against the two-kernel sequence conv -> K8, and K8's own
    parity target F.interpolate(scale_factor=2, bilinear, align_corners=True).  Rows of whole 32-pixel tiles (W % 32 == 0) take
    the row-walking form of the epilogue (each lane 16 consecutive pixels, the left source pair reused): same bits as the generic
    form (tuning knob 13)."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(N * H + W)
    x = torch.randn(N, H, W, Cin, device='cuda', generator=g)
    coarse = torch.randn(N, H // 2, W // 2, Cout, device='cuda', generator=g)
    pc = ops.PackedConv(torch.randn(Cout, Cin, 1, 1, device='cuda', generator=g) * 0.08)
    fused = ops.conv_nhwc(x, pc, up=coarse)
    lateral = ops.conv_nhwc(x, pc)
    two = ops.upsample2x_add(coarse.permute(0, 3, 1, 2), lateral.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert torch.equal(fused, two.contiguous())
    from far_amd import _lib
    _lib.load().far_set_tuning(13, 1)
    try:
        assert torch.equal(fused, ops.conv_nhwc(x, pc, up=coarse))
    finally:
        _lib.load().far_set_tuning(13, 0)
    # torch's own fp32 form is the parity target: its source index rx * X is an fp32 product (at 160 columns a float64 index
    # already differs by 1e-5 of a unit-variance map)
    ref = lateral + F.interpolate(coarse.permute(0, 3, 1, 2), scale_factor=2., mode='bilinear', align_corners=True).permute(0, 2, 3, 1)
    assert float((fused - ref).abs().max()) < 1e-5
    from far_amd._lib import FarHipError
    with pytest.raises(FarHipError):                      # 3x3 convolutions have no fused merge
        ops.conv_nhwc(x, ops.PackedConv(torch.randn(Cout, Cin, 3, 3, device='cuda', generator=g)), up=coarse)


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(2, 24, 32, 128, 196), (3, 15, 21, 64, 128), (1, 240, 320, 128, 208), (5, 8, 6, 196, 256)])
def test_conv1x1_on_subsampled_input_in_place(N, H, W, Cin, Cout):
    """in_stride = 2: the down-sampling shortcut conv1x1(stride 2) + BatchNorm (resnet_fpn.py:26-29) reading x[:, ::2, ::2] in
    place -- the same bits as the convolution of the subsampled copy (odd sizes included), with residual and activation."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(H * W + Cin)
    x = torch.randn(N, H, W, Cin, device='cuda', generator=g)
    pc = ops.PackedConv(torch.randn(Cout, Cin, 1, 1, device='cuda', generator=g) * 0.1, torch.rand(Cout, device='cuda', generator=g) + 0.5,
                        torch.randn(Cout, device='cuda', generator=g))
    sub = x[:, ::2, ::2, :].contiguous()
    res = torch.randn(*sub.shape[:3], Cout, device='cuda', generator=g)
    assert torch.equal(ops.conv_nhwc(x, pc, in_stride=2), ops.conv_nhwc(sub, pc))
    assert torch.equal(ops.conv_nhwc(x, pc, residual=res, act='relu', in_stride=2), ops.conv_nhwc(sub, pc, residual=res, act='relu'))
    from far_amd._lib import FarHipError
    with pytest.raises(FarHipError):
        ops.conv_nhwc(x, ops.PackedConv(torch.randn(Cout, Cin, 3, 3, device='cuda', generator=g)), in_stride=2)


def test_conv_is_run_to_run_deterministic_under_load():
    """K9's slab / pixel waits count memory requests; an under-wait would be a race.  Same launch repeated with other
    work in flight on a second stream: every output bit-identical to the first (tools/k9_stress.py is the long form)."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(8)
    side = torch.cuda.Stream()
    junk = torch.randn(2048, 2048, device='cuda')
    for (N, H, W, Cin, Cout, ks, st) in [(4, 60, 80, 196, 196, 3, 1), (4, 60, 80, 128, 128, 3, 1), (1, 1, 40000, 256, 256, 1, 1),
                                         (4, 60, 80, 128, 196, 3, 2)]:
        for split in (True, False):
            x = torch.randn(N, H, W, Cin, device='cuda', generator=g)
            pc = ops.PackedConv(torch.randn(Cout, Cin, ks, ks, device='cuda', generator=g) * 0.03, split=split, stride=st)
            first = ops.conv_nhwc(x, pc, act='relu')
            for r in range(8):
                with torch.cuda.stream(side):
                    for _ in range(r % 3):
                        junk = junk @ junk * 1e-3
                assert torch.equal(ops.conv_nhwc(x, pc, act='relu'), first), (N, H, W, Cin, Cout, ks, st, split, r)
    torch.cuda.synchronize()


@pytest.mark.parametrize('rows,K,Co', [(4800, 256, 256), (1500, 128, 128), (4800, 256, 768), (333, 512, 256), (9600, 256, 128), (70, 256, 196)])
def test_linear_few_row_paths_equal_full_height(rows, K, Co):
    """Few-row Linear launches run the few-row kernel (linear_small_f16s.hip: K % 32 == 0, K <= 512, no fused LayerNorm) or K9 on
    32 MW-row tiles instead of 64 MW: the same products in the same order per output, so every epilogue (bias / activation /
    residual / LayerNorm / output planes) equals the full-height K9 launch bit for bit (tuning knob 7 forces that path)."""
    from far_amd import _lib
    ops = _ops()
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(rows + Co)
    x = torch.randn(rows, K, device='cuda', generator=g)
    w = torch.randn(Co, K, device='cuda', generator=g) * 0.06
    b = torch.randn(Co, device='cuda', generator=g)
    r = torch.randn(rows, Co, device='cuda', generator=g)
    gamma, beta = torch.rand(Co, device='cuda', generator=g) + 0.5, torch.randn(Co, device='cuda', generator=g) * 0.2
    pc, pcb = ops.PackedConv(w), ops.PackedConv(w, None, b)

    def variants():
        out = [ops.linear_f16s(x, pc), ops.linear_f16s(x, pcb, act='relu'), ops.linear_f16s(x, pc, residual=r, act='leaky')]
        if Co in (128, 256):
            out.append(ops.linear_f16s(x, pc, ln=(gamma, beta, 1e-5), post_residual=r))
        if Co % 3 == 0 and (Co // 3) % 4 == 0:
            out.append(ops.linear_f16s(x, pc, out_planes=3))
        # the dgrad form of the training path: device activation scale, residual accumulation, two output planes
        gsc = ops.grad_scale(x * 1e-5)
        out.append(ops.linear_f16s(x * 1e-5, pc, residual=r, act_scale_dev=gsc))
        if Co % 8 == 0:
            out.append(ops.linear_f16s(x * 1e-5, pc, out_planes=2, act_scale_dev=gsc))
        return out
    small = variants()
    lib.far_set_tuning(7, 1)
    try:
        full = variants()
    finally:
        lib.far_set_tuning(7, 0)
    for a, bb in zip(small, full):
        assert torch.equal(a, bb)
    assert _rel(small[0], x.double() @ w.double().t())[0] < 4e-6


def test_pack_table_repacks_every_training_image_like_a_fresh_pack():
    """After an optimizer-style in-place update of ALL weights the first stale lookup re-packs every registered training image in
    two launches (ops.PACK_TABLE -> far_pack_table_run); the images, their scales and the epilogue scale vectors must equal
    freshly built ones bit for bit, for forward and dgrad / transposed images, 1x1, 3x3 and stride 2.  A partial update (one
    weight) takes the per-image path and gives the same."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(3)
    ws = {'lin': torch.randn(256, 128, device='cuda', generator=g) * 0.05, 'c3': torch.randn(196, 128, 3, 3, device='cuda', generator=g) * 0.03,
          'c3s2': torch.randn(256, 196, 3, 3, device='cuda', generator=g) * 0.02, 'c1': torch.randn(196, 256, 1, 1, device='cuda', generator=g)}
    for w in ws.values():
        w.requires_grad_(True)
    cache = ops.PackCache()

    def images():
        out = {'lin': ops.train_pack(cache, 'lin', ws['lin']), 'linT': ops.train_pack_t(cache, 'lin', ws['lin'])}
        for name, st in (('c3', 1), ('c3s2', 2), ('c1', 1)):
            x = torch.randn(1, ws[name].shape[1], 16, 16, device='cuda', generator=g).requires_grad_(True)
            ops.conv_train(x, ws[name], st, cache, name).sum().backward()           # forward + dgrad images get (re)built
            out[name] = cache.get((name, 'fwd', True), [ws[name]], None)
            out[name + 'd'] = cache.get((name, 'dgrad', True), [ws[name]], None)
        return out

    def fresh():
        f = {'lin': ops.PackedConv(ws['lin']), 'c3': ops.PackedConv(ws['c3']), 'c3s2': ops.PackedConv(ws['c3s2'], stride=2), 'c1': ops.PackedConv(ws['c1'])}
        f['linT'] = ops.PackedConv(ws['lin'], dgrad=True, pack_scale=f['lin'].pack_scale)
        for n in ('c3', 'c3s2', 'c1'):
            f[n + 'd'] = ops.PackedConv(ws[n], dgrad=True, pack_scale=f[n].pack_scale)
        return f

    def same(a, b):
        for k in b:
            assert torch.equal(a[k].packed, b[k].packed) and torch.equal(a[k].pack_scale, b[k].pack_scale) and torch.equal(a[k].scale, b[k].scale), k

    first = images()
    same(first, fresh())
    ptrs = {k: v.packed.data_ptr() for k, v in first.items()}
    with torch.no_grad():
        for i, w in enumerate(ws.values()):
            w.mul_(1.7 + i).add_(0.01)                       # "optimizer step": every weight changes in place
    second = images()
    assert all(second[k] is first[k] and second[k].packed.data_ptr() == ptrs[k] for k in first)       # same objects, same buffers
    same(second, fresh())
    with torch.no_grad():
        ws['c3'].mul_(0.5)                                   # one weight only: per-image refresh
    same(images(), fresh())
    ops.USE_PACK_TABLE = False
    try:
        with torch.no_grad():
            for w in ws.values():
                w.mul_(1.1)
        same(images(), fresh())
    finally:
        ops.USE_PACK_TABLE = True


@pytest.mark.parametrize('table', [True, False])
def test_training_forward_on_k17_follows_the_weights_after_an_optimizer_step(table):
    """The training forward of a stride-1 3x3 layer runs on K17 at >= 32x32 inputs (autograd.Function.forward has grad mode
    off).  Its Winograd image must follow every re-pack of the K9 image -- the per-image refresh AND the whole-model table
    (far_pack_table_run re-packs K9 images only): after an in-place update of all weights the next forward equals the forward
    of a freshly built pack bit for bit, in the same buffers."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(11)
    ws = {n: (torch.randn(co, ci, 3, 3, device='cuda', generator=g) * 0.03).requires_grad_(True)
          for n, (co, ci) in {'a': (128, 128), 'b': (196, 128), 'c': (64, 196)}.items()}
    cache = ops.PackCache()
    xs = {n: torch.randn(2, w.shape[1], 40, 36, device='cuda', generator=g) for n, w in ws.items()}

    def step():
        out = {}
        for n, w in ws.items():
            x = xs[n].clone().requires_grad_(True)
            y = ops.conv_train(x, w, 1, cache, n)
            y.sum().backward()
            out[n] = y.detach().clone()
        return out

    def fresh():
        with torch.no_grad():
            return {n: ops.conv_nhwc(xs[n].permute(0, 2, 3, 1).contiguous(), ops.PackedConv(w)).permute(0, 3, 1, 2) for n, w in ws.items()}

    ops.USE_PACK_TABLE = table
    try:
        first = step()
        pcs = {n: cache.get((n, 'fwd', True), [w], None) for n, w in ws.items()}
        assert all(pc._wino for pc in pcs.values()), 'the training forward did not take K17 at this size'
        ptrs = {n: pc._wino.packed.data_ptr() for n, pc in pcs.items()}
        for n in ws:
            assert torch.equal(first[n], fresh()[n])
        for it in range(2):
            with torch.no_grad():
                for i, w in enumerate(ws.values()):
                    w.mul_(1.3 + i).add_(0.004)
                    w.grad = None
            got, exp = step(), fresh()
            for n in ws:
                assert torch.equal(got[n], exp[n]), (n, it)
                assert not torch.equal(got[n], first[n])
            assert all(pcs[n]._wino.packed.data_ptr() == ptrs[n] for n in ws)          # re-packed in place
    finally:
        ops.USE_PACK_TABLE = True


# ---------------------------------------------------------------------------------------------------------------------
# K17: Winograd F(2x2, 3x3) on split-fp16 operands (far_conv3x3_wino_f32) -- the same contract as K9's stride-1 3x3 mode
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('N,H,W,Cin,Cout', [
    (1, 16, 16, 16, 64),             # one tile block, one k-step
    (2, 24, 40, 32, 64),
    (3, 30, 37, 196, 196),           # ragged tile blocks, a partial last k-step (196 = 12 x 16 + 4), 4 channel blocks (256 > 196)
    (1, 17, 16, 128, 128),
    (2, 9, 50, 256, 196),
    (1, 5, 7, 196, 128),             # smaller than one tile block
    (1, 33, 18, 20, 4),              # Cin, Cout far below a block
])
def test_conv_winograd_matches_float64(N, H, W, Cin, Cout):
    """K17 against a float64 convolution, next to K9 on the same tensors: the go / no-go bar of the Winograd kernel was
    2e-6 of max |ref| (measured 2-4e-7: fewer accumulation steps per output than the direct kernel's 9 Cin)."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(N * 1000 + Cin + Cout)
    x = (torch.randn(N, H, W, Cin, device='cuda', generator=g) * 1.5).relu_()
    w = torch.randn(Cout, Cin, 3, 3, device='cuda', generator=g) * (2.0 / (Cin * 9)) ** 0.5
    scale = torch.rand(Cout, device='cuda', generator=g) + 0.5
    shift = torch.randn(Cout, device='cuda', generator=g) * 0.1
    res = torch.randn(N, H, W, Cout, device='cuda', generator=g)
    ref_bn = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1) * scale.double() + shift.double()
    pw, pc = ops.PackedWino(w, scale, shift), ops.PackedConv(w, scale, shift)
    ops.overflow_flag('cuda').zero_()
    for act, residual in (('none', None), ('relu', res), ('leaky', None)):
        y = ops.conv3x3_wino(x, pw, residual=residual, act=act, slope=0.01)
        r = ref_bn + (residual.double() if residual is not None else 0)
        r = {'none': lambda t: t, 'relu': torch.relu, 'leaky': lambda t: F.leaky_relu(t, 0.01)}[act](r)
        emax, erms = _rel(y, r)
        kmax, _ = _rel(ops.conv_nhwc(x, pc, residual=residual, act=act, slope=0.01), r)
        assert emax < 2e-6 and erms < 1e-6, (act, emax, erms)
        assert emax < 2.5 * kmax + 2e-7, (act, emax, kmax)          # never worse than the direct kernel by more than noise
    assert not ops.activation_overflowed('cuda')


def test_conv_winograd_is_deterministic_and_independent_of_the_batch():
    """Run-to-run bit equality over many launches (the request rings are waited for by counting: a stale LDS read would show
    as a difference in some workgroups of some launches), and image n of a batch == the same image alone."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(16, 64, 96, 128, device='cuda', generator=g).relu_()
    w = torch.randn(128, 128, 3, 3, device='cuda', generator=g) * 0.03
    pw = ops.PackedWino(w)
    y0 = ops.conv3x3_wino(x, pw, act='relu')
    for _ in range(20):
        assert torch.equal(ops.conv3x3_wino(x, pw, act='relu'), y0)
    for n in (0, 7, 15):
        assert torch.equal(ops.conv3x3_wino(x[n:n + 1].contiguous(), pw, act='relu')[0], y0[n])


def test_conv_winograd_activation_range_flag():
    """|a| <= 16376 survives K17's unscaled split (the transformed operand is at most 4 |a|); beyond it the launch raises the
    device flag like K9, and the entry point refuses an activation exponent below 0 (the host then uses K9)."""
    from far_amd import _lib
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(9)
    x = torch.randn(1, 16, 16, 32, device='cuda', generator=g)
    w = torch.randn(64, 32, 3, 3, device='cuda', generator=g) * 0.05
    pw = ops.PackedWino(w)
    ops.overflow_flag('cuda').zero_()
    x[0, 2:4, 2:4, 5] = 16000.0          # a whole 2 x 2 tile: one transformed operand is the sum of the four
    y = ops.conv3x3_wino(x, pw)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    assert not ops.activation_overflowed('cuda') and _rel(y, ref)[0] < 2e-6
    x[0, 2:4, 2:4, 5] = 17000.0          # 4 x 17000 > 65504: the fp16 hi part of that operand is inf
    ops.conv3x3_wino(x, pw)
    assert ops.activation_overflowed('cuda')
    with ops.activation_exponent(-4):
        y = ops.conv3x3_wino(x, pw)          # the wrapper clamps the exponent it passes to 0: same launch, flag again
    assert ops.activation_overflowed('cuda')
    with pytest.raises(_lib.FarHipError):
        ops.conv3x3_wino(x[..., :30].contiguous(), ops.PackedWino(w[:, :28].contiguous()))      # channel count mismatch


def test_conv_nhwc_dispatches_inference_3x3_layers_to_winograd_and_everything_else_to_k9():
    """Round 4: under no_grad, conv_nhwc runs a stride-1 3x3 split-precision layer on K17 (bit-identical to conv3x3_wino, and to
    float64 within the K9 bar); with gradients enabled, below activation exponent 0, with USE_WINO off, and for stride-2 / 1x1 /
    plain-fp16 / fused-input layers it stays on K9 (bit-identical to the K9 launch)."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(21)
    x = torch.randn(2, 40, 48, 64, device='cuda', generator=g).relu_()
    w = torch.randn(128, 64, 3, 3, device='cuda', generator=g) * 0.05
    sc, sh = torch.rand(128, device='cuda', generator=g) + 0.5, torch.randn(128, device='cuda', generator=g)
    res = torch.randn(2, 40, 48, 128, device='cuda', generator=g)
    pc = ops.PackedConv(w, sc, sh)
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1) * sc.double() + sh.double() + res.double())
    y9 = ops.conv_nhwc(x, pc, residual=res, act='relu')                      # gradients enabled: K9
    with torch.no_grad():
        assert ops.USE_WINO
        y17 = ops.conv_nhwc(x, pc, residual=res, act='relu')
        assert pc.wino() is not None
        assert torch.equal(y17, ops.conv3x3_wino(x, pc.wino(), residual=res, act='relu'))
        assert not torch.equal(y17, y9)                                      # two different kernels ...
        e17, e9 = _rel(y17, ref)[0], _rel(y9, ref)[0]
        print(f'[deviation] conv_nhwc 64->128 3x3: K17 {e17:.2e}, K9 {e9:.2e} of max|ref| vs float64')
        assert e17 < 2e-6 and e9 < 2e-6                                      # ... both at the float64 bar
        with ops.activation_exponent(-4):                                    # widened range: K9
            ya = ops.conv_nhwc(x, pc, residual=res, act='relu')
        with ops.activation_exponent(-4):
            yb = ops.conv_nhwc(x, pc, residual=res, act='relu')
        assert torch.equal(ya, yb) and _rel(ya, ref)[0] < 1e-4
        ops.USE_WINO = False
        try:
            assert torch.equal(ops.conv_nhwc(x, pc, residual=res, act='relu'), y9)
        finally:
            ops.USE_WINO = True
        # layers K17 does not serve keep their K9 image only
        assert ops.PackedConv(w, sc, sh, stride=2).wino() is None
        assert ops.PackedConv(w[:, :, :1, :1].contiguous(), sc, sh).wino() is None
        assert ops.PackedConv(w, sc, sh, split=False).wino() is None
        assert ops.PackedConv(w, dgrad=True).wino() is None
        small = x[:, :16, :16].contiguous()                                   # below WINO_MIN_PIXELS: K9
        ops.USE_WINO = False
        ys = ops.conv_nhwc(small, pc, act='relu')
        ops.USE_WINO = True
        assert torch.equal(ops.conv_nhwc(small, pc, act='relu'), ys)


def test_conv_winograd_beyond_2_gib():
    """K17 on an input of 2.75 GiB (72 images x 240 x 320 x 128 fp32: byte offsets beyond 2^31, element offsets beyond 2^29): the last
    image of the batch equals the same image convolved alone, bit for bit."""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(77)
    w = torch.randn(128, 128, 3, 3, device='cuda', generator=g) * 0.03
    pw = ops.PackedWino(w, torch.rand(128, device='cuda', generator=g) + 0.5, torch.randn(128, device='cuda', generator=g))
    x = torch.empty(72, 240, 320, 128, device='cuda')
    x.normal_(generator=g).relu_()
    y = ops.conv3x3_wino(x, pw, act='relu')
    for n in (0, 41, 71):
        assert torch.equal(ops.conv3x3_wino(x[n:n + 1].contiguous(), pw, act='relu')[0], y[n]), n
    ref = F.conv2d(x[71:72, :64, :64].permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)[0, 1:-1, 1:-1]
    got = ops.conv3x3_wino(x[71:72, :64, :64].contiguous(), ops.PackedWino(w))[0, 1:-1, 1:-1]
    assert _rel(got, ref)[0] < 2e-6


def test_conv_winograd_random_shapes_sweep():
    """K17 over 30 random layer shapes (ragged 16 x 16 workgroup tiles, channel counts that are multiples of 4 only, last channel
    blocks of every width incl. the HALF body's <= 32, single-pixel rows / columns): 2e-6 of max|ref| vs float64, with and without
    residual, and bit-identical to the same layer on K9's contract (conv_nhwc under no_grad dispatches here)."""
    ops = _ops()
    rng = torch.Generator().manual_seed(2024)
    g = torch.Generator(device='cuda').manual_seed(2024)
    worst = 0.0
    for it in range(30):
        N = int(torch.randint(1, 4, (1,), generator=rng))
        H = int(torch.randint(1, 70, (1,), generator=rng))
        W = int(torch.randint(1, 70, (1,), generator=rng))
        ci = 4 * int(torch.randint(1, 70, (1,), generator=rng))
        co = 4 * int(torch.randint(1, 70, (1,), generator=rng))
        x = torch.randn(N, H, W, ci, device='cuda', generator=g)
        w = torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (ci * 9)) ** 0.5
        sc, sh = torch.rand(co, device='cuda', generator=g) + 0.5, torch.randn(co, device='cuda', generator=g) * 0.1
        res = torch.randn(N, H, W, co, device='cuda', generator=g) if it % 2 else None
        act = ('none', 'relu', 'leaky')[it % 3]
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1) * sc.double() + sh.double()
        if res is not None:
            ref = ref + res.double()
        ref = {'none': lambda t: t, 'relu': torch.relu, 'leaky': lambda t: F.leaky_relu(t, 0.01)}[act](ref)
        y = ops.conv3x3_wino(x, ops.PackedWino(w, sc, sh), residual=res, act=act, slope=0.01)
        e = _rel(y, ref)[0]
        worst = max(worst, e)
        assert e < 2e-6, (it, N, H, W, ci, co, act, e)
        assert torch.isfinite(y).all()
    print(f'[deviation] K17 random sweep: worst {worst:.2e} of max|ref| over 30 shapes')
    assert not ops.activation_overflowed('cuda')

"""One autograd node per LoFTR encoder layer for training on the GPU.

`LoFTREncoderLayer.forward` under autograd (mp3d_loftr/src/loftr/loftr_module/transformer.py:44-67) is five Linear layers, the
linear-attention core, two LayerNorms, a ReLU, a concatenation and a residual add.  As separate autograd nodes that is ~12
Python-level nodes per layer call, each with its own engine hop, plus the elementwise kernels autograd inserts between them
(gradient accumulation adds, the split of the concatenation's gradient, the ReLU mask) -- and the training step at batch 1 is
bound by that host work, not by the kernels.  Here the layer is ONE node: the forward launches the same
kernels the per-op path launches (K9, K5, K6), the backward launches their backward kernels in sequence (K6 backward, K9 dgrad,
K16 wgrad, K5 backward) and folds what was elementwise glue into them:
  * the residual path and the three input-gradient contributions of q / k / v projections accumulate through K9's `residual`
    epilogue input (no add kernels);
  * the gradient of cat([x, message]) leaves the mlp[0] dgrad launch as two output planes (no split copies);
  * every gradient tensor gets its power-of-two scale once (far_grad_scale_f32) for both its dgrad and its wgrad launch.
Same arithmetic as the per-op path kernel for kernel; `LoFTREncoderLayer.layer_node = False` selects that path (the tests
compare the two).
"""
import ctypes

import torch

from .. import ops


class _EncoderLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, source, wq, wk, wv, wm, w0, w2, g1, b1, g2, b2, layer):
        self_attn = source is None
        xc = x.detach().float().contiguous()
        sc = xc if self_attn else source.detach().float().contiguous()
        pk = layer.__dict__.setdefault('_packs', ops.PackCache())
        sp = layer.split_operands
        P = lambda w, name: ops.train_pack(pk, ('train', name), w, None, sp)
        nh = layer.nhead
        # the three projections are independent and each fills a fraction of the CUs: k and v on side streams
        Pq, Pk, Pv = P(wq, 'q'), P(wk, 'k'), P(wv, 'v')
        if layer.overlap:
            with ops.side(0):
                k = ops.linear_f16s(sc, Pk)
            with ops.side(1):
                v = ops.linear_f16s(sc, Pv)
            q = ops.linear_f16s(xc, Pq)
            ops.join(0)
            ops.join(1)
        else:
            q, k, v = ops.linear_f16s(xc, Pq), ops.linear_f16s(sc, Pk), ops.linear_f16s(sc, Pv)
        msg0 = ops.linear_attention(q, k, v, nh, None, None, layer.attention.eps)
        m1 = ops.linear_f16s(msg0, P(wm, 'merge'))
        n1 = ops.layernorm(m1, g1.detach(), b1.detach(), layer.norm1.eps)
        xcat = torch.cat([xc, n1], dim=-1)
        h = ops.linear_f16s(xcat, P(w0, 'mlp0'), act='relu')
        m2 = ops.linear_f16s(h, P(w2, 'mlp2'))
        y = ops.layernorm(m2, g2.detach(), b2.detach(), layer.norm2.eps, residual=xc)
        ctx.save_for_backward(xc, sc, q, k, v, msg0, m1, xcat, h, m2, wq, wk, wv, wm, w0, w2, g1, g2)
        ctx.layer, ctx.self_attn, ctx.act_exp = layer, self_attn, ops.activation_exponent_value()
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, sc, q, k, v, msg0, m1, xcat, h, m2, wq, wk, wv, wm, w0, w2, g1, g2 = ctx.saved_tensors
        layer, ae = ctx.layer, ctx.act_exp
        pk = layer.__dict__.setdefault('_packs', ops.PackCache())
        sp = layer.split_operands
        PT = lambda w, name: ops.train_pack_t(pk, ('train', name), w, None, sp)
        C = xc.shape[-1]
        flat = lambda t: t.reshape(-1, t.shape[-1])
        gy = gy.float().contiguous()
        # Every weight gradient (K16 + its reduction: ~25 us on a fraction of the CUs) goes to side stream 0 and overlaps with the
        # chain of input-gradient launches on the main stream; `keep` holds what the side launches touch until the join.
        keep = []
        ov = layer.overlap

        def wgrad(xin, dy, s):
            if not ov:
                return ops.linear_wgrad(flat(xin), flat(dy), s, ae)
            with ops.side(0):
                return ops.linear_wgrad(flat(xin), flat(dy), s, ae, keep)
        # norm2 (+ residual: its gradient is gy itself) and mlp[2]
        dm2, dg2, db2 = ops.layernorm_bwd(m2, g2, gy, layer.norm2.eps)
        s2 = ops.grad_scale(dm2)
        dw2 = wgrad(h, dm2, s2)
        dh = ops.linear_f16s(dm2, PT(w2, 'mlp2'), act_scale_dev=s2)
        # ReLU, mlp[0]: the gradient of cat([x, norm1(..)]) as two planes
        dh = torch.ops.aten.threshold_backward(dh, h, 0.0)
        s0 = ops.grad_scale(dh)
        dw0 = wgrad(xcat, dh, s0)
        dcat = ops.linear_f16s(dh, PT(w0, 'mlp0'), out_planes=2, act_scale_dev=s0)
        dxa = gy + dcat[0]
        # norm1, merge
        dm1, dg1, db1 = ops.layernorm_bwd(m1, g1, dcat[1], layer.norm1.eps)
        sm = ops.grad_scale(dm1)
        dwm = wgrad(msg0, dm1, sm)
        dmsg = ops.linear_f16s(dm1, PT(wm, 'merge'), act_scale_dev=sm)
        # attention core
        dq, dk, dv = ops.linear_attention_bwd(q, k, v, dmsg, layer.nhead, None, None, layer.attention.eps)
        # the three projections: input gradients accumulate through the dgrad launches' residual input
        sq, sk, sv = ops.grad_scale(dq), ops.grad_scale(dk), ops.grad_scale(dv)
        dwq, dwk, dwv = wgrad(xc, dq, sq), wgrad(sc, dk, sk), wgrad(sc, dv, sv)
        dx = ops.linear_f16s(dq, PT(wq, 'q'), residual=dxa, act_scale_dev=sq)
        if ctx.self_attn:
            dx = ops.linear_f16s(dk, PT(wk, 'k'), residual=dx, act_scale_dev=sk)
            dx = ops.linear_f16s(dv, PT(wv, 'v'), residual=dx, act_scale_dev=sv)
            ds = None
        else:
            ds = ops.linear_f16s(dk, PT(wk, 'k'), act_scale_dev=sk)
            ds = ops.linear_f16s(dv, PT(wv, 'v'), residual=ds, act_scale_dev=sv)
        if ov:
            ops.join(0)
        del keep
        return dx, ds, dwq, dwk, dwv, dwm, dw0, dw2, dg1, db1, dg2, db2, None


class _EncoderLayerNativeFn(torch.autograd.Function):
    """The same node with its launch sequences issued by the library (far_enc_layer_fwd / far_enc_layer_bwd,
    csrc/encoder_layer_train.hip): two calls per layer instead of ~50 launches' worth of Python."""

    @staticmethod
    def _desc(layer, ws_, bs, L, S, C, self_attn, act_exp):
        lib = ops._lib.load()
        pk = layer.__dict__.setdefault('_packs', ops.PackCache())
        sp = layer.split_operands
        d = ops._lib.EncLayer()
        d.bs, d.L, d.S, d.C, d.nhead, d.self_attn, d.split = bs, L, S, C, layer.nhead, int(self_attn), int(sp)
        d.act_exp, d.overlap = act_exp, int(layer.overlap)
        d.eps1, d.eps2, d.attn_eps = layer.norm1.eps, layer.norm2.eps, layer.attention.eps
        keep = []
        for i, (w, name) in enumerate(zip(ws_, ('q', 'k', 'v', 'merge', 'mlp0', 'mlp2'))):
            f = ops.train_pack(pk, ('train', name), w, None, sp)
            t = ops.train_pack_t(pk, ('train', name), w, None, sp)
            d.img[i], d.img_scale[i], d.imgT[i], d.imgT_scale[i] = f.packed.data_ptr(), f.scale.data_ptr(), t.packed.data_ptr(), t.scale.data_ptr()
            keep += [f, t]
        d.overflow = ops.overflow_flag(ws_[0].device).data_ptr()
        return d, keep, lib

    @staticmethod
    def forward(ctx, x, source, wq, wk, wv, wm, w0, w2, g1, b1, g2, b2, layer):
        self_attn = source is None
        xc = x.detach().float().contiguous()
        sc = xc if self_attn else source.detach().float().contiguous()
        bs, L, C = xc.shape
        ae = ops.activation_exponent_value()
        d, keep, lib = _EncoderLayerNativeFn._desc(layer, (wq, wk, wv, wm, w0, w2), bs, L, sc.shape[1], C, self_attn, ae)
        d.g1, d.b1, d.g2, d.b2 = g1.data_ptr(), b1.data_ptr(), g2.data_ptr(), b2.data_ptr()
        dref = ctypes.byref(d)
        saved = torch.empty(int(lib.far_enc_layer_saved_floats(dref)), dtype=torch.float32, device=xc.device)
        nws = int(lib.far_enc_layer_fwd_ws_bytes(dref))
        if saved.numel() == 0 or nws == 0:
            raise ops._lib.FarHipError('far_enc_layer_fwd: layer shape not covered')
        ws = torch.empty(nws, dtype=torch.uint8, device=xc.device)
        y = torch.empty_like(xc)
        rc = lib.far_enc_layer_fwd(dref, xc.data_ptr(), None if self_attn else sc.data_ptr(), saved.data_ptr(), y.data_ptr(), ws.data_ptr(), nws,
                                   ops._stream())
        ops._lib.check(rc, 'far_enc_layer_fwd')
        ctx.save_for_backward(xc, sc, saved, wq, wk, wv, wm, w0, w2, g1, b1, g2, b2)
        ctx.layer, ctx.self_attn, ctx.act_exp = layer, self_attn, ae
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, sc, saved, wq, wk, wv, wm, w0, w2, g1, b1, g2, b2 = ctx.saved_tensors
        layer = ctx.layer
        bs, L, C = xc.shape
        d, keep, lib = _EncoderLayerNativeFn._desc(layer, (wq, wk, wv, wm, w0, w2), bs, L, sc.shape[1], C, ctx.self_attn, ctx.act_exp)
        d.g1, d.b1, d.g2, d.b2 = g1.data_ptr(), b1.data_ptr(), g2.data_ptr(), b2.data_ptr()
        dref = ctypes.byref(d)
        gy = gy.float().contiguous()
        grads = torch.empty(int(lib.far_enc_layer_grads_floats(dref)), dtype=torch.float32, device=xc.device)
        nws = int(lib.far_enc_layer_bwd_ws_bytes(dref))
        ws = torch.empty(nws, dtype=torch.uint8, device=xc.device)
        rc = lib.far_enc_layer_bwd(dref, xc.data_ptr(), None if ctx.self_attn else sc.data_ptr(), saved.data_ptr(), gy.data_ptr(), grads.data_ptr(),
                                   ws.data_ptr(), nws, ops._stream())
        ops._lib.check(rc, 'far_enc_layer_bwd')
        off = (ctypes.c_long * 12)()
        ops._lib.check(lib.far_enc_layer_grads_offsets(dref, off), 'far_enc_layer_grads_offsets')
        R, Rs = bs * L, bs * sc.shape[1]
        piece = lambda i, n, shape: grads.narrow(0, off[i], n).view(shape)
        dx = piece(0, R * C, xc.shape)
        ds = None if ctx.self_attn else piece(1, Rs * C, sc.shape)
        dws = [piece(2 + i, w.numel(), w.shape) for i, w in enumerate((wq, wk, wv, wm, w0, w2))]
        dgb = [piece(8 + i, C, (C,)) for i in range(4)]
        return (dx, ds, *dws, *dgb, None)


def encoder_layer_train(layer, x, source):
    """LoFTREncoderLayer.forward(x, source) (no masks) with gradients as one autograd node; `source is x` = self-attention."""
    if any(m.bias is not None for m in (layer.q_proj, layer.k_proj, layer.v_proj, layer.merge, layer.mlp[0], layer.mlp[2])):
        raise ops._lib.FarHipError('encoder_layer_train: the FAR encoder layers have bias-free Linear layers')
    fn = _EncoderLayerNativeFn if layer.native_node else _EncoderLayerFn
    return fn.apply(x, None if source is x else source, layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight,
                                 layer.merge.weight, layer.mlp[0].weight, layer.mlp[2].weight, layer.norm1.weight, layer.norm1.bias,
                                 layer.norm2.weight, layer.norm2.bias, layer)

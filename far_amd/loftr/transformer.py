"""LoFTR encoder stack and the EMM regression head, on the far_amd kernels.

Mirrors the module/parameter structure of mp3d_loftr/src/loftr/loftr_module/transformer.py so that reference
checkpoints load unchanged:
  LoFTREncoderLayer :12-67, LocalFeatureTransformer :69-112, get_positional_encodings :183-248,
  CrossAttention :250-303, CrossBlock :305-348, LocalFeatureTransformerRegressor :350-499
and of mp3d_loftr/src/loftr/loftr_module/vit_layers/mlp.py:8-28 (Mlp).
The attention cores run in libfar_hip.so: K5 (linear attention) and K2 (bilinear dual-softmax, never
materialising the (B, 4, 4800, 4800) score tensors), K6 (LayerNorm) and K9 / K15 (every Linear layer of the inference path).

Batch semantics: the reference head is batch-size-1 only (its pairing reshape :339-341 and the gate
broadcasts :466-469 break for B > 1).  Here B pairs are processed as B
independent B = 1 problems stacked along dim 0; for B = 1 every tensor has the reference's shape.
"""
import copy
import os
from functools import partial

import torch
import torch.nn as nn

from .. import _vendor as ag          # needs_grad + the door to the test-side vendor compositions (far_amd/_vendor.py)
from .. import flags, ops
from ..pose6d import pose_mean_6d, pose_std_6d


class LinearAttention(nn.Module):
    """Module-shaped handle on K5 (reference: linear_attention.py:12-52)."""

    def __init__(self, eps=1e-6, use_num_corres=False):
        super().__init__()
        self.eps = eps
        self.use_num_corres = use_num_corres

    def forward(self, queries, keys, values, q_mask=None, kv_mask=None, loftr_preds=None):
        N, L, H, D = queries.shape
        S = keys.shape[1]
        if ag.needs_grad(queries, keys, values):
            if queries.is_cuda and D in (16, 32):     # training on the GPU: K5 forward + K5 backward kernels
                return ops.linear_attention_train(queries.reshape(N, L, H * D), keys.reshape(N, S, H * D),
                                                  values.reshape(N, S, H * D), H, q_mask, kv_mask, self.eps).view(N, L, H, D)
            vend = ag.require('LinearAttention under autograd on CPU tensors / head dims without a K5 backward')
            return vend.linear_attention(queries.reshape(N, L, H * D), keys.reshape(N, S, H * D),
                                         values.reshape(N, S, H * D), H, q_mask, kv_mask, self.eps).view(N, L, H, D)
        as_u8 = lambda m: None if m is None else m.to(torch.uint8).contiguous()
        out = ops.linear_attention(queries.reshape(N, L, H * D), keys.reshape(N, S, H * D),
                                   values.reshape(N, S, H * D), H, as_u8(q_mask), as_u8(kv_mask), self.eps)
        return out.view(N, L, H, D)


class LoFTREncoderLayer(nn.Module):
    split_operands = True        # K9 operand precision of the Linear layers (False: plain fp16, LoFTR.set_precision)
    dense_split = True           # ... of the launches that are plain Linear layers at d_model 256 (merge, mlp[0], mlp[2]): False =
                                 # plain fp16 there while the fused k|v-state / q-apply launches stay split ('mixed16')
    hip_training = True          # training on the GPU runs K9 / K5 (forward + backward kernels); False: vendor ops + autograd
    layer_node = True            # ... as one autograd node per layer call (layer_train.py); False: one node per operator
    overlap = True               # layer node: weight gradients and the k / v projections on the library's side streams
    native_node = True           # layer node: launch sequences issued by the library (far_enc_layer_fwd / _bwd) instead of Python
    fused_mlp = True             # d_model 128, split operands: the MLP block as one K13 launch (False: two K9 launches)
    fused_attn = True            # d_model 128, sequences <= 32 tokens, no masks: the attention block as one K14 launch
    fused_apply = not flags.off('FAR_NO_QAPPLY')   # ... and the q projection + LinearAttention's second half in one launch
    fused_kv = not flags.off('FAR_NO_KV')    # d_model 256, 8 heads, no masks: k | v projection + K'^T V in one launch (k, v never stored)

    def __init__(self, d_model, nhead, attention='linear', use_num_corres=False):
        super().__init__()
        if attention != 'linear':
            raise NotImplementedError("only attention='linear' (the FAR configuration) has a kernel")
        self.dim = d_model // nhead
        self.nhead = nhead
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.attention = LinearAttention(use_num_corres=use_num_corres)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(
            nn.Linear(d_model * 2, d_model * 2, bias=False),
            nn.ReLU(True),
            nn.Linear(d_model * 2, d_model, bias=False),
        )
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, x, source, x_mask=None, source_mask=None, loftr_preds=None, out=None):
        bs = x.size(0)
        if ag.needs_grad(x, source, self.norm1.weight) or not x.is_cuda:
            if (x.is_cuda and self.hip_training and self.layer_node and x_mask is None and source_mask is None and x.numel()
                    and source.numel() and self.dim in (16, 32) and x.shape[-1] % 4 == 0 and x.shape[-1] <= 512 and torch.is_grad_enabled()):
                # the whole layer as ONE autograd node (layer_train.py): same kernels, no elementwise glue between them
                from .layer_train import encoder_layer_train
                return encoder_layer_train(self, x, source)
            if x.is_cuda and self.hip_training:
                # training on the GPU: the five Linear layers on K9 (forward and dgrad) and K16 (wgrad), the attention core on
                # K5 forward + backward, both LayerNorms (the second with the residual add) on K6 forward + backward
                pk = self.__dict__.setdefault('_packs', ops.PackCache())
                sp = self.split_operands
                lin = lambda t, mod, name: ops.linear_train(t, mod.weight, mod.bias, pk, ('train', name), split=sp)
                q = lin(x, self.q_proj, 'q').view(bs, -1, self.nhead, self.dim)
                k = lin(source, self.k_proj, 'k').view(bs, -1, self.nhead, self.dim)
                v = lin(source, self.v_proj, 'v').view(bs, -1, self.nhead, self.dim)
                msg = self.attention(q, k, v, q_mask=x_mask, kv_mask=source_mask, loftr_preds=loftr_preds)
                msg = ops.layernorm_train(lin(msg.view(bs, -1, self.nhead * self.dim), self.merge, 'merge'), self.norm1)
                msg = lin(torch.relu(lin(torch.cat([x, msg], dim=2), self.mlp[0], 'mlp0')), self.mlp[2], 'mlp2')
                return ops.layernorm_train(msg, self.norm2, residual=x)          # x + norm2(msg): the add in K6's epilogue
            # CPU tensors / hip_training = False: the reference-style module composition lives with the tests (tests/vendor_ops.py)
            return ag.require('LoFTREncoderLayer on CPU tensors or with hip_training = False').encoder_layer(
                self, x, source, x_mask, source_mask, loftr_preds)
        # inference: the five Linear layers on K9 (split-fp16 operands: fp32-grade, and -- unlike a vendor GEMM whose
        # kernel is chosen by the row count -- every output row depends on its input row only)
        pk = self.__dict__.setdefault('_packs', ops.PackCache())
        sp = self.split_operands
        lin = lambda name, *mods, sp=sp: pk.get((name, sp), [m.weight for m in mods],
                                                lambda: ops.PackedConv(torch.cat([m.weight for m in mods], 0), split=sp))
        spd = sp and (self.dense_split or x.shape[-1] != 256)
        x = x.contiguous()
        heads = lambda t: t.view(bs, -1, self.nhead, self.dim)
        fuse = True     # measured: q | k | v (and k | v) in one launch pays at d_model 256 and, with 128-channel blocks, at 128
        source = source.contiguous()
        if (self.fused_attn and x.shape[-1] == 128 and self.nhead == 8 and x.shape[1] <= 32 and source.shape[1] <= 32
                and x_mask is None and source_mask is None):
            # d_model 128 on short sequences (the fine-level windows: bandwidth-bound): the whole layer in two launches --
            # K14 (q / k / v projections, linear attention, merge, norm1) and K13 (the MLP block, norm2, residual)
            pa = pk.get(('attn-fused',), [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.merge.weight],
                        lambda: ops.PackedAttn(self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.merge.weight))
            msg = ops.attn_block(x, source, pa, self.nhead, self.norm1.weight, self.norm1.bias, self.norm1.eps, self.attention.eps,
                                 plain16=not sp)
            if self.fused_mlp:
                pm = pk.get(('mlp-fused',), [self.mlp[0].weight, self.mlp[2].weight],
                            lambda: ops.PackedMlp(self.mlp[0].weight, self.mlp[2].weight))
                return ops.mlp_fused(x, msg, pm, self.norm2.weight, self.norm2.bias, self.norm2.eps, out=out, plain16=not sp)
            h = ops.linear_f16s(x, lin('mlp0', self.mlp[0]), act='relu', x2=msg)
            return ops.linear_f16s(h, lin('mlp2', self.mlp[2]), ln=(self.norm2.weight, self.norm2.bias, self.norm2.eps),
                                   post_residual=x, out=out)
        if (self.fused_kv and x.shape[-1] == 256 and self.nhead == 8 and source.shape[1] >= 64 and x_mask is None
                and source_mask is None):
            # d_model 256 (the coarse level, the head's layers): the k | v projection ends in K'^T V (linear_attention.py:38-45)
            # instead of a store -- k and v (4 of the layer's 19 passes over a (rows, 256) tensor) never exist.  Partial sums are
            # per 64-token block of an image (lengths that are no multiple of 64 run on a padded launch geometry): an image's bits
            # do not depend on the rest of the batch
            pkv = pk.get(('kv-state', sp), [self.k_proj.weight, self.v_proj.weight],
                         lambda: ops.PackedConv(ops.kv_interleaved_weight(self.k_proj.weight, self.v_proj.weight, self.nhead), split=sp))
            S = source.shape[1]
            if self.fused_apply and x.shape[1] >= 64:
                # ... and the q projection ends in (Q' KV) Z S (:46-50): q is never stored either, K5's apply launch disappears
                _, image = ops.linear_kv_state(source, pkv, S, want_image=True)
                msg = ops.linear_q_apply(x, lin('q', self.q_proj), image, S, eps=self.attention.eps)
            else:
                kv = ops.linear_kv_state(source, pkv, S)
                q = ops.linear_f16s(x, lin('q', self.q_proj))
                msg = ops.linear_attention_apply(q, kv, self.nhead, S, eps=self.attention.eps)
        else:
            if fuse and source is x:  # self attention: q | k | v of the one input in a single launch, three output tensors
                q, k, v = ops.linear_f16s(x, lin('qkv', self.q_proj, self.k_proj, self.v_proj), out_planes=3)
            elif fuse:
                q = ops.linear_f16s(x, lin('q', self.q_proj))
                k, v = ops.linear_f16s(source, lin('kv', self.k_proj, self.v_proj), out_planes=2)
            else:
                q = ops.linear_f16s(x, lin('q', self.q_proj))
                k = ops.linear_f16s(source, lin('k', self.k_proj))
                v = ops.linear_f16s(source, lin('v', self.v_proj))
            q, k, v = heads(q), heads(k), heads(v)
            msg = self.attention(q, k, v, q_mask=x_mask, kv_mask=source_mask, loftr_preds=loftr_preds)
        # merge + norm1 (:60-61) in one launch: the LayerNorm runs in the Linear layer's epilogue
        msg = ops.linear_f16s(msg.view(bs, -1, self.nhead * self.dim), lin('merge', self.merge, sp=spd),
                              ln=(self.norm1.weight, self.norm1.bias, self.norm1.eps))
        if self.fused_mlp and x.shape[-1] == 128:
            # d_model 128 (the fine-level windows: bandwidth-bound): mlp[0] + ReLU + mlp[2] + norm2 + residual (:64-67) in
            # ONE launch (K13), the 256-channel hidden tensor stays in the accumulator registers
            pm = pk.get(('mlp-fused',), [self.mlp[0].weight, self.mlp[2].weight],
                        lambda: ops.PackedMlp(self.mlp[0].weight, self.mlp[2].weight))
            return ops.mlp_fused(x, msg.contiguous(), pm, self.norm2.weight, self.norm2.bias, self.norm2.eps, out=out, plain16=not sp)
        # mlp[0](cat[x, msg]) reads both inputs in place (:64), ReLU in the epilogue
        h = ops.linear_f16s(x, lin('mlp0', self.mlp[0], sp=spd), act='relu', x2=msg)
        # mlp[2] + norm2 + the residual `x + message` (:65-67) in one launch
        return ops.linear_f16s(h, lin('mlp2', self.mlp[2], sp=spd), ln=(self.norm2.weight, self.norm2.bias, self.norm2.eps),
                               post_residual=x, out=out)


def _halves_of_one_buffer(a, b):
    """a and b are contiguous, equally shaped, in the same storage, b right behind a (cat([a, b], 0) is then a free view)."""
    return (a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and
            a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and b.storage_offset() == a.storage_offset() + a.numel())


class LocalFeatureTransformer(nn.Module):
    stack_self = not flags.off('FAR_NO_STACK')            # on the GPU: the two self-attention calls of a layer as one call on both images

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.d_model = config['d_model']
        self.nhead = config['nhead']
        self.layer_names = config['layer_names']
        if 'regress_use_num_corres' not in config:
            config['regress_use_num_corres'] = False
        layer = LoFTREncoderLayer(config['d_model'], config['nhead'], config['attention'],
                                  config['regress_use_num_corres'])
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(len(self.layer_names))])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, feat0, feat1, mask0=None, mask1=None, loftr_preds=None, inv_loftr_preds=None, joint_out=False):
        """joint_out (inference): the last layer writes feat0 / feat1 into the two halves of one (2N, L, C) buffer,
        so that a consumer of cat([feat0, feat1], 0) (the EMM head, transformer.py:488) gets it without a copy."""
        assert self.d_model == feat0.size(2), "the feature number of src and transformer must be equal"
        joint_out = joint_out and feat0.is_cuda and feat0.shape == feat1.shape and not ag.needs_grad(feat0, feat1)
        n = feat0.size(0)
        last = len(self.layers) - 1
        for li, (layer, name) in enumerate(zip(self.layers, self.layer_names)):
            o0 = o1 = None
            # inference: every layer writes feat0 / feat1 into the two halves of one buffer, so that the next 'self' layer can
            # run ONCE on both images (rows are independent, the attention core works per image: identical results, half the
            # launches, fewer partial rounds of workgroups per launch)
            infer_joint = (self.stack_self and feat0.is_cuda and feat0.dtype == torch.float32 and feat0.shape == feat1.shape
                           and mask0 is None and mask1 is None and not ag.needs_grad(feat0, feat1, layer.norm1.weight))
            if (joint_out and li == last) or infer_joint:
                buf = torch.empty(2 * n, *feat0.shape[1:], dtype=torch.float32, device=feat0.device)
                o0, o1 = buf[:n], buf[n:]
            kw0 = {} if o0 is None else {'out': o0}
            kw1 = {} if o1 is None else {'out': o1}
            if name == 'self' and infer_joint and _halves_of_one_buffer(feat0, feat1):
                x01 = torch.as_strided(feat0, (2 * n,) + tuple(feat0.shape[1:]), feat0.stride())
                layer(x01, x01, None, None, None, out=buf)
                feat0, feat1 = o0, o1
            elif (name == 'self' and self.stack_self and feat0.is_cuda and feat0.shape == feat1.shape and mask0 is None and mask1 is None
                    and layer.hip_training and layer.layer_node and torch.is_grad_enabled() and ag.needs_grad(feat0, feat1, layer.norm1.weight)):
                # training: the two self-attention calls of a layer are independent -- one call on both images (at batch 1 a
                # launch fills a fraction of the CUs; rows are independent, the attention core works per image); unbind's
                # backward is one stack, where two slices' would be two zero-fills, two copies and an add
                x01 = torch.cat([feat0, feat1], 0)
                y01 = layer(x01, x01)
                feat0, feat1 = y01.view(2, n, *feat0.shape[1:]).unbind(0)
            elif name == 'self':
                feat0 = layer(feat0, feat0, mask0, mask0, inv_loftr_preds, **kw0)
                feat1 = layer(feat1, feat1, mask1, mask1, loftr_preds, **kw1)
            elif name == 'cross':
                feat0 = layer(feat0, feat1, mask0, mask1, inv_loftr_preds, **kw0)
                feat1 = layer(feat1, feat0, mask1, mask0, loftr_preds, **kw1)  # uses the UPDATED feat0 (:107-108)
            else:
                raise KeyError(name)
        return feat0, feat1


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x, residual=None):
        """fc2(act(fc1(x))) (+ residual).  Inference on the GPU: both Linear layers on K9 (every output row depends on its
        input row only), the activation as an elementwise torch op in between, the residual in fc2's epilogue."""
        if not x.is_cuda:
            ag.require('Mlp on CPU tensors')
        if ag.needs_grad(x, self.fc1.weight) or not x.is_cuda:      # GPU training: autograd through the two parameter containers
            y = self.fc2(self.act(self.fc1(x)))
            return y if residual is None else residual + y
        pk = self.__dict__.setdefault('_packs', ops.PackCache())
        p1 = pk.get('fc1', [self.fc1.weight, self.fc1.bias], lambda: ops.PackedConv(self.fc1.weight, None, self.fc1.bias))
        p2 = pk.get('fc2', [self.fc2.weight, self.fc2.bias], lambda: ops.PackedConv(self.fc2.weight, None, self.fc2.bias))
        hid = self.act(ops.linear_f16s(x.contiguous(), p1))
        return ops.linear_f16s(hid, p2, residual=None if residual is None else residual.contiguous())


def positional_table(h=60, w=80):
    """(h*w, 6) fp32 table [y^2, x^2, xy, y, x, 1] of K^-1-normalised cell coordinates.

    Same arithmetic as get_positional_encodings (transformer.py:183-248) with its hard-coded intrinsics
    (:194-196), minus the 4800-iteration Python loop (:236-240): evaluated once, vectorised."""
    fx, fy, cx, cy = (torch.tensor(v, dtype=torch.float32) for v in (517 / 9, 517 / 8, 40., 30.))
    hpix, wpix = cy * 2, cx * 2
    K = torch.zeros(3, 3)
    K[0, 0] = (fx / wpix) * 2
    K[1, 1] = (fy / hpix) * 2
    K[0, 2] = (cx / wpix) * 2 - 1
    K[1, 2] = (cy / hpix) * 2 - 1
    K[2, 2] = 1
    Kinv = torch.inverse(K)
    ys = torch.linspace(-1, 1, steps=h)
    xs = torch.linspace(-1, 1, steps=w)
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')                # n = j*w + k  ->  (ys[j], xs[k])
    pts = torch.stack([gx.reshape(-1), gy.reshape(-1), torch.ones(h * w)], 0)   # (3, h*w)
    wv = Kinv @ pts
    p4 = wv[0] / wv[2]
    p3 = wv[1] / wv[2]
    return torch.stack([p3 * p3, p4 * p4, p3 * p4, p3, p4, torch.ones_like(p3)], dim=1).contiguous()


def positional_table_vit(h=24, w=24, intrinsics=None):
    """(h*w, 6) fp32 table [p3^2, p4^2, p3 p4, p3, p4, 1] of the 8-Point-ViT's CrossAttention
    (interiornetStreetlearn_8ptVit/src/modules/vision_transformer.py:90-158): cell (j, k) -- row j, column k -- sits at index
    n = k*w + j there (:150-151; mp3d's table uses j*w + k), and the intrinsics (fx, fy, cx, cy) on the feature grid are an
    argument instead of constants; None = the plain linspace coordinates (:109-110).  h == w as in the reference (24 x 24)."""
    if h != w:
        raise ValueError('the reference indexes this table with k*w + j, which covers the grid only when h == w (24 x 24)')
    ys = torch.linspace(-1, 1, steps=h)
    xs = torch.linspace(-1, 1, steps=w)
    if intrinsics is None:
        p3 = ys.repeat(w)                                          # :109  p3[n] = ys[n % h]
        p4 = xs.repeat_interleave(h)                               # :110  p4[n] = xs[n // h]
    else:
        fx, fy, cx, cy = (torch.tensor(float(v), dtype=torch.float32) for v in intrinsics)
        hpix, wpix = cy * 2, cx * 2
        K = torch.zeros(3, 3)
        K[0, 0] = (fx / wpix) * 2
        K[1, 1] = (fy / hpix) * 2
        K[0, 2] = (cx / wpix) * 2 - 1
        K[1, 2] = (cy / hpix) * 2 - 1
        K[2, 2] = 1
        Kinv = torch.inverse(K)
        gx, gy = torch.meshgrid(xs, ys, indexing='ij')            # n = k*w + j  ->  (xs[k], ys[j])
        pts = torch.stack([gx.reshape(-1), gy.reshape(-1), torch.ones(h * w)], 0)
        wv = Kinv @ pts
        p4 = wv[0] / wv[2]
        p3 = wv[1] / wv[2]
    return torch.stack([p3 * p3, p4 * p4, p3 * p4, p3, p4, torch.ones_like(p3)], dim=1).contiguous()


class CrossAttention(nn.Module):
    hip_training = True          # training on the GPU runs K2's forward + backward kernels; False: vendor ops + autograd
    plain16 = False              # inference: K2 on plain fp16 operands (far_emm_pv_f16; LoFTR.set_precision('mixed16'))
    exact_f32 = False            # inference: K2 on the exact-f32 MFMA kernels (no operand range limit; LoFTR sets it after an
                                 # activation-range overflow of the split-fp16 variant, model.py:_widen_activation_range)

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0., pos=None):
        """pos: the (N, 6) positional table appended to v (default: mp3d's 60 x 80 table with its hard-coded intrinsics,
        transformer.py:194-196; the 8-Point-ViT form -- dim 192, 3 heads, N = 576 -- passes positional_table_vit(24, 24, intrinsics))."""
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj_fundamental = nn.Linear(dim + 6 * num_heads, dim)
        self.register_buffer('pos6', positional_table() if pos is None else pos.detach().float().contiguous(), persistent=False)

    def forward(self, x1, x2, intrinsics=None, loftr_preds=None, inv_loftr_preds=None):
        B, N, C = x1.shape
        if not x1.is_cuda:
            ag.require('CrossAttention on CPU tensors')
        h, d = self.num_heads, C // self.num_heads
        if self.pos6.shape[0] != N:
            raise ValueError(f'the positional table of this CrossAttention has {self.pos6.shape[0]} rows (mp3d: the 60x80 coarse grid, '
                             f'transformer.py:194), got N={N} tokens')
        if ag.needs_grad(x1, x2, self.qkv.weight) or not x1.is_cuda:
            # (2, B, N, 3, h, d) -> (3, 2, B, h, N, d): one packed copy feeds both directions
            qkv = self.qkv(torch.stack([x1, x2], 0)).reshape(2, B, N, 3, h, d).permute(3, 0, 1, 4, 2, 5).contiguous()
            q, k, v = qkv[0], qkv[1], qkv[2]                           # each (2, B, h, N, d); index 0 = image 1
            # direction 1: attn_1 = q2 k1^T, contracted with v1 (:275,:291); direction 2: q1 k2^T with v2 (:276,:292)
            qq = torch.stack([q[1], q[0]], 0).reshape(2 * B * h, N, d)
            kk, vv = k.reshape(2 * B * h, N, d), v.reshape(2 * B * h, N, d)
            if x1.is_cuda and d == 64 and self.hip_training:       # K2 forward + backward kernels
                F = ops.emm_bilinear_train(qq, kk, vv, self.pos6, self.scale)
            else:                                                  # CPU / hip_training = False: dense score tensors, test-side helper
                F = ag.require('CrossAttention on CPU tensors or with hip_training = False').bilinear_attention(qq, kk, vv, self.pos6, self.scale)
        else:
            # inference: the qkv Linear on K9 with one output plane per (tensor, head) -- the layout K2 reads in place
            pk = self.__dict__.setdefault('_packs', ops.PackCache())
            ts = [self.qkv.weight] + ([self.qkv.bias] if self.qkv.bias is not None else [])
            pc = pk.get(('qkv', self.plain16), ts, lambda: ops.PackedConv(self.qkv.weight, None, self.qkv.bias, split=not self.plain16))
            # the two images are the two halves of one (2B, N, C) buffer when they come from CrossBlock: no copy
            # (same storage, x2 right behind x1: neighbours in the allocator's pool do not qualify)
            adjacent = (x1.is_contiguous() and x2.is_contiguous() and x1.shape == x2.shape and
                        x1.untyped_storage().data_ptr() == x2.untyped_storage().data_ptr() and
                        x2.storage_offset() == x1.storage_offset() + x1.numel())
            x12 = torch.as_strided(x1, (2, B, N, C), (B * N * C, N * C, C, 1)) if adjacent else torch.stack([x1, x2], 0)
            planes = ops.linear_f16s(x12.reshape(2 * B, N, C), pc, out_planes=3 * h)     # (3h, 2B, N, d)
            if self.exact_f32:
                P = planes.view(3, h, 2, B, N, d).permute(0, 2, 3, 1, 4, 5)                # (tensor, image, B, h, N, d)
                qq = torch.stack([P[0, 1], P[0, 0]], 0).reshape(2 * B * h, N, d).contiguous()
                F, _ = ops.emm_bilinear(qq, P[1].reshape(2 * B * h, N, d).contiguous(), P[2].reshape(2 * B * h, N, d).contiguous(),
                                        self.pos6, self.scale, exact_f32=True)
            else:
                F, _ = ops.emm_bilinear_planes(planes, self.pos6, self.scale, B, plain16=self.plain16)         # (2Bh, 70, 70)
        F = F.view(2, B, h, d + 6, d + 6)
        # raw reshape of (B, h, 70, 70) to (B, 280, 70), then transpose (:294-295)
        f1 = F[0].reshape(B, C + 6 * h, (C + 6 * h) // h).transpose(-2, -1)
        f2 = F[1].reshape(B, C + 6 * h, (C + 6 * h) // h).transpose(-2, -1)
        if ag.needs_grad(f1, self.proj_fundamental.weight) or not f1.is_cuda:
            f2 = self.proj_fundamental(f2)
            f1 = self.proj_fundamental(f1)
        else:                                                      # K9 (row-independent), both directions in one launch
            pk = self.__dict__.setdefault('_packs', ops.PackCache())
            pw, pb = self.proj_fundamental.weight, self.proj_fundamental.bias
            # K9 reads 16-byte channel groups: 280 input channels at mp3d's shape; the 8-Point-ViT's 210 (= 192 + 6 x 3) get two
            # zero columns on both operands (exact)
            pad = (-pw.shape[1]) % 4
            pp = pk.get('proj', [pw, pb], lambda: ops.PackedConv(nn.functional.pad(pw.detach(), (0, pad)) if pad else pw, None, pb))
            f12 = torch.stack([f1, f2], 0)
            f12 = ops.linear_f16s(nn.functional.pad(f12, (0, pad)) if pad else f12.contiguous(), pp)
            f1, f2 = f12[0], f12[1]
        return f2, f1                                              # flipped on purpose (:301-303)


def _init_vit(m):
    # the non-jax branch of the reference's _init_vit_weights (:151-181)
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


class CrossBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, use_pos_embedding=False, distilled=False, pos=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.cross_attn = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, pos=pos)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)
        self.h, self.w = 60, 80
        self.pos_embed = 0
        if use_pos_embedding:
            self.pos_embed = nn.Parameter(torch.zeros(1, self.h * self.w, 256))
            nn.init.trunc_normal_(self.pos_embed, std=.02)
        self.apply(_init_vit)

    def forward(self, x, intrinsics=None, loftr_preds=None, inv_loftr_preds=None):
        """x = cat([feat0, feat1], dim=0): (2B, N, C).  Returns (2B, 70, C), rows (2b, 2b+1) = pair b."""
        if not x.is_cuda:
            ag.require('CrossBlock on CPU tensors')
        b_s, h_w, nf = x.shape
        B = b_s // 2
        x = x + self.pos_embed
        x1_in, x2_in = x[:B], x[B:]           # == x.reshape(-1, 2, h_w, nf)[:, 0/1] for the reference's B = 1
        if ag.needs_grad(x, self.norm1.weight):
            n1_1, n1_2 = self.norm1(x1_in), self.norm1(x2_in)
        else:                                 # one LayerNorm launch over both images; the halves stay adjacent
            n = ops.layernorm(x.contiguous(), self.norm1.weight, self.norm1.bias, self.norm1.eps)
            n1_1, n1_2 = n[:B], n[B:]
        f1, f2 = self.cross_attn(n1_1, n1_2, intrinsics=intrinsics,
                                 loftr_preds=loftr_preds, inv_loftr_preds=inv_loftr_preds)
        f = torch.cat([f1.unsqueeze(1), f2.unsqueeze(1)], dim=1).reshape(b_s, -1, nf)
        if ag.needs_grad(f, self.norm2.weight) or not f.is_cuda:
            return f + self.mlp(self.norm2(f))
        f = f.contiguous()
        return self.mlp(ops.layernorm(f, self.norm2.weight, self.norm2.bias, self.norm2.eps), residual=f)   # K6 + K9


_POSE_STATS_DEV = {}


def _pose_stats_on(device):
    """(pose_mean_6d, pose_std_6d) on `device`, uploaded once (a .to(device) of a pageable CPU tensor per head call was two blocking copies)."""
    key = str(device)
    if key not in _POSE_STATS_DEV:
        _POSE_STATS_DEV[key] = (pose_mean_6d.to(device), pose_std_6d.to(device))
    return _POSE_STATS_DEV[key]


class HeadFeatures:
    """What LocalFeatureTransformerRegressor.compute_features hands to forward_emm on the GPU inference path: the (B, 35840)
    features of transformer.py:497 plus the feature-only part of the two 35840-wide first layers (encoder[0], moe_predictor[0])."""

    def __init__(self, feats, enc0, moe0):
        self.feats, self.enc0, self.moe0 = feats, enc0, moe0

    def reshape(self, *shape):          # callers that only want the reference's tensor
        return self.feats.reshape(*shape)


class LocalFeatureTransformerRegressor(nn.Module):
    """LoFTR layer(s) + EMM head + solver/regressor gate (transformer.py:350-499)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        num_heads, feat_size, pos_enc = 4, 256, 6
        pose_size_in = pose_size = 9
        self.pose_size = pose_size
        if config['regress']['regress_use_num_corres']:
            pose_size_in += 1
        if config['use_many_ransac_thr']:
            pose_size_in += 3
        self.pose_size_in = pose_size_in
        self.H = int(num_heads * 2 * (feat_size // num_heads + pos_enc) * (feat_size // num_heads))   # 35840
        self.H2 = 512
        rc = config['regress']
        if rc['use_simple_moe']:
            self.encoder = nn.Sequential(nn.Linear(self.H, self.H2), nn.ReLU(), nn.Linear(self.H2, self.H2))
            n_gate = 1 if rc['use_1wt'] else (2 if rc['use_2wt'] else pose_size)
            self.moe_predictor = nn.Sequential(
                nn.Linear(self.H + pose_size + pose_size_in, self.H2), nn.ReLU(),
                nn.Linear(self.H2, self.H2), nn.ReLU(),
                nn.Linear(self.H2, n_gate), nn.Sigmoid())
            self.pose_regressor_simple_moe = nn.Sequential(
                nn.Linear(self.H2, self.H2), nn.ReLU(), nn.Linear(self.H2, pose_size))
        else:
            self.pose_regressor = nn.Sequential(
                nn.Linear(self.H, self.H2), nn.ReLU(), nn.Linear(self.H2, self.H2), nn.ReLU(),
                nn.Linear(self.H2, pose_size))
        self.norm = partial(nn.LayerNorm, eps=1e-6)(feat_size)
        self.emm = CrossBlock(dim=feat_size, num_heads=num_heads, qkv_bias=True,
                              use_pos_embedding=rc['use_pos_embedding'])
        if config['regress_loftr_layers'] > 0:
            self.loftr = LocalFeatureTransformer(config['regress'])
        # The evaluation loop calls the head FINE_PRED_STEPS times on the SAME coarse features; only the 13 solver
        # numbers change between calls (lightning_loftr.py:338-343).  Everything up to `features` (2 LoFTR layers,
        # K2, CrossBlock, LayerNorm) does not depend on them -- LinearAttention ignores loftr_preds
        # (linear_attention.py:20).  This module keeps NO state about it: `compute_features` is pure, and the reuse
        # lives in the caller-owned data dict (LoFTR.forward_rt_prediction), whose lifetime is one batch.
        self.cache_features = True          # False: LoFTR.forward_rt_prediction recomputes the features on every call

    def feature_stamp(self):
        """What `compute_features` depends on besides its inputs: every weight it reads (storage + version, so that
        load_state_dict / an optimizer step / an in-place edit invalidate) and the operand precision of its layers."""
        mods = [self.emm, self.norm] + ([self.loftr] if self.config['regress_loftr_layers'] > 0 else [])
        if self.config['regress']['use_simple_moe']:        # their feature-only part travels with the features (HeadFeatures)
            mods += [self.encoder[0], self.moe_predictor[0]]
        ws = tuple((p.data_ptr(), ops.tensor_version(p)) for m in mods for p in m.parameters())
        prec = tuple((m.split_operands, m.dense_split) for m in self.modules() if isinstance(m, LoFTREncoderLayer))
        return ws, prec, self.training, ops.activation_exponent_value(), self.emm.cross_attn.exact_f32, self.emm.cross_attn.plain16

    def compute_features(self, feat0, feat1, loftr_preds=None, inv_loftr_preds=None):
        """(B, 35840) head features of transformer.py:488-497 + :424-428 (LoFTR layer(s), CrossBlock, LayerNorm)."""
        f0, f1 = feat0, feat1
        if self.config['regress_loftr_layers'] > 0:
            f0, f1 = self.loftr(f0, f1, loftr_preds=loftr_preds, inv_loftr_preds=inv_loftr_preds, joint_out=True)
        B = f0.shape[0]
        # the two halves of ONE buffer (joint_out): same storage, f1 right behind f0 -- two separate allocations that happen to be
        # neighbours in the allocator's pool do not qualify (as_strided may not reach past f0's storage)
        adjacent = (f0.is_cuda and f0.is_contiguous() and f1.is_contiguous() and f0.shape == f1.shape and
                    f0.untyped_storage().data_ptr() == f1.untyped_storage().data_ptr() and
                    f1.storage_offset() == f0.storage_offset() + f0.numel())
        x01 = torch.as_strided(f0, (2 * B,) + tuple(f0.shape[1:]), f0.stride()) if adjacent else torch.cat([f0, f1], dim=0)
        x = self.emm(x01, loftr_preds=loftr_preds, inv_loftr_preds=inv_loftr_preds)
        if ag.needs_grad(x, self.norm.weight) or not x.is_cuda:
            return self.norm(x).reshape([B, -1])
        feats = ops.layernorm(x.contiguous(), self.norm.weight, self.norm.bias, self.norm.eps).reshape(B, -1)
        if not self.config['regress']['use_simple_moe']:
            return feats
        # The first layers of `encoder` and `moe_predictor` read only these features (:428, :448: cat[features, pose, solver
        # numbers]): their feature part is computed here, once per batch, and travels with the features -- the second head
        # call of a step then re-runs only the 22-number-dependent remainder instead of streaming 2 x 73 MB of weights again
        pk = self.__dict__.setdefault('_packs', ops.PackCache())
        e0, m0 = self.encoder[0], self.moe_predictor[0]
        both = pk.get('enc0|moe0', [e0.weight, m0.weight],
                      lambda: ops.PackedRows(torch.cat([e0.weight, m0.weight[:, :self.H]], 0)))
        pre = ops.rows_linear(feats, both)                                  # (B, 1024): encoder[0] | moe_predictor[0], no bias yet
        return HeadFeatures(feats, pre[:, :self.H2], pre[:, self.H2:])

    def forward_emm(self, feat0, feat1, loftr_preds=None, inv_loftr_preds=None, features=None):
        if features is None:                  # reference call shape: feat0 / feat1 are the LoFTR-layer outputs
            B = feat0.shape[0]
            x = self.emm(torch.cat([feat0, feat1], dim=0), loftr_preds=loftr_preds, inv_loftr_preds=inv_loftr_preds)
            features = self.norm(x).reshape([B, -1])
        rc = self.config['regress']
        pre = None
        if isinstance(features, HeadFeatures):
            pre, features = features, features.feats
        if not rc['use_simple_moe']:
            return self.pose_regressor(features), (features if rc['save_mlp_feats'] else None), None
        if pre is not None:
            # inference on the GPU: the 22-number-dependent remainder of the head.  (Captured as ONE HIP graph -- static shapes, ~45
            # launches -- and replayed it measured the same 8.7-8.9 ms per pair and 89-91 ms per 32 pairs as launched eagerly: without
            # a profiler attached a launch costs the host ~10 us, no more than these kernels run)
            pose, gate = self._tail(pre.enc0, pre.moe0, loftr_preds)
            return pose, (features if rc['save_mlp_feats'] else None), gate
        mean_t, std_t = (t[:3] for t in _pose_stats_on(features.device))
        pred_reg_6d = self.pose_regressor_simple_moe(self.encoder(features))
        gate_in = lambda: self.moe_predictor(torch.cat([features, pred_reg_6d, loftr_preds], dim=-1))
        pose, gate = self._blend(pred_reg_6d, loftr_preds, gate_in, mean_t, std_t)
        return pose, (features if rc['save_mlp_feats'] else None), gate

    def _blend(self, pred_reg_6d, loftr_preds, gate_fn, mean_t, std_t):
        """transformer.py:432-467: the solver's translation at the regressor's length, the gate, the blended pose."""
        rc = self.config['regress']
        pred_reg_t = pred_reg_6d[..., :3]
        loftr_pred_t = loftr_preds[..., :3]
        if rc['scale_8pt']:
            # give the solver's unit translation the regressor's length (:436-446)
            solver_t = loftr_pred_t * std_t + mean_t
            reg_t = pred_reg_t * std_t + mean_t
            solver_t = solver_t * torch.linalg.norm(reg_t, dim=-1, keepdim=True) \
                / torch.clamp(torch.linalg.norm(solver_t, dim=-1, keepdim=True), 1e-3, 100)
            loftr_pred_t = (solver_t - mean_t) / std_t
        extra = self.pose_size_in - self.pose_size
        loftr_pred_R = loftr_preds[..., 3:-extra] if extra > 0 else loftr_preds[..., 3:]   # :452-455
        gate = gate_fn()
        if rc['use_2wt']:
            if rc['use_5050_weight']:
                raise NotImplementedError('use_5050_weight is a debugging branch in the reference (:461-464)')
            w_t, w_r = gate[..., 0:1], gate[..., 1:2]
        else:
            w_t = w_r = gate[..., 0:1]
        pred_T = w_t * pred_reg_t + (1 - w_t) * loftr_pred_t                               # :466
        pred_R = w_r * pred_reg_6d[..., 3:] + (1 - w_r) * loftr_pred_R                     # :467
        return torch.cat([pred_T, pred_R], dim=-1), gate

    def _tail(self, enc0, moe0, loftr_preds):
        """The part of the head that reads the solver's numbers (transformer.py:428-467 behind the 35840-wide first layers, whose
        feature-only halves enc0 / moe0 come from compute_features): every Linear on K15 (exact fp32, one fma chain per output in a
        fixed order: pair b's pose does not depend on the batch it is computed in).  A pure function of its three tensors and the
        weights -- about 45 launches of 2-60 us."""
        pk = self.__dict__.setdefault('_packs', ops.PackCache())
        lin = lambda name, m, **kw: pk.get(name, [m.weight, m.bias], lambda: ops.PackedRows(m.weight, m.bias, **kw))
        enc, reg, moe = self.encoder, self.pose_regressor_simple_moe, self.moe_predictor
        mean_t, std_t = (t[:3] for t in _pose_stats_on(enc0.device))
        h = torch.relu(enc0 + enc[0].bias)
        h = ops.rows_linear(h, lin('enc2', enc[2]))
        h = ops.rows_linear(h, lin('reg0', reg[0]), act='relu')
        pred_reg_6d = ops.rows_linear(h, lin('reg2', reg[2]))

        def gate_fn():
            tail = torch.cat([pred_reg_6d, loftr_preds.to(pred_reg_6d.dtype)], dim=-1).contiguous()      # (B, 9 + 13)
            g = ops.rows_linear(tail, lin('moe0-tail', moe[0], cols=(self.H, moe[0].weight.shape[1])), act='relu', add=moe0)
            g = ops.rows_linear(g, lin('moe2', moe[2]), act='relu')
            return ops.rows_linear(g, lin('moe4', moe[4]), act='sigmoid')
        return self._blend(pred_reg_6d, loftr_preds, gate_fn, mean_t, std_t)

    def forward(self, feat0, feat1, loftr_preds=None, inv_loftr_preds=None, mask0=None, mask1=None, F=None,
                features=None):
        """features: the output of compute_features on these same inputs, when the caller already has it."""
        if features is None:
            features = self.compute_features(feat0, feat1, loftr_preds, inv_loftr_preds)
        return self.forward_emm(feat0, feat1, loftr_preds, inv_loftr_preds, features=features)

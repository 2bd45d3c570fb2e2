#!/usr/bin/env python
"""Per-launch floor table of ONE steady-state step of the headline workload (32 pairs @ 640x480: match + 2 solver rounds + 2 head
calls) -- VERDICT r4 item 2.

Every call the step makes through the C ABI (include/far_hip.h) is recorded at the ctypes boundary with its integer arguments
(the shapes) and bracketed by two HIP events on the stream it is launched on; a model per entry point turns the shapes into

    bytes       ALGORITHMIC HBM bytes: every input tensor read once, every output written once, fp32 weights once
    mfma        EXECUTED f16 MFMA flops: 2 x MACs the matrix pipe really runs -- x3 for split-fp16 operands (hi.hi + hi.lo + lo.hi),
                x16/36 for K17's Winograd F(2x2, 3x3), with the kernels' channel / tile padding
    hbm_floor   bytes / 6.3 TB/s   (the write / copy bandwidth the part sustains: tools/hbm_probe.py; nominal 8)
    mfma_floor  mfma / the dense-f16 rate far_mfma_probe_f16 sustains on THIS box in THIS run (operands from LDS)
    max, sum    of the two floors: `max` is the launch's speed of light when the two overlap perfectly, `sum` when nothing does
    ms          measured (events; mean over --steps recorded steps)

Calls with no matrix work and no model (K4's float64 solver kernels, small glue) carry their HBM floor only, or none: their
measured time counts as its own floor (`floor = ms`), so the step's speed of light is not flattered by them.
Output: the table in execution order, the totals, and the ranking by ms - max(floors).

    python tools/step_floors.py [--steps 3] [--out profiles/r05_step_floors.txt]
"""
import argparse
import ctypes
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_ACHIEVABLE = 6.3e12       # B/s (tools/hbm_probe.py: the copy / store rate the part sustains; bench.py prices against the nominal 8)


def c16(x):
    return (x + 15) // 16 * 16


def cN(x, n):
    return (x + n - 1) // n * n


class Recorder:
    """Stands in for the loaded CDLL: far_* functions whose last argument is the stream are bracketed by events and logged."""

    def __init__(self, lib, sigs):
        self._lib, self._sigs = lib, sigs
        self.on = False
        self.calls = []

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        sig = self._sigs.get(name)
        if not name.startswith('far_') or sig is None or not sig[1] or sig[1][-1] is not ctypes.c_void_p or sig[0] is not ctypes.c_int:
            return fn
        rec = self

        def wrapped(*args):
            if not rec.on:
                return fn(*args)
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = fn(*args)
            e1.record(st)
            rec.calls.append((name, [rec._plain(a) for a in args[:-1]], e0, e1))
            return rc
        return wrapped

    @staticmethod
    def _plain(a):
        if hasattr(a, '_obj'):                          # ctypes.byref(struct): a snapshot of its fields
            o = a._obj
            return {f[0]: (getattr(o, f[0]) if not isinstance(getattr(o, f[0]), ctypes.Array) else None) for f in o._fields_}
        if isinstance(a, ctypes.c_void_p):
            return a.value or 0
        if isinstance(a, (int, float)) or a is None:
            return a or 0
        return getattr(a, 'value', 0) or 0


# ----------------------------------------------------------------------------------------------------------------------------
# models: (label, bytes, executed mfma flops)
# ----------------------------------------------------------------------------------------------------------------------------
def conv_model(d, kind='k9', extra=None):
    N, H, W, Cin, Cout, ks, st = d['N'], d['H'], d['W'], d['Cin'], d['Cout'], d['ksize'], d['stride']
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    P = N * Ho * Wo
    rd_in = N * H * W * Cin if ks == 3 else P * Cin                # a strided 1x1 reads the pixels it uses
    by = 4 * (rd_in + Cout * Cin * ks * ks)
    if kind != 'kv':
        by += 4 * P * Cout
    if d['res']:
        by += 4 * P * Cout // max(d['res_group'], 1)
    if d['post_res']:
        by += 4 * P * Cout
    if d['up']:
        by += 4 * P * Cout // 4
    terms = 3 if d['split'] else 1
    if kind == 'wino':
        tiles = N * cN(H, 16) * cN(W, 16)
        co = Cout if Cout % 64 == 0 or Cout % 64 > 32 else Cout - Cout % 64 + 32      # HALF body on a short last block
        mf = 2.0 * tiles * c16(Cin) * cN(co, 32) * 9 * terms * 16 / 36
        label = f'K17 3x3 {Cin}->{Cout} @{H}x{W}'
    else:
        mf = 2.0 * cN(P, 128) * c16(Cin) * cN(Cout, 64) * ks * ks * terms
        what = {'k9': 'conv' if (H > 1 or ks == 3) else 'linear', 'kv': 'linear k|v + K^T V state', 'q': 'linear q + attention apply',
                'gather': 'linear on gathered windows'}[kind]
        label = f'K9 {what} {ks}x{ks}' + (f'/s{st}' if st > 1 else '') + f' {Cin}->{Cout} rows {P}' + \
                (' +res' if d['res'] else '') + (' +LN' if d['ln_gamma'] else '') + (' +up' if d['up'] else '') + \
                (f' planes {d["out_planes"]}' if d['out_planes'] > 1 else '')
    return label, by, mf


def model(name, a):
    """-> (label, algorithmic bytes or None, executed f16 MFMA flops or None)."""
    if name == 'far_conv_nhwc_f32':
        return conv_model(a[0])
    if name == 'far_conv3x3_wino_f32':
        return conv_model(a[0], 'wino')
    if name == 'far_linear_kv_f16s':
        return conv_model(a[0], 'kv')
    if name == 'far_linear_q_apply_f16s':
        return conv_model(a[0], 'q')
    if name == 'far_linear_gather_f16s':
        lab, by, mf = conv_model(a[0], 'gather')
        return lab, by, mf
    if name == 'far_stem7x7_nhwc_f32':
        N, H, W, Co = a[4], a[5], a[6], a[7]
        P = N * (H // 2) * (W // 2)
        return f'K10 stem 7x7/s2 1->{Co} @{H}x{W}', 4 * (N * H * W + P * Co), 2.0 * P * 64 * cN(Co, 64) * 3
    if name == 'far_coarse_match_f16s':
        Z, L, S, C = a[2], a[3], a[4], a[5]
        conf = 4 * Z * L * S if a[19] else 0
        # k1_rowstats x2 (split: 3 MFMA terms each) + k1_screen (hi.hi: 1 term) + k1_match on the surviving ~3 % of the tiles
        return f'K1 coarse match Z={Z} L={L}', 4 * Z * (L + S) * C + conf + 40 * Z * L, 2.0 * Z * L * S * C * (6 + 1 + 0.09)
    if name in ('far_emm_pv_f16s', 'far_emm_pv_f16'):
        Z, N = a[4], a[5]
        by = 4 * Z * N * (3 * 64 + 70)
        terms = 3 if name.endswith('s') else 1
        stats = 2.0 * Z * N * N * 64 * terms
        pv = Z * math.ceil(N / 128) * math.ceil(N / 64) * 20 * terms * 4 * 32768.0   # per 128 queries x 64 keys: (8 + 12) x terms MFMAs per wave, 4 waves
        return f'K2 bilinear attention Z={Z} N={N}', by, stats + pv
    if name == 'far_emm_contract_f32':
        Z = a[6]
        return f'K2 contraction v~^T T Z={Z}', None, None
    if name in ('far_attn_block_f16s', 'far_attn_block_f16'):
        nwin, L, S, dm = a[3], a[4], a[5], a[6]
        rows = nwin * L
        # k, v, q, merge projections (128 x 128 each) on rows x 128; the attention core itself is VALU / small
        return f'K14 attention block d128 windows {nwin}', 4 * (nwin * (L + S) * dm + rows * dm + 4 * dm * dm), 2.0 * cN(rows, 32) * dm * dm * 4 * (3 if name.endswith('s') else 1)
    if name in ('far_mlp_fused_f16s', 'far_mlp_fused_f16'):
        R, dm = a[3], a[4]
        return f'K13 MLP block d128 rows {R}', 4 * (3 * R * dm + 6 * dm * dm), 2.0 * cN(R, 32) * (2 * dm * 2 * dm + 2 * dm * dm) * (3 if name.endswith('s') else 1)
    if name == 'far_layernorm_f32':
        rows, C = a[4], a[5]
        return f'K6 LayerNorm rows {rows} C={C}', 4 * rows * C * (3 if a[3] else 2), None
    if name == 'far_linear_attention_apply_f32':
        N, L, H = a[2], a[3], a[5]
        return f'K5 attention apply N={N} L={L}', 4 * 2 * N * L * H * 32, None
    if name == 'far_linear_attention_f32':
        N, L, S, H, D = a[3], a[4], a[5], a[6], a[7]
        return f'K5 linear attention N={N} L={L}', 4 * N * (2 * L + 2 * S) * H * D, None
    if name == 'far_fine_expect_f32':
        M, W, C = a[2], a[3], a[4]
        return f'K3 fine expectation M={M}', 4 * 2 * M * W * W * C, None
    if name == 'far_fine_gather_f32':
        M, W, C = a[13], a[11], a[5]
        return f'K3 window gather M={M}', 4 * 2 * M * W * W * C, None
    if name == 'far_rows_linear_f32':
        B, Nn, K = a[6], a[7], a[8]
        return f'K15 rows linear {K}->{Nn} B={B}', 4 * (Nn * K + B * (K + Nn)), None
    if name == 'far_solver_f64':
        return f'K4 solver B={a[3]} H={a[14]} Mtot={a[4]}', None, None
    if name == 'far_affine_act_f32':
        return 'K7 affine + activation', None, None
    if name == 'far_upsample2x_add_f32':
        return 'K8 upsample + add', None, None
    return name.replace('far_', ''), None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--pairs', type=int, default=32)
    ap.add_argument('--out', default=None)
    ap.add_argument('--precision', default='fp32', help="LoFTR.set_precision mode; the MFMA floors follow each launch's operand form (x3 for split pairs, x1 for plain fp16 / bf16)")
    ap.add_argument('--bf16-k1', action='store_true', help='with --precision fp16: the coarse matcher on its bf16 variant too')
    a = ap.parse_args()
    from far_amd import _lib, synth
    from far_amd.config import far_eval_config
    from far_amd.loftr import LoFTR
    from far_amd.pipeline import test_step
    import numpy as np
    import bench
    real = _lib.load()
    rec = Recorder(real, _lib.SIGNATURES)
    _lib._lib = rec
    dev = torch.device('cuda', 0)
    cfg = far_eval_config()
    model_ = LoFTR(cfg).eval()
    synth.load_synthetic(model_, seed=0)
    model_ = model_.to(dev)
    model_.set_precision(a.precision)
    if a.bf16_k1:
        model_.coarse_matching.bf16 = True
    im0, im1 = synth.synth_image_pair(a.pairs, seed=1234)
    K = torch.from_numpy(np.stack([synth.MP3D_K] * a.pairs)).to(dev)
    base = {'image0': torch.from_numpy(im0).to(dev), 'image1': torch.from_numpy(im1).to(dev), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
    for _ in range(4):
        test_step(model_, dict(base), H=2048, seed=0)
    torch.cuda.synchronize()
    probe = bench.mfma_sustained_peak()
    sustained = probe['operands_from_lds']['tflops'] * 1e12
    sustained_reg = probe['operands_in_registers']['tflops'] * 1e12
    # un-instrumented step time (events around whole steps): the product configuration first (the head's feature stage on its side
    # stream), then with every launch on the step's own stream -- the configuration the per-call times below are taken in (a call
    # that shares the GPU with another stream's launches has no duration of its own)
    def plain():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            test_step(model_, dict(base), H=2048, seed=0)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 5
    product_ms = plain()
    model_.head_side_stream = model_.fpn_side_stream = False
    test_step(model_, dict(base), H=2048, seed=0)
    plain_ms = plain()
    steps = []
    step_ms = []
    for _ in range(a.steps):
        rec.calls = []
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rec.on = True
        s0.record()
        test_step(model_, dict(base), H=2048, seed=0)
        s1.record()
        rec.on = False
        torch.cuda.synchronize()
        steps.append([(n, ar, x.elapsed_time(y)) for n, ar, x, y in rec.calls])
        step_ms.append(s0.elapsed_time(s1))
    n = len(steps[0])
    assert all(len(s) == n and [c[0] for c in s] == [c[0] for c in steps[0]] for s in steps), 'the recorded steps differ in their call sequence'
    rows = []
    for i in range(n):
        name, args, _ = steps[0][i]
        ms = sum(s[i][2] for s in steps) / len(steps)
        label, by, mf = model(name, args)
        hb = by / HBM_ACHIEVABLE * 1e3 if by else None
        mm = mf / sustained * 1e3 if mf else None
        fl = [x for x in (hb, mm) if x is not None]
        rows.append(dict(i=i, name=name, label=label, bytes=by, mfma=mf, hbm=hb, mm=mm, mx=max(fl) if fl else None, sm=sum(fl) if fl else None, ms=ms))
    out = []
    w = out.append
    tot_ms = sum(r['ms'] for r in rows)
    w(f'# tools/step_floors.py: one steady-state step of the headline workload ({a.pairs} pairs @ 640x480, match + 2 solver rounds + 2 head calls),')
    w(f'# mean of {a.steps} recorded steps; library build id {real.far_build_id().decode()}, commit {os.environ.get("FAR_COMMIT", "?")}')
    w(f'# step time: {product_ms:.2f} ms in the product configuration (the head\'s feature stage on a side stream next to K1 and the fine level);')
    w(f'#   on ONE stream (LoFTR.head_side_stream = False, as everything below): {plain_ms:.2f} ms un-instrumented (events around 5 steps); {sum(step_ms) / len(step_ms):.2f} ms with an event pair around each of the {n} C-ABI calls;')
    w(f'#   sum of the calls {tot_ms:.2f} ms, the rest = ATen glue kernels + idle')
    w(f'# floors: HBM at {HBM_ACHIEVABLE / 1e12:.1f} TB/s (achievable; nominal 8); MFMA at the dense-f16 rate far_mfma_probe_f16 sustained in this run with operands')
    w(f'#   from LDS: {sustained / 1e12:.0f} TFLOP/s (operands in registers: {sustained_reg / 1e12:.0f}; nominal 2 500)')
    w('# bytes = algorithmic HBM bytes (inputs once + outputs once + fp32 weights); mfma = EXECUTED f16 MFMA flops (x3 split operands, x16/36 Winograd, padding)')
    w('# a call without a model (float64 solver, small glue) has floor = its measured time')
    w('')
    w(f'{"#":>3} {"ms":>8} {"hbm_fl":>7} {"mfma_fl":>7} {"max":>7} {"sum":>7} {"ms-max":>7} {"GB":>7} {"TFLOP":>7}  launch')
    for r in rows:
        f = lambda v, p=3: f'{v:.{p}f}' if v is not None else '-'
        gap = r['ms'] - r['mx'] if r['mx'] is not None else None
        w(f'{r["i"]:>3} {r["ms"]:8.3f} {f(r["hbm"]):>7} {f(r["mm"]):>7} {f(r["mx"]):>7} {f(r["sm"]):>7} {f(gap):>7} '
          f'{f(r["bytes"] / 1e9 if r["bytes"] else None):>7} {f(r["mfma"] / 1e12 if r["mfma"] else None):>7}  {r["label"]}')
    sol_max = sum(r['mx'] if r['mx'] is not None else r['ms'] for r in rows)
    sol_sum = sum(r['sm'] if r['sm'] is not None else r['ms'] for r in rows)
    unattributed = plain_ms - tot_ms
    w('')
    w(f'## totals over the {n} calls')
    w(f'measured                         {tot_ms:8.2f} ms   (+ {max(unattributed, 0):.2f} ms outside the C ABI: ATen glue, idle)')
    w(f'sum of max(hbm, mfma) floors     {sol_max:8.2f} ms   = the step\'s speed of light in precision mode {a.precision!r} if every launch overlapped its two floors perfectly')
    w(f'sum of (hbm + mfma) floors       {sol_sum:8.2f} ms   = if no launch overlapped them at all')
    pairs = a.pairs
    w(f'pairs/s: measured {pairs / plain_ms * 1e3:.0f}; at sum-of-max {pairs / (sol_max + max(unattributed, 0)) * 1e3:.0f}; at sum-of-sum {pairs / (sol_sum + max(unattributed, 0)) * 1e3:.0f}'
      f'   (500 pairs/s/GPU needs {pairs / 500 * 1e3:.0f} ms per step)')
    w('')
    w('## by kernel family: measured, floors, gap')
    fam = {}
    for r in rows:
        k = r['label'].split(' ')[0] + ' ' + (r['label'].split(' ')[1] if r['label'].startswith('K9') else '')
        e = fam.setdefault(k.strip(), [0, 0.0, 0.0, 0.0])
        e[0] += 1
        e[1] += r['ms']
        e[2] += r['mx'] if r['mx'] is not None else r['ms']
        e[3] += r['sm'] if r['sm'] is not None else r['ms']
    w(f'{"family":<14} {"calls":>5} {"ms":>8} {"max-floor":>9} {"sum-floor":>9} {"ms-max":>8}')
    for k, e in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        w(f'{k:<14} {e[0]:>5} {e[1]:8.2f} {e[2]:9.2f} {e[3]:9.2f} {e[1] - e[2]:8.2f}')
    w('')
    w('## launches ranked by measured - max(floors) (top 40)')
    rk = sorted([r for r in rows if r['mx'] is not None], key=lambda r: -(r['ms'] - r['mx']))[:40]
    for r in rk:
        w(f'{r["ms"] - r["mx"]:7.3f} ms over  (ms {r["ms"]:.3f}, hbm {r["hbm"] or 0:.3f}, mfma {r["mm"] or 0:.3f})  #{r["i"]} {r["label"]}')
    txt = '\n'.join(out) + '\n'
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, 'w') as f:
            f.write(txt)
    print(txt)
    print(json.dumps({'step_ms': round(plain_ms, 3), 'sum_max_ms': round(sol_max, 3), 'sum_sum_ms': round(sol_sum, 3), 'calls': n,
                      'sustained_tflops_lds': round(sustained / 1e12, 1)}))


if __name__ == '__main__':
    main()

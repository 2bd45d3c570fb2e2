"""Derived per-kernel figures from the two counter passes of tools/kprobe.py util (see tools/pmc_util.py for the raw dump):
    pass 1: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES      pass 2: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA
  shader clock   = GRBM_GUI_ACTIVE / 8 XCCs / duration          (the counter is summed over the 8 XCCs; an HBM-bound torch
                                                                 kernel gives 2.35-2.45 GHz = the nominal 2.4 GHz)
  MFMA busy      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)     (rocprofv3's MfmaUtil expression)
  LDS conflicts  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;   LDS busy = SQ_LDS_IDX_ACTIVE / 256 CUs / (GRBM_GUI_ACTIVE / 8)
Usage: python tools/pmc_util_summary.py pass1.txt pass2.txt"""
import re
import sys


def parse(path):
    out = {}
    for ln in open(path):
        m = re.match(r'(.*) grid=(\d+) n=(\d+) (.*)', ln)
        if not m:
            continue
        vals = dict((k, float(v)) for k, v in (kv.split('=') for kv in m.group(4).split()))
        out[(m.group(1), int(m.group(2)))] = vals
    return out


a, b = parse(sys.argv[1]), parse(sys.argv[2])
names = [('k_conv<3, 2, 2, 4, true, 1', 9830400, 'K9 3x3 196->196 @240x320 (64 images)'), ('k_conv<3, 4, 1, 4', 4915200, 'K9 3x3 128->128 @240x320'),
         ('k_conv<3, 2, 2, 4, true, 1', 2457600, 'K9 3x3 256->256 @120x160'), ('k_conv<1, 2, 2, 4', 614400, 'K9 Linear 512->512, 153 600 rows'),
         ('k1_rowstats', 311296, 'K1 k1_rowstats (32 pairs)'), ('k1_match', 311296, 'K1 k1_match (32 pairs)'),
         ('10k_rowstats', 2490368, 'K2 k_rowstats (256 problems)'), ('4k_pv', 2490368, 'K2 k_pv (256 problems)'),
         ('k_la_kv_partial<32>', 491520, 'K5 kv (64 maps)'), ('k_la_apply<32>', 1245184, 'K5 apply (64 maps)'),
         ('vectorized_elementwise_kernel<4, at::native::(anon', 120422400, 'torch elementwise (randn arithmetic, HBM-bound reference)'),
         ('FillF', 92160000, 'torch fill_ (HBM-bound reference)')]
print(f'{"kernel":58s} {"us":>9s} {"clock GHz":>9s} {"MFMA busy":>9s} {"MFMA instr":>11s} {"LDS busy":>8s} {"LDS confl.":>10s}')
for key, grid, label in names:
    ka = [k for k in a if key in k[0] and k[1] == grid]
    kb = [k for k in b if key in k[0] and k[1] == grid]
    if not ka:
        continue
    va, vb = a[ka[0]], (b[kb[0]] if kb else {})
    cyc = va['GRBM_GUI_ACTIVE'] / 8
    clock = cyc / (va['dur_us'] * 1e3)
    busy = va['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc)
    idx = vb.get('SQ_LDS_IDX_ACTIVE', 0.0)
    conf = vb.get('SQ_LDS_BANK_CONFLICT', 0.0) / idx if idx else 0.0
    print(f'{label:58s} {va["dur_us"]:9.1f} {clock:9.2f} {100 * busy:8.1f}% {vb.get("SQ_INSTS_MFMA", 0):11.4g} {100 * idx / 256 / cyc:7.1f}% {100 * conf:9.1f}%')

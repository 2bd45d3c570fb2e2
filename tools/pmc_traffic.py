"""HBM/fabric traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE counts 128-B requests at 64 B on gfx950 -> x2 (calibrated here on
k_la_kv_partial / k_la_apply, whose byte counts are known exactly: 629 MB and 315 MB read, measured 2 x 307,261 KB
and 2 x 158,943 KB); WRITE_SIZE matched the known byte counts 1:1.  Units of both counters: KiB.
Usage: python tools/pmc_traffic.py fetch.db write.db > profiles/rNN_pmc_traffic.json"""
import json
import sqlite3
import sys


def load(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, grid_size, avg(value), count(*) from counters_collection "
                       "where counter_name=? group by kernel_name, grid_size", (counter,)).fetchall()
    return {(k, g): (v, n) for k, g, v, n in rows}


def short(name):
    # K9 instantiations carry a trailing template argument (the fused-merge flag): report them under the plain key
    # (round 3: two trailing flags -- fused FPN merge, seven-tile mode; both forms of the plain kernels map to one key)
    for key in ['k_conv<3, 2, 2, 4, true, 1', 'k_conv<1, 2, 2, 4, true, 1']:
        for tail in (', false>', ', false, false>', ', false, true>', ', 0, false, 0>', ', 0, true, 0>'):
            if key + tail in name:
                return key + '>'
    for key, lab in (('k_conv<1, 2, 2, 4, true, 1, 0, false, 1>', 'k_conv<1,2,2,4> EPI 1 (k | v projection -> K^T V state)'),
                     ('k_conv<1, 2, 2, 4, true, 1, 0, false, 2>', 'k_conv<1,2,2,4> EPI 2 (q projection -> attention message)'),
                     ('k_conv<1, 4, 1, 4, true, 1, 0, false, 3>', 'k_conv<1,4,1,4> EPI 3 (merge_feat, rows gathered through the match indices)'),
                     ('k_kv_blocks_reduce', 'k_kv_blocks_reduce')):
        if key in name:
            return lab
    for key in ['k_wino<', 'k1_conf_wide', 'k1_conf_fix', 'k1_bwd', 'k2_bwd', 'k1_rowstatsILb1', 'k_cvw_apply', 'k_stats_f32', 'k_match_f32', 'k_emm_pv_f32', 'k_la_kv_partial<32>', 'k_la_apply<32>', 'k_layernorm',
                'k_conv<3, 2, 2, 4, true, 1>', 'k_conv<1, 2, 2, 4, true, 1>', 'k_pv', 'k_rowstats', 'k1_rowstats', 'k1_matchILb0', 'k1_matchILb1']:
        if key in name:
            return key
    return None


if __name__ == '__main__':
    f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for (k, g), (fv, n) in f.items():
        s = short(k)
        if s is None or (k, g) not in w:
            continue
        rd = 2.0 * fv * 1024
        wr = w[(k, g)][0] * 1024
        ent = {'read_bytes': round(rd), 'write_bytes': round(wr), 'total_bytes': round(rd + wr), 'launches': n}
        # the launch bench.py prices as the dominant kernel: 196 -> 196 3x3 at 240 x 320 on 64 images = 64 * 30 * 20 tiles of 256 threads
        if s == 'k_conv<3, 2, 2, 4, true, 1>' and g == 64 * 30 * 20 * 256:
            ent['label'] = 'k_conv[K9 3x3 196->196 @240x320]'
        # K17 on the same layer (tensors stored with 208 channels, as in the step): 64 images x 15 x 20 tiles of 16 x 16 outputs x 4 blocks
        # of 64 output channels, 512 threads each
        if s == 'k_wino<' and g == 64 * 15 * 20 * 4 * 512:
            ent['label'] = 'k_wino[K17 3x3 196->196 @240x320]'
        out[f'{s}|grid={g}'] = ent
    import os
    import subprocess
    commit = os.environ.get('FAR_COMMIT')
    if not commit:
        try:
            commit = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], text=True, stderr=subprocess.DEVNULL).strip()
        except Exception:
            commit = None
    json.dump({'precision': 'fp32', 'commit': commit, 'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/kprobe.py all 32 2 (+ k9 32 2); '
                         'read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB', 'per_launch': out}, sys.stdout, indent=1)

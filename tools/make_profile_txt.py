"""Assembles profiles/<tag>_bench_fp32_kernel_trace.txt from the files tools/collect_profiles.sh leaves in gpurun_out/:
  bench_prof.log (bench.py under rocprofv3), <tag>_trace.txt (tools/profile_report.py on the rocpd database),
  bench_plain.log / bench_c4.log (the un-profiled python bench.py lines), and copies <tag>_pmc_traffic.json.
Usage: python tools/make_profile_txt.py r02 <commit>"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
commit = sys.argv[2] if len(sys.argv) > 2 else 'unknown'
G = ROOT + '/gpurun_out/'
last_json = lambda path: [l for l in open(path).read().splitlines() if l.startswith('{')][-1]
plain, prof, c4 = last_json(G + 'bench_plain.log'), last_json(G + 'bench_prof.log'), last_json(G + 'bench_c4.log')
tr = open(G + f'{tag}_trace.txt').read()
whole, win = tr.split('\n\n', 1)
rnd = tag[1:].lstrip('0')
out = f"""# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-other-modes    (round {rnd}, 1x MI355X, default fp32-grade path, commit {commit})
# collected by tools/collect_profiles.sh; summarised by tools/profile_report.py (rocpd database -> per-kernel tables)
# bench line of the same profiled run:
{prof}

# un-profiled bench line (python bench.py: 5 steps after 3 warm-up steps) on the same box:
{plain}

# un-profiled line of the second workload (python bench.py --workload c4: BASELINE configs[3], cached-LoFTR path, batch 256):
{c4}

## one steady-state step (32 pairs): per kernel
# (two streams since round 6: the head's feature stage -- the encoder-layer k_conv launches behind K1, k_pv, k_emm_contract, k_rows_partial --
#  overlaps K1 and the fine level, so kernel-busy time exceeds the wall time of the window, and the durations of k_attn128 / k_mlp128 /
#  k_pv here are those of kernels that SHARE the GPU: their own durations are in profiles/{tag}_step_floors.txt, taken on one stream)
{win.strip()}

## whole process (includes the warm-up steps and the isolated kernel timings of bench.kernel_rooflines)
{whole.strip()}
"""
open(ROOT + f'/profiles/{tag}_bench_fp32_kernel_trace.txt', 'w').write(out)
if os.path.exists(G + f'{tag}_pmc_traffic.json'):
    import json
    doc = json.load(open(G + f'{tag}_pmc_traffic.json'))
    if os.path.exists(G + f'{tag}_stride_ab.json'):          # what a padded pixel stride is worth, measured
        doc['padded_stride_experiment'] = json.loads(last_json(G + f'{tag}_stride_ab.json'))
    json.dump(doc, open(ROOT + f'/profiles/{tag}_pmc_traffic.json', 'w'), indent=1)
print('wrote', f'profiles/{tag}_bench_fp32_kernel_trace.txt')

# ---- the training step (BASELINE configs[2]): kernel trace of bench.py --workload c3, both legs' bench lines, K16 vs the vendor
if os.path.exists(G + f'{tag}_c3_trace.txt') and os.path.exists(G + 'bench_c3.log'):
    c3, c3v, c3p = last_json(G + 'bench_c3.log'), last_json(G + 'bench_c3_vendor.log'), last_json(G + 'bench_c3_prof.log')
    tr3 = open(G + f'{tag}_c3_trace.txt').read()
    whole3, win3 = tr3.split('\n\n', 1)
    win3 = win3.split('\n## launches of')[0]
    wg = open(G + f'{tag}_wgrad_time.txt').read() if os.path.exists(G + f'{tag}_wgrad_time.txt') else ''
    vk = open(G + f'{tag}_c3_vendor_kernels.txt').read() if os.path.exists(G + f'{tag}_c3_vendor_kernels.txt') else ''
    out3 = f"""# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes    (round {rnd}, 1x MI355X, commit {commit})
# bench line of the profiled run:
{c3p}

# un-profiled lines on the same box: HIP training kernels / the vendor-op comparison leg (--vendor-train)
{c3}
{c3v}

## one steady-state training step (1 pair): per kernel
{win3.strip()}

## vendor / ATen kernels left in that step (tools/aten_in_step.py)
{vk.strip(chr(10))}

## K16 (far_conv_wgrad_f16s + reduction) against the vendor's backward-weights on the backbone's layer shapes (tools/wgrad_time.py, wall per call)
{wg.strip()}
"""
    open(ROOT + f'/profiles/{tag}_c3_training_kernel_trace.txt', 'w').write(out3)
    print('wrote', f'profiles/{tag}_c3_training_kernel_trace.txt')
if os.path.exists(G + 'bench_c5.log'):
    open(ROOT + f'/profiles/{tag}_bench_c5_line.json', 'w').write(last_json(G + 'bench_c5.log') + '\n')

# ---- the 16-bit-operand mode (LoFTR.set_precision('fp16')): informational line + its kernel trace
if os.path.exists(G + f'{tag}_trace_fp16.txt') and os.path.exists(G + 'bench_fp16_prof.log'):
    trm = open(G + f'{tag}_trace_fp16.txt').read()
    wholem, winm = trm.split('\n\n', 1)
    outm = f"""# rocprofv3 --kernel-trace --stats -- python3 bench.py --precision fp16 --steps 3 --warmup 3 --no-cpu-baseline --no-other-modes --no-other-workloads --skip-rooflines    (round {rnd}, 1x MI355X, commit {commit})
# the 16-bit-operand class BASELINE configs[1] names: plain fp16 operands in K9 (backbone, encoder layers), K13 / K14 and K2, bf16 in K1;
# fp32 tensors and accumulation.  NOT the parity configuration (match-set IoU ~0.997 vs it): bench.py reports it under other_modes, never as `value`.
# bench line of the profiled run:
{last_json(G + 'bench_fp16_prof.log')}

## one steady-state step (32 pairs): per kernel
{winm.strip()}
"""
    open(ROOT + f'/profiles/{tag}_bench_fp16_kernel_trace.txt', 'w').write(outm)
    print('wrote', f'profiles/{tag}_bench_fp16_kernel_trace.txt')

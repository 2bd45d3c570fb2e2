"""Development aid: checks the generated code of K9 (conv_igemm_f16s.hip) for the one thing its asm pixel loads rely on -- between an
asm `global_load_dwordx4` (stage_load) and the first instruction that touches its destination registers there is an asm
`s_waitcnt vmcnt(K)` with K <= the number of memory requests issued in between (the counter retires in order, so the load has landed).
A register copy or spill of a staged value scheduled above that wait would read a register the load has not written yet.

  python tools/k9_asm_check.py            # compiles the file to assembly (about 2 minutes) and scans every k_conv instantiation
  python tools/k9_asm_check.py file.s     # scans an existing -S output
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'far_amd', 'csrc', 'conv_igemm_f16s.hip')


def regs(text):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r'\bv(\d+)\b', text))
    return out


def scan(lines, name):
    """lines: the instruction lines of one function (comments stripped, ;APP / ;NO_APP kept).  Returns (#asm loads, [problems])."""
    in_app = False
    items = []                     # (is_asm, text)
    for ln in lines:
        t = ln.strip()
        if t.startswith(';;#ASMSTART') or t.startswith(';APP'):
            in_app = True
            continue
        if t.startswith(';;#ASMEND') or t.startswith(';NO_APP'):
            in_app = False
            continue
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        items.append((in_app, t.split(';')[0].strip()))
    loads = [i for i, (a, t) in enumerate(items) if a and t.startswith('global_load_dwordx4')]
    problems = []
    for i in loads:
        dest = regs(items[i][1].split(',')[0])
        younger, waited = 0, False
        for j in range(i + 1, len(items)):
            a, t = items[j]
            if a and t.startswith('s_waitcnt') and 'vmcnt' in t:
                k = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
                if k <= younger:
                    waited = True
                    break
                continue
            if re.match(r'(global_|buffer_|flat_|scratch_)', t):
                if a and t.startswith('global_load_dwordx4') and regs(t.split(',')[0]) & dest:
                    problems.append(f'{name}: asm load #{i} re-issued into its registers before a wait ({t})')
                    break
                younger += 1
                continue
            if t.startswith('s_endpgm'):
                waited = True              # never read: the trailing loads of the last chunk (vmcnt(0) precedes every exit)
                break
            if regs(t) & dest:
                problems.append(f'{name}: `{t}` touches {sorted(regs(t) & dest)} of asm load `{items[i][1]}` before its wait')
                break
        if not waited and not problems:
            problems.append(f'{name}: no wait found after asm load `{items[i][1]}`')
    return len(loads), problems


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.gettempdir(), 'far_k9_check.s')
        from far_amd import build
        cmd = ['/opt/rocm/bin/hipcc'] + build.BASE_FLAGS + ['-I', os.path.dirname(SRC), '--cuda-device-only', '-S', SRC, '-o', path]
        subprocess.run([c for c in cmd if c not in ('-fPIC',)], check=True, stderr=subprocess.DEVNULL)
    text = open(path).read().splitlines()
    starts = [(i, l.split(':')[0]) for i, l in enumerate(text) if re.match(r'^_Z\w*k_conv\w*:', l)]
    total, bad = 0, []
    for n, (i, name) in enumerate(starts):
        end = next((j for j in range(i, len(text)) if text[j].strip().startswith('s_endpgm')), len(text))
        k, pr = scan(text[i + 1:end + 1], name)
        total += k
        bad += pr
    print(f'{len(starts)} k_conv instantiations, {total} asm pixel loads checked, {len(bad)} problems')
    for b in bad[:40]:
        print('  ' + b)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.path.insert(0, ROOT)
    sys.exit(main())

"""Builds far_amd/lib/libfar_hip.so (hipcc, gfx950 only) and the oracle's C pieces.

In-tree build so that the .so travels to the GPU box with the repo snapshot.
Usage: python -m far_amd.build [--force]
"""
import hashlib
import os
import re
import subprocess
import sys

from far_amd import flags

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libfar_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
BASE_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result', '-Wno-unused-value']
EXTRA_FLAGS = (flags.value('FAR_EXTRA_HIPCC_FLAGS') or '').split()            # experiment builds (-DFAR_WINO_EXP=..., tools/)
FLAGS = BASE_FLAGS + ['-I', CSRC] + EXTRA_FLAGS


def flags_skip_asm_check():
    return flags.value('FAR_SKIP_ASM_CHECK') == '1'


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _digest(paths, extra=''):
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode() + b'\0')
        with open(p, 'rb') as f:
            h.update(f.read())
        h.update(b'\0')
    return h.hexdigest()


def source_id():
    """The build id: sha256 (first 16 hex digits) over every file of far_amd/csrc (names and contents) and the compiler flags.
    Compiled into the library (far_build_id()); _lib.load() recomputes it from the sources next to the library and refuses a
    library built from other sources -- the .so travels outside git, this is what ties it to the tree it is loaded from."""
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h', '.inc'))]
    return _digest(files, ' '.join(BASE_FLAGS + EXTRA_FLAGS))[:16]


# ---- K9's asm pixel loads (conv_igemm_f16s.hip: stage_load / stage_arrived) rely on one property of the GENERATED code: between an
# asm `global_load_dwordx4` and the first instruction that touches its destination registers there is an asm `s_waitcnt vmcnt(K)` with
# K <= the number of memory requests issued in between (the counter retires in order, so the load has landed by then).  A register
# copy or spill of a staged value scheduled above that wait would read a register the load has not written yet -- silently.  The
# build compiles the file to assembly as well and scans every k_conv instantiation; a violation fails the build.
ASM_CHECKED = {'conv_igemm_f16s.hip': 'k_conv'}


def _asm_regs(text):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r'\bv(\d+)\b', text))
    return out


def _asm_scan(lines, name):
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
        dest = _asm_regs(items[i][1].split(',')[0])
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
                if a and t.startswith('global_load_dwordx4') and _asm_regs(t.split(',')[0]) & dest:
                    problems.append(f'{name}: asm load #{i} re-issued into its registers before a wait ({t})')
                    break
                younger += 1
                continue
            if t.startswith('s_endpgm'):
                waited = True              # never read: the trailing loads of the last chunk (vmcnt(0) precedes every exit)
                break
            if _asm_regs(t) & dest:
                problems.append(f'{name}: `{t}` touches {sorted(_asm_regs(t) & dest)} of asm load `{items[i][1]}` before its wait')
                break
        if not waited and not problems:
            problems.append(f'{name}: no wait found after asm load `{items[i][1]}`')
    return len(loads), problems



def asm_check(path, kernel):
    """Scans the -S output `path` for the property above in every function whose name contains `kernel`.  -> (functions, loads, problems)"""
    text = open(path).read().splitlines()
    starts = [(i, l.split(':')[0]) for i, l in enumerate(text) if re.match(r'^_Z\w*' + kernel + r'\w*:', l)]
    total, bad = 0, []
    for i, name in starts:
        end = next((j for j in range(i, len(text)) if text[j].strip().startswith('s_endpgm')), len(text))
        k, pr = _asm_scan(text[i + 1:end + 1], name)
        total += k
        bad += pr
    return len(starts), total, bad


def build(force=False, verbose=True):
    """Objects are rebuilt when the CONTENT of their source, of any header, or the flags changed (a sidecar <obj>.hash records the
    digest they were built from -- modification times do not survive a checkout or a snapshot copy)."""
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.inc'))]
    bid = source_id()
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        is_abi = os.path.basename(src) == 'abi.hip'
        flags = FLAGS + ([f'-DFAR_BUILD_ID="{bid}"'] if is_abi else [])          # abi.hip carries the id: it recompiles with every change
        want = _digest([src] + hdrs, ' '.join(flags).replace(CSRC, '<csrc>'))
        side = obj + '.hash'
        have = open(side).read().strip() if os.path.exists(side) and os.path.exists(obj) else None
        if force or have != want:
            cmd = [HIPCC] + flags + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            chk = None
            if os.path.basename(src) in ASM_CHECKED and not flags_skip_asm_check():
                asm = obj[:-2] + '.s'
                chk = (asm, subprocess.Popen([HIPCC] + [f for f in flags if f != '-fPIC'] + ['--cuda-device-only', '-S', src, '-o', asm],
                                             stderr=subprocess.DEVNULL))
            procs.append((src, subprocess.Popen(cmd), side, want, chk))
    for src, p, side, want, chk in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
        if chk is not None:
            asm, pa = chk
            if pa.wait() != 0:
                raise RuntimeError(f'hipcc -S failed on {src}')
            nfn, nld, bad = asm_check(asm, ASM_CHECKED[os.path.basename(src)])
            os.remove(asm)
            if verbose:
                print(f'{os.path.basename(src)}: {nfn} {ASM_CHECKED[os.path.basename(src)]} instantiations, {nld} asm loads checked, {len(bad)} problems', flush=True)
            if bad or nld == 0:
                raise RuntimeError(f'{src}: the generated code touches a staged register before its wait (or no asm load was found):\n  ' + '\n  '.join(bad[:10]))
        with open(side, 'w') as f:
            f.write(want + '\n')
    for f in os.listdir(LIBDIR):                      # objects of sources that no longer exist
        if (f.endswith('.o') and os.path.join(LIBDIR, f) not in objs) or (f.endswith('.o.hash') and os.path.join(LIBDIR, f[:-5]) not in objs):
            os.remove(os.path.join(LIBDIR, f))
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)

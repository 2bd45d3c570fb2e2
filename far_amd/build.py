"""Builds far_amd/lib/libfar_hip.so (hipcc, gfx950 only) and the oracle's C pieces.

In-tree build so that the .so travels to the GPU box with the repo snapshot.
Usage: python -m far_amd.build [--force]
"""
import hashlib
import os
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
            procs.append((src, subprocess.Popen(cmd), side, want))
    for src, p, side, want in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
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

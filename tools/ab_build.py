"""Same-box A/B aid: builds far_amd/lib/libfar_hip_base.so from the sources of a git revision (default HEAD) next to the
working tree's library, so that a timing script can load either (FAR_HIP_LIB=far_amd/lib/libfar_hip_base.so).  python tools/ab_build.py [rev]"""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from far_amd import build as B  # noqa: E402

rev = sys.argv[1] if len(sys.argv) > 1 else 'HEAD'
B.build(verbose=False)
tmp = tempfile.mkdtemp(prefix='far_ab_')
names = subprocess.check_output(['git', 'ls-tree', '--name-only', rev, 'far_amd/csrc/'], text=True).split()
for n in names:
    open(os.path.join(tmp, os.path.basename(n)), 'w').write(subprocess.check_output(['git', 'show', f'{rev}:{n}'], text=True))
objs = []
for n in sorted(os.listdir(tmp)):
    if n.endswith('.hip'):
        o = os.path.join(tmp, n[:-4] + '.o')
        cur = os.path.join(B.CSRC, n)
        if os.path.exists(cur) and open(cur).read() == open(os.path.join(tmp, n)).read() and \
                all(open(os.path.join(B.CSRC, h)).read() == open(os.path.join(tmp, h)).read() for h in os.listdir(tmp) if h.endswith(('.h', '.inc'))):
            o = os.path.join(B.LIBDIR, n[:-4] + '.o')            # unchanged: reuse the working tree's object
        else:
            subprocess.check_call([B.HIPCC] + B.FLAGS + ['-I', tmp, '-c', os.path.join(tmp, n), '-o', o])
        objs.append(o)
# a base of another ABI would be called with this tree's argument lists (far_amd/_lib.py checks it at load: refuse here already)
import re  # noqa: E402
abi_of = lambda txt: int(re.search(r'far_abi_version\(void\) \{ return (\d+); \}', txt).group(1))
a_base, a_cur = abi_of(open(os.path.join(tmp, 'abi.hip')).read()), abi_of(open(os.path.join(B.CSRC, 'abi.hip')).read())
if a_base != a_cur:
    sys.exit(f'{rev} has ABI version {a_base}, the working tree {a_cur}: not loadable through these bindings')
out = os.path.join(B.LIBDIR, 'libfar_hip_base.so')
subprocess.check_call([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
print(out)

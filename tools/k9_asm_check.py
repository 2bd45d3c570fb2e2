"""Development aid: the scan of K9's generated code that its asm pixel loads rely on (far_amd/build.py: asm_check -- the build runs it
whenever conv_igemm_f16s.hip is recompiled; this script runs it on demand).

  python tools/k9_asm_check.py            # compiles the file to assembly (about 2 minutes) and scans every k_conv instantiation
  python tools/k9_asm_check.py file.s     # scans an existing -S output
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from far_amd import build  # noqa: E402

SRC = os.path.join(build.CSRC, 'conv_igemm_f16s.hip')
if len(sys.argv) > 1:
    path = sys.argv[1]
else:
    path = os.path.join(tempfile.gettempdir(), 'far_k9_check.s')
    subprocess.run([build.HIPCC] + [f for f in build.FLAGS if f != '-fPIC'] + ['--cuda-device-only', '-S', SRC, '-o', path], check=True,
                   stderr=subprocess.DEVNULL)
nfn, nld, bad = build.asm_check(path, 'k_conv')
print(f'{nfn} k_conv instantiations, {nld} asm pixel loads checked, {len(bad)} problems')
for b in bad[:40]:
    print('  ' + b)
sys.exit(1 if bad or nld == 0 else 0)

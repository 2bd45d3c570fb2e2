"""demo.py (counterpart of mp3d_loftr/demo.py, BASELINE configs[0]) as a child process on the GPU: `--synthetic --check` runs one
pair through matcher -> solver -> head -> solver(prior) -> head, prints the solver pose of the last round (demo.py:145-151) and has the
CPU oracle check the matcher stage.  Without a GPU the script exits 2 and computes nothing (covered on CPU in test_host_logic.py)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1500)
def test_demo_synthetic_pair_with_oracle_check():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'demo.py'), '--synthetic', '--check'], capture_output=True, text=True,
                       timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = r.stdout
    m = re.search(r'matches:\s*(\d+)\s+inliers:\s*(\d+)', out)
    assert m and int(m.group(1)) > 500 and int(m.group(2)) > 100, out[-1500:]
    pose = re.search(r'predicted pose is:\s*\n\s*(\[\[.*?\]\])', out, re.S)
    assert pose, out[-1500:]
    rt = np.array([[float(v) for v in row.replace('[', ' ').replace(']', ' ').split()] for row in pose.group(1).strip().split('\n')])
    assert rt.shape == (3, 4)
    R, t = rt[:, :3], rt[:, 3]
    assert abs(np.linalg.det(R) - 1) < 1e-3 and abs(np.linalg.norm(t) - 1) < 1e-3            # a rotation and a unit translation
    assert np.linalg.norm(R - np.eye(3)) < 0.05                                              # the synthetic pair: R = I, lateral t
    chk = re.search(r'oracle check: (\d+) of (\d+) oracle matches reproduced \((\d+) found\); max \|featmap0 - oracle\| = ([0-9.e+-]+)', out)
    assert chk, out[-1500:]
    same, ref, found, dev = int(chk.group(1)), int(chk.group(2)), int(chk.group(3)), float(chk.group(4))
    assert same >= 0.97 * ref and found <= 1.03 * ref and dev < 1e-3, chk.group(0)

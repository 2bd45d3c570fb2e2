"""Event-timed isolated kernel timings at bench shapes (prints a table; used for A/B of kernel variants)."""
import sys, json
sys.path.insert(0, '.')
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
from far_amd import _lib
for a in sys.argv[2:]:
    k, v = a.split('=')
    _lib.load().far_set_tuning(int(k), int(v))
    print('tuning', k, v)
r = bench.kernel_rooflines(n)
for k, v in r.items():
    print(f"{k:<62} " + "  ".join(f"{a}={b:.3f}" for a, b in v.items()))

"""Shared synthetic-input builders for the parity tests (seeded, no reference access)."""
import numpy as np


def correlated_features(N, hw, C, seed=0, amp=1.2, noise=0.1, frac=1.0):
    """f1 = f0[perm] + noise: gives a dense set of confident mutual matches."""
    rng = np.random.default_rng(seed)
    L = hw[0] * hw[1]
    f0 = (amp * rng.standard_normal((N, L, C))).astype(np.float32)
    f1 = np.empty_like(f0)
    perms = []
    for n in range(N):
        perm = rng.permutation(L)
        perms.append(perm)
        f1[n] = f0[n][perm]
        if frac < 1.0:
            k = int(L * (1 - frac))
            bad = rng.choice(L, k, replace=False)
            f1[n][bad] = (amp * rng.standard_normal((k, C))).astype(np.float32)
    f1 = (f1 + noise * rng.standard_normal(f1.shape)).astype(np.float32)
    return f0, f1, perms

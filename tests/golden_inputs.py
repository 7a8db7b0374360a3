"""Input generators shared by the oracle tests and the GPU parity tests.  They restate the
generators of tests/golden/make_golden.py (which runs only where the reference exists)."""
import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gen_events(seed, n, W, H, burst=False, same_ts=False, weights_p=False):
    if n == 0:
        z = np.zeros(0, np.float32)
        return z, z.copy(), z.copy(), z.copy()
    rng = np.random.default_rng(seed)
    t64 = np.sort(rng.uniform(0, n * 1e-6 + 1e-3, n))
    if same_ts:
        t64[:] = t64[0]
    x = rng.integers(0, W, n); y = rng.integers(0, H, n)
    if burst and n > 8:
        hot = rng.integers(0, n, n // 2)
        x[hot] = x[hot[0]] if n < 64 else rng.integers(0, 3, len(hot))
        y[hot] = y[hot[0]] if n < 64 else rng.integers(0, 2, len(hot))
    p = rng.integers(0, 2, n) * 2.0 - 1.0
    if weights_p:
        p = rng.normal(size=n)
    return (x.astype(np.float32), y.astype(np.float32), (t64 - t64[0]).astype(np.float32),
            p.astype(np.float32))

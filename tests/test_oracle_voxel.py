"""Pin the voxelizer oracle (numpy and C restatements) to the reference's golden vectors."""
import ctypes
import json
import os

import numpy as np
import pytest

from oracle import voxel as ov
from conftest import ROOT, load_npz, load_json
from golden_inputs import gen_events, sha


def c_oracle():
    path = os.path.join(ROOT, 'oracle', 'liboracle.so')
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    lib = ctypes.CDLL(path)
    lib.oracle_voxelize.restype = ctypes.c_int
    return lib


def c_voxelize(lib, x, y, t, p, offs, B, H, W):
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    out = np.empty((len(offs) - 1, B, H, W), dtype=np.float32)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(ctypes.c_void_p)
    rc = lib.oracle_voxelize(f(x), f(y), f(t), f(p), offs.ctypes.data_as(ctypes.c_void_p),
                             ctypes.c_int(len(offs) - 1), ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W),
                             out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out


def small_cases():
    z = load_npz('voxel_small.npz')
    meta = json.loads(bytes(z['meta']).decode())
    return z, meta


def test_numpy_oracle_bit_exact_small():
    z, meta = small_cases()
    for m in meta:
        n = m['name']
        v = ov.events_to_voxel(z[n + '.x'], z[n + '.y'], z[n + '.t'], z[n + '.p'], m['B'], (m['H'], m['W']))
        assert v.dtype == np.float32
        assert np.array_equal(v.view(np.uint32), z[n + '.voxel'].view(np.uint32)), n


def test_c_oracle_bit_exact_small():
    lib = c_oracle()
    z, meta = small_cases()
    for m in meta:
        n = m['name']
        x, y, t, p = (z[n + '.' + k] for k in 'xytp')
        v = c_voxelize(lib, x, y, t, p, [0, len(x)], m['B'], m['H'], m['W'])[0]
        assert np.array_equal(v.view(np.uint32), z[n + '.voxel'].view(np.uint32)), n


def test_oracles_bit_exact_large_hashes():
    lib = c_oracle()
    for c in load_json('voxel_large.json'):
        x, y, t, p = gen_events(c['seed'], c['n'], c['W'], c['H'], **c['flags'])
        assert sha(np.stack([x, y, t, p])) == c['in_sha'], c['name']
        v = c_voxelize(lib, x, y, t, p, [0, c['n']], c['B'], c['H'], c['W'])[0]
        assert sha(v) == c['out_sha'], c['name']
        assert int((v != 0).sum()) == c['nnz']
        if c['n'] <= 50000:
            vn = ov.events_to_voxel(x, y, t, p, c['B'], (c['H'], c['W']))
            assert sha(vn) == c['out_sha'], c['name']


def test_batch_form_and_empty_window():
    lib = c_oracle()
    x, y, t, p = gen_events(5, 1000, 48, 32)
    offs = [0, 0, 300, 300, 1000]
    a = ov.voxelize_windows(x, y, t, p, offs, 5, (32, 48))
    b = c_voxelize(lib, x, y, t, p, offs, 5, 32, 48)
    assert np.array_equal(a, b)
    assert not a[0].any() and not a[2].any()
    assert np.array_equal(a[1], ov.events_to_voxel(x[:300], y[:300], t[:300], p[:300], 5, (32, 48)))


def test_weights_sum_to_polarity():
    # each event spreads |p| over <=2 adjacent bins
    x, y, t, p = gen_events(11, 4000, 48, 32)
    v = ov.events_to_voxel(x, y, t, p, 5, (32, 48))
    assert abs(float(v.astype(np.float64).sum()) - float(p.astype(np.float64).sum())) < 1e-2

"""GPU parity: HIP tensorizer (through the C ABI) vs the oracle and the reference goldens -- BIT exact."""
import json

import numpy as np
import pytest
import torch

from conftest import load_npz, load_json
from golden_inputs import gen_events, sha

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vox():
    from evreal_amd.voxel import Voxelizer
    return Voxelizer()


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def run(vox, x, y, t, p, offs, B, H, W, stats=False):
    st = torch.zeros((len(offs) - 1, 3), dtype=torch.float64, device='cuda') if stats else None
    out = vox.voxelize(dev(x), dev(y), dev(t), dev(p), dev(np.asarray(offs, dtype=np.int64)), B, (H, W), stats=st)
    torch.cuda.synchronize()
    return (out.cpu().numpy(), st.cpu().numpy()) if stats else out.cpu().numpy()


def test_small_goldens_bit_exact(vox):
    z = load_npz('voxel_small.npz')
    for m in json.loads(bytes(z['meta']).decode()):
        n = m['name']
        x, y, t, p = (z[n + '.' + k] for k in 'xytp')
        v = run(vox, x, y, t, p, [0, len(x)], m['B'], m['H'], m['W'])[0]
        assert np.array_equal(v.view(np.uint32), z[n + '.voxel'].view(np.uint32)), n


def test_large_goldens_hash(vox):
    for c in load_json('voxel_large.json'):
        x, y, t, p = gen_events(c['seed'], c['n'], c['W'], c['H'], **c['flags'])
        v, st = run(vox, x, y, t, p, [0, c['n']], c['B'], c['H'], c['W'], stats=True)
        assert sha(v[0]) == c['out_sha'], c['name']
        assert int(st[0, 2]) == c['nnz']
        assert abs(st[0, 0] - c['sum']) < 1e-6 * max(1.0, c['abs_sum'])
        assert vox.dropped() == 0


def test_many_windows_vs_oracle(vox):
    from oracle import voxel as ov
    rng = np.random.default_rng(3)
    W, H, B = 346, 260, 5
    sizes = [0, 1, 2, 15000, 0, 777, 64, 65, 20000, 3, 0]
    xs, ys, ts, ps, offs = [], [], [], [], [0]
    for i, n in enumerate(sizes):
        x, y, t, p = gen_events(100 + i, n, W, H, burst=(i % 3 == 0), same_ts=(n == 3))
        xs.append(x); ys.append(y); ts.append(t); ps.append(p); offs.append(offs[-1] + n)
    x, y, t, p = map(np.concatenate, (xs, ys, ts, ps))
    got, st = run(vox, x, y, t, p, offs, B, H, W, stats=True)
    want = ov.voxelize_windows(x, y, t, p, offs, B, (H, W))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for w in range(len(sizes)):
        assert int(st[w, 2]) == int((want[w] != 0).sum())
        # {sum, sum of squares}: per-thread fp32 partials over a few dozen cells, folded in fp64 in a fixed order
        # (deterministic).  They feed eval.py:402-405, whose own fp32 torch.sum is only good to ~1e-6.
        w64 = want[w].astype(np.float64)
        np.testing.assert_allclose(st[w, 0], w64.sum(), rtol=0, atol=1e-6 * max(1.0, np.abs(w64).sum()))
        np.testing.assert_allclose(st[w, 1], (w64 ** 2).sum(), rtol=1e-6, atol=1e-12)


def test_raw_form_matches_dataset_path(vox):
    """evr_voxelize_raw on memmap-typed arrays == dataset.py:222-228,53-57 followed by the tensorizer."""
    from oracle import voxel as ov
    from evreal_amd import synth
    t, x, y, p = synth.poisson_events(5, 60000, 1.0e6, 346, 260)
    xy = np.stack([x, y], axis=1)
    offs = np.array([0, 15000, 30000, 30000, 60000], dtype=np.int64)
    out = vox.voxelize_raw(dev(xy), dev(t), dev(p), dev(offs), 5, (260, 346))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for w in range(4):
        a, b = int(offs[w]), int(offs[w + 1])
        if b > a:
            xs, ys, ts, ps = synth.window_events_f32(t, xy, p, a, b)
            want = ov.events_to_voxel(xs, ys, ts, ps, 5, (260, 346))
        else:
            want = np.zeros((5, 260, 346), np.float32)
        assert np.array_equal(got[w].view(np.uint32), want.view(np.uint32)), w


def test_raw_form_keeps_any_polarity_byte(vox):
    """The raw form's 8-B records carry the polarity BYTE (p = 2 pol - 1 is formed by the range kernel, dataset.py:227):
    files whose polarity column holds something other than 0 / 1 give what the reference's arithmetic gives."""
    from oracle import voxel as ov
    from evreal_amd import synth
    t, x, y, p = synth.poisson_events(9, 20000, 1.0e6, 346, 260)
    p = np.random.default_rng(1).choice(np.array([0, 1, 2, 7, 255], np.uint8), len(p))
    xy = np.stack([x, y], axis=1)
    offs = np.array([0, 20000], dtype=np.int64)
    got = vox.voxelize_raw(dev(xy), dev(t), dev(p), dev(offs), 5, (260, 346)).cpu().numpy()
    xs, ys, ts, ps = synth.window_events_f32(t, xy, p, 0, 20000)
    want = ov.events_to_voxel(xs, ys, ts, ps, 5, (260, 346))
    assert np.array_equal(got[0].view(np.uint32), want.view(np.uint32))


def test_out_of_range_events_are_dropped_and_counted(vox):
    x = np.array([1, 400, 3, -2], np.float32); y = np.array([1, 2, 300, 5], np.float32)
    t = np.array([0, 1e-3, 2e-3, 3e-3], np.float32); p = np.ones(4, np.float32)
    v = run(vox, x, y, t, p, [0, 4], 5, 260, 346)
    assert vox.dropped() == 3
    assert v[0, 0, 1, 1] == 1.0 and np.count_nonzero(v) == 1


def test_dropped_events_accumulate_across_calls_and_regrows():
    """The cumulative counter a frame loop polls once per sequence (evr_voxelize_dropped_total): it survives later clean
    calls and a workspace regrow, and raise_if_dropped() turns it into the reference's failure (index_put_ raises)."""
    from evreal_amd.voxel import Voxelizer
    vz = Voxelizer()
    assert vz.dropped_total() == 0
    x = np.array([1, 400, 3, -2], np.float32); y = np.array([1, 2, 300, 5], np.float32)
    t = np.array([0, 1e-3, 2e-3, 3e-3], np.float32); p = np.ones(4, np.float32)
    run(vz, x, y, t, p, [0, 4], 5, 260, 346)
    run(vz, x, y, t, p, [0, 4], 5, 260, 346)
    assert vz.dropped() == 3 and vz.dropped_total() == 6
    n = 200000                                            # a clean, much larger call: the workspace regrows
    rng = np.random.default_rng(0)
    xs = rng.integers(0, 346, n).astype(np.float32); ys = rng.integers(0, 260, n).astype(np.float32)
    ts = np.sort(rng.uniform(0, 1e-2, n)).astype(np.float32)
    old = vz.ws.data_ptr()
    run(vz, xs, ys, ts, np.ones(n, np.float32), [0, n], 5, 260, 346)
    assert vz.ws.data_ptr() != old
    assert vz.dropped() == 0 and vz.dropped_total() == 6
    with pytest.raises(IndexError, match='outside the sensor'):
        vz.raise_if_dropped()


def test_full_size_properties(vox):
    """BASELINE config size (346x260, 15k events, 5 bins), 256 windows: size-independent properties --
    linearity in p, per-window sum == sum(p), empty windows stay zero, determinism."""
    W, H, B, n, nw = 346, 260, 5, 15000, 256
    t, x, y, p = __import__('evreal_amd.synth', fromlist=['x']).poisson_events(11, n * nw, 1.0e6, W, H)
    offs = np.arange(nw + 1, dtype=np.int64) * n
    xs = x.astype(np.float32); ys = y.astype(np.float32)
    ts = np.concatenate([(t[a:a + n] - t[a]).astype(np.float32) for a in offs[:-1]])
    ps = (p * 2.0 - 1.0).astype(np.float32)
    a = run(vox, xs, ys, ts, ps, offs, B, H, W)
    b = run(vox, xs, ys, ts, ps, offs, B, H, W)
    assert np.array_equal(a, b)
    c = run(vox, xs, ys, ts, 2.0 * ps, offs, B, H, W)
    assert np.array_equal(c, 2.0 * a)            # exact: scaling by 2 commutes with fp32 rounding
    sums = a.astype(np.float64).reshape(nw, -1).sum(1)
    psum = ps.astype(np.float64).reshape(nw, n).sum(1)
    np.testing.assert_allclose(sums, psum, atol=2e-2)
    assert np.abs(a).max() <= 64


def test_clustered_events(vox):
    """Windows whose events sit in a few sensor rows (one range with 20k records = 80 ticket chunks, ranges around the
    one-chunk limit, cancelling +1/-1 pairs that return a cell to +0 before it is touched again, a full-sensor window
    behind them) stay bit-exact, statistics included."""
    from oracle import voxel as ov
    rng = np.random.default_rng(21)
    W, H, B = 346, 260, 5
    xs, ys, ts, ps, offs = [], [], [], [], [0]
    for i, (n, y_lo, y_hi) in enumerate([(20000, 100, 104), (700, 8, 12), (520, 0, 4), (3000, 250, 260), (9000, 0, 260)]):
        x = rng.integers(0, W, n); y = rng.integers(y_lo, y_hi, n)
        t = np.sort(rng.uniform(0, 2e-2, n)); t = (t - t[0]).astype(np.float32)
        p = rng.integers(0, 2, n) * 2.0 - 1.0
        if i == 2:                                # pairs on one pixel and identical timestamps: +v then -v
            x[1::2] = x[0::2]; y[1::2] = y[0::2]; t[1::2] = t[0::2]; p[1::2] = -p[0::2]
        xs.append(x.astype(np.float32)); ys.append(y.astype(np.float32)); ts.append(t); ps.append(p.astype(np.float32))
        offs.append(offs[-1] + n)
    x, y, t, p = map(np.concatenate, (xs, ys, ts, ps))
    got, st = run(vox, x, y, t, p, offs, B, H, W, stats=True)
    want = ov.voxelize_windows(x, y, t, p, offs, B, (H, W))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for w in range(len(offs) - 1):
        w64 = want[w].astype(np.float64)
        assert int(st[w, 2]) == int((want[w] != 0).sum())
        np.testing.assert_allclose(st[w, 0], w64.sum(), rtol=0, atol=1e-6 * max(1.0, np.abs(w64).sum()))
        np.testing.assert_allclose(st[w, 1], (w64 ** 2).sum(), rtol=1e-6, atol=1e-12)

"""Pin the pre/post-processing, dataset-window and aggregation oracles to the reference goldens."""
import numpy as np
import pytest

from oracle import prepost, dataset as ods, metrics as om
from evreal_amd import synth
from conftest import load_npz, load_json
from golden_inputs import gen_events, sha
from oracle import voxel as ov


def test_normalize_event_tensor():
    z = load_npz('normalize.npz')
    for k in ['zeros', 'single', 'dense']:
        out = prepost.normalize_event_tensor(z[k + '.in'])
        np.testing.assert_allclose(out, z[k + '.out'], rtol=2e-6, atol=2e-6)
    x, y, t, p = gen_events(int(z['vox15k.seed']), 15000, 346, 260)
    v = ov.events_to_voxel(x, y, t, p, 5, (260, 346))[None]
    out = prepost.normalize_event_tensor(v)
    np.testing.assert_allclose(out, z['vox15k.out'], rtol=2e-6, atol=2e-6)
    assert np.array_equal(out == 0, z['vox15k.out'] == 0)


def test_crop_table():
    for c in load_json('crop_table.json'):
        cp = prepost.CropParams(c['W'], c['H'], c['num_encoders'])
        assert [cp.padding_left, cp.padding_right, cp.padding_top, cp.padding_bottom] == c['pad']
        assert [cp.ix0, cp.ix1, cp.iy0, cp.iy1] == c['crop']
        x = np.zeros((1, 1, c['H'], c['W']), np.float32)
        assert list(cp.pad(x).shape[-2:]) == c['padded_shape']
        assert cp.crop(cp.pad(x)).shape == x.shape


def test_post_process_normalization():
    z = load_npz('robust_norm.npz')
    for k in ['unit', 'wide', 'small', 'ties']:
        for norm in ['robust', 'standard', 'exprobust']:
            out = prepost.post_process_normalization(z[k + '.in'].copy(), norm)
            assert out.dtype == np.float32
            assert np.array_equal(out, z[f'{k}.{norm}'], equal_nan=True), (k, norm)


def test_metric_tracker():
    for case in load_json('metric_tracker.json'):
        mt = om.MetricTracker()
        for k, v, c in case['updates']:
            mt.update(k, v, c)
        assert set(mt.data) == set(case['data'])
        for k, d in case['data'].items():
            assert mt.data[k]['count'] == d['count']
            assert mt.data[k]['total'] == d['total'] and mt.data[k]['average'] == d['average']
    assert om.mean_score([]) == -1
    assert om.mean_score([0.5, float('nan'), float('inf'), 1.5]) == 1.0


def test_dataset_windows(tmp_path):
    g = load_json('dataset_windows.json')
    s = g['seq']
    seq = synth.write_sequence(str(tmp_path), s['seed'], s['n_events'], s['rate_hz'], s['width'], s['height'], s['fps'])
    assert sha(seq['t']) == s['t_sha'] and sha(seq['xy']) == s['xy_sha'] and sha(seq['p']) == s['p_sha']
    assert sha(seq['images']) == s['images_sha']
    t, xy, p = seq['t'], seq['xy'], seq['p']
    frame_ts = [float(v) for v in seq['images_ts'][:, 0]]
    N = len(t)
    for name in ['between_frames', 'k_events', 'k_events_slide', 't_seconds', 't_seconds_slide']:
        c = g[name]; vm = c['voxel_method']
        if vm['method'] == 'k_events':
            table = ods.k_indices(N, vm['k'], vm['sliding_window_w'])
        elif vm['method'] == 't_seconds':
            table = ods.timeblock_indices(t, vm['t'], vm['sliding_window_t'])
        else:
            table = ods.frame_indices(seq['image_event_indices'])
        length = len(table) if vm['method'] != 'between_frames' else len(frame_ts) - 1
        assert length == c['length'], name
        for i, it in enumerate(c['items']):
            if 'raises' in it:
                with pytest.raises(ValueError):
                    ods.window_item(vm['method'], i, table, t, frame_ts, N, vm.get('t'))
                continue
            w = ods.window_item(vm['method'], i, table, t, frame_ts, N, vm.get('t'))
            assert (w['idx0'], w['idx1'], w['event_count']) == (it['idx0'], it['idx1'], it['event_count']), (name, i)
            assert w['dt'] == it['dt'] and w['voxel_timestamp'] == it['voxel_timestamp'], (name, i)
            assert frame_ts[w['frame_index']] == it['frame_timestamp']
            if i % 5 == 0:
                xs, ys, ts, ps = synth.window_events_f32(t, xy, p, w['idx0'], w['idx1'])
                if w['event_count'] > 0:
                    v = ov.events_to_voxel(xs, ys, ts, ps, 5, (s['height'], s['width']))
                else:
                    v = np.zeros((5, s['height'], s['width']), np.float32)
                assert sha(v) == it['voxel_sha'], (name, i)
                fr = (seq['images'][w['frame_index']][:, :, 0].astype(np.float32) / 255)[None]
                assert sha(fr.astype(np.float32)) == it['frame_sha']


def test_ssim_mse_properties():
    rng = np.random.default_rng(0)
    a = rng.random((64, 80)).astype(np.float32)
    assert om.mse(a, a) == 0.0
    assert abs(om.ssim(a, a) - 1.0) < 1e-6
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape).astype(np.float32), 0, 1)
    s = om.ssim(b, a)
    assert 0.0 < s < 1.0 and abs(om.ssim(a, b) - s) < 1e-6
    assert abs(om.mse(b, a) - float(np.mean((a.astype(np.float64) - b) ** 2))) < 1e-8


def test_ssim_mse_against_the_published_definition():
    """tests/golden/metrics_published.npz: SSIM exactly as Wang et al. (2004) / ssim_index.m define it (11x11 Gaussian,
    sigma 1.5, 'valid' region, K1 = 0.01, K2 = 0.03), evaluated in float64 with explicit window loops by
    make_golden.py -- independent of scipy's filters and of this repo's restatement.  scikit-image's float32 rounding
    is the only part of the reference's call that stays unpinned."""
    import json
    from conftest import load_npz
    from oracle import metrics as om
    z = load_npz('metrics_published.npz')
    for m in json.loads(bytes(z['meta']).decode()):
        x, y = z[m['name'] + '.x'], z[m['name'] + '.y']
        assert abs(om.ssim(x, y) - m['ssim']) < 1e-6, m['name']
        assert abs(om.mse(x, y) - m['mse']) < 1e-9, m['name']

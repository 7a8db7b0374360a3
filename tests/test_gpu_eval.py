"""GPU parity of the evaluation loop (dataset reader + window planner + frame loop + trackers + output files)
against the reference's own evaluate() runs (tests/golden/eval_loop.json: FireNet real weights, synthetic
sequences, between_frames / k_events / t_seconds configs incl. start/end-time gating and eval_infer_all;
tests/golden/eval_loop_e2vid.json: the E2VID registry branch with 'robust' post-normalisation and a pickled-ConfigParser
method, synthetic weights in the reference's checkpoint layouts)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_json, load_npz

pytestmark = pytest.mark.gpu


def _write_tree(root, g, model_path):
    from evreal_amd import synth
    for sub in ('eval', 'method', 'dataset'):
        os.makedirs(os.path.join(root, 'config', sub), exist_ok=True)
    for name, cfg in g['cfgs'].items():
        json.dump(cfg, open(os.path.join(root, 'config', 'eval', name + '.json'), 'w'))
    json.dump({"model_name": "FireNet", "model_path": model_path, "event_tensor_normalization": True,
               "post_process_norm": "none"}, open(os.path.join(root, 'config', 'method', 'FireNet.json'), 'w'))
    seqs = {}
    for name, (seed, n, rate, W, H, fps, st, en) in g['seqs'].items():
        synth.write_sequence(os.path.join(root, 'data', 'SYN', name), seed, n, rate, W, H, fps)
        seqs[name] = {} if st is None else {"start_time_s": st, "end_time_s": en}
    json.dump({"root_path": os.path.join(root, 'data', 'SYN'), "sequences": seqs},
              open(os.path.join(root, 'config', 'dataset', 'SYN.json'), 'w'))


def _parse(txt):
    return [(int(a), float(b)) for a, b in (l.split() for l in txt.strip().splitlines())] if txt.strip() else []


def _evaluate_and_compare(tmp_path, monkeypatch, batch_sequences):
    from evreal_amd import eval as ev
    monkeypatch.setenv('EVREAL_BATCH_SEQUENCES', str(batch_sequences))
    g = load_json('eval_loop.json')
    w = load_npz('firenet_weights.npz')
    ckpt = {'state_dict': {k: torch.from_numpy(w[k]) for k in w.files},
            'config': {'model': {'num_bins': 5, 'skip_type': 'no_skip', 'recurrent_block_type': 'convgru',
                                 'base_num_channels': 16, 'num_residual_blocks': 2,
                                 'recurrent_blocks': {'resblock': [0]}, 'kernel_size': 3,
                                 'final_activation': 'none', 'norm': 'none', 'BN_momentum': 0.01}}}
    model_path = str(tmp_path / 'firenet.pth')
    torch.save(ckpt, model_path)
    _write_tree(str(tmp_path), g, model_path)
    monkeypatch.chdir(tmp_path)
    res = ev.evaluate(['FireNet'], list(g['cfgs']), ['SYN'], ['mse', 'ssim'])
    for rel, want in g['files'].items():
        got = open(tmp_path / rel).read()
        name = os.path.basename(rel)
        if name in ('timestamps.txt', 'event_rate.txt'):
            assert got == want, rel                      # byte-identical
        else:
            a, b = _parse(got), _parse(want)
            assert [i for i, _ in a] == [i for i, _ in b], rel
            np.testing.assert_allclose([s for _, s in a], [s for _, s in b], rtol=0, atol=2e-5, err_msg=rel)
    for cfg, want in g['scores'].items():
        dm = res[cfg][0][0].data_dict
        for metric, d in want[0][0].items():
            assert dm[metric]['count'] == d['count'], (cfg, metric)
            assert abs(dm[metric]['average'] - d['average']) < 1e-5, (cfg, metric, dm[metric]['average'], d['average'])


def test_evaluate_matches_reference_run(tmp_path, monkeypatch):
    _evaluate_and_compare(tmp_path, monkeypatch, 1)


def test_evaluate_with_batched_sequences_matches_reference_run(tmp_path, monkeypatch):
    """--batch-sequences: the sequences of a dataset advance together, one batch slot each; every output file and
    score must still equal the reference's one-sequence-at-a-time run."""
    _evaluate_and_compare(tmp_path, monkeypatch, 3)


# ------------------------------------------------------------------------------------------------------------------
# The 'E2VID' registry branch (eval.py:141-144: checkpoint['model'] kwargs + sigmoid; config/method/E2VID.json:
# event_tensor_normalization + 'robust' post-normalisation inside the frame loop, eval.py:380-395) and a method whose
# checkpoint carries a pickled parse_config.ConfigParser (eval.py:149-151, the E2VID+ / HyperE2VID / FireNet+ / ET-Net
# form), against tests/golden/eval_loop_e2vid.json = the reference's own evaluate() on the same checkpoints
# (make_golden.py make_eval_e2vid: synthetic weights in the reference's checkpoint layouts, 70x50 -> padded 72x56).
def _write_e2vid_tree(root, g):
    import sys
    import types
    from evreal_amd import synth, weights
    for sub in ('eval', 'method', 'dataset'):
        os.makedirs(os.path.join(root, 'config', sub), exist_ok=True)
    for name, cfg in g['cfgs'].items():
        json.dump(cfg, open(os.path.join(root, 'config', 'eval', name + '.json'), 'w'))
    # the checkpoints, written as make_golden.py e2vid_eval_checkpoints writes them.  The reference's ConfigParser is not on
    # this box: an object of a same-named class in a module called parse_config pickles to the same GLOBAL reference.
    mod = types.ModuleType('parse_config')

    class ConfigParser:
        def __init__(self, config):
            self._config = config
    ConfigParser.__module__ = 'parse_config'
    ConfigParser.__qualname__ = 'ConfigParser'
    mod.ConfigParser = ConfigParser
    kw = {k: v for k, v in weights.E2VID_KWARGS.items() if k != 'final_activation'}
    sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=g['seeds']['E2VID'])
    kwp = dict(weights.E2VID_PLUS_KWARGS)
    sdp = weights.synth_state_dict(weights.unet_recurrent_schema(**kwp), seed=g['seeds']['E2VID+'])
    paths = {'E2VID': os.path.join(root, 'e2vid.pth'), 'E2VID+': os.path.join(root, 'e2vid_plus.pth'),
             'E2VID_big': os.path.join(root, 'e2vid_big.pth')}
    torch.save({'model': dict(kw), 'state_dict': {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}, paths['E2VID'])
    big = weights.rescale_encoder_conv(sd, enc=1, K=float(g['big_k']))      # the same function with one tensor beyond +-4094
    torch.save({'model': dict(kw), 'state_dict': {k: torch.from_numpy(np.asarray(v)) for k, v in big.items()}}, paths['E2VID_big'])
    had = sys.modules.get('parse_config')
    sys.modules['parse_config'] = mod
    try:
        torch.save({'config': ConfigParser({'arch': {'type': 'E2VIDRecurrent', 'args': {'unet_kwargs': kwp}}}),
                    'state_dict': {k: torch.from_numpy(np.asarray(v)) for k, v in sdp.items()}}, paths['E2VID+'])
    finally:
        if had is not None:
            sys.modules['parse_config'] = had
        else:
            del sys.modules['parse_config']
    for mname, mcfg in g['methods'].items():
        json.dump(dict(mcfg, model_path=paths[mname]), open(os.path.join(root, 'config', 'method', mname + '.json'), 'w'))
    seqs = {}
    for name, (seed, n, rate, W, H, fps, st, en) in g['seqs'].items():
        synth.write_sequence(os.path.join(root, 'data', 'SYN', name), seed, n, rate, W, H, fps)
        seqs[name] = {} if st is None else {"start_time_s": st, "end_time_s": en}
    json.dump({"root_path": os.path.join(root, 'data', 'SYN'), "sequences": seqs},
              open(os.path.join(root, 'config', 'dataset', 'SYN.json'), 'w'))


def _evaluate_e2vid_and_compare(tmp_path, monkeypatch, batch_sequences, methods):
    from evreal_amd import eval as ev
    monkeypatch.setenv('EVREAL_BATCH_SEQUENCES', str(batch_sequences))
    g = load_json('eval_loop_e2vid.json')
    _write_e2vid_tree(str(tmp_path), g)
    monkeypatch.chdir(tmp_path)
    order = g['method_order']
    res = ev.evaluate(methods, list(g['cfgs']), ['SYN'], ['mse', 'ssim'])
    assert len(g['files']) == 3 * 2 * 2 * 4
    for rel, want in g['files'].items():
        if rel.split(os.sep)[4] not in methods:
            continue
        got = open(tmp_path / rel).read()
        name = os.path.basename(rel)
        if name in ('timestamps.txt', 'event_rate.txt'):
            assert got == want, rel                      # byte-identical
        else:
            a, b = _parse(got), _parse(want)
            assert [i for i, _ in a] == [i for i, _ in b], rel
            np.testing.assert_allclose([s for _, s in a], [s for _, s in b], rtol=0, atol=2e-5, err_msg=rel)
    for cfg, want in g['scores'].items():
        for mi, mname in enumerate(methods):
            dm = res[cfg][mi][0].data_dict
            ref = want[order.index(mname)][0]
            assert set(dm) == set(ref), (cfg, mname)
            for metric, d in ref.items():
                assert dm[metric]['count'] == d['count'] > 0, (cfg, mname, metric)
                assert abs(dm[metric]['average'] - d['average']) < 2e-5, (cfg, mname, metric, dm[metric]['average'], d['average'])


def test_evaluate_e2vid_branch_matches_reference_run(tmp_path, monkeypatch):
    _evaluate_e2vid_and_compare(tmp_path, monkeypatch, 1, ['E2VID', 'E2VID+'])


def test_evaluate_e2vid_branch_batched_matches_reference_run(tmp_path, monkeypatch):
    _evaluate_e2vid_and_compare(tmp_path, monkeypatch, 2, ['E2VID', 'E2VID+'])


@pytest.mark.parametrize('batch_sequences', [1, 2])
def test_out_of_range_activations_are_rerun_in_exact_fp32(tmp_path, monkeypatch, capsys, batch_sequences):
    """Saturation must not change results.  'E2VID_big' is the E2VID checkpoint with one intermediate tensor 65536 times larger
    (weights.rescale_encoder_conv: the same function bit for bit in fp32 -- make_golden.py asserts the reference writes the same
    files for both): its activations leave the +-4094 range of the default arithmetic's H2 format.  evaluate() must notice before
    it books anything of the chunk, say so, re-run the sequences on the library's exact-fp32 kernels and produce the reference's
    files: timestamps / event rates byte-identical, scores within 2e-5."""
    from evreal_amd import eval as ev
    _evaluate_e2vid_and_compare(tmp_path, monkeypatch, batch_sequences, ['E2VID_big'])
    out = capsys.readouterr().out
    exact = bool(os.environ.get('EVR_FP32')) or os.environ.get('EVR_ARITH') == 'fp32'
    if exact:
        assert 're-running in exact fp32' not in out          # (the whole suite on the exact mode: nothing to re-run)
    elif os.environ.get('EVR_ARITH') == 'mx6':
        pass      # P6 groups carry their own scale: activations of 2e4 stay inside the format (the scores above held the gate either way)
    else:
        assert out.count('re-running in exact fp32') >= 2 and "layer 'enc1.conv'" in out, out[-2000:]      # once per eval config at least
    ev._MODEL_CACHE.clear()


def test_png_writers_async_equals_sync(tmp_path, monkeypatch):
    """save_images: frames are encoded and written by background threads (joined in finalize); the files must be the
    ones the synchronous path writes -- one per timestamp line, round(clip(img)*255) 8-bit gray (eval_utils.py:80-84)."""
    import hashlib
    from PIL import Image
    from evreal_amd import eval as ev
    g = load_json('eval_loop.json')
    w = load_npz('firenet_weights.npz')
    ckpt = {'state_dict': {k: torch.from_numpy(w[k]) for k in w.files},
            'config': {'model': {'num_bins': 5, 'skip_type': 'no_skip', 'recurrent_block_type': 'convgru',
                                 'base_num_channels': 16, 'num_residual_blocks': 2,
                                 'recurrent_blocks': {'resblock': [0]}, 'kernel_size': 3,
                                 'final_activation': 'none', 'norm': 'none', 'BN_momentum': 0.01}}}
    from evreal_amd import eval_metrics as em
    seen = {}
    orig = em.EvalMetricsTracker._save_pngs

    def spy(self, folder, indices, imgs, u8=None):          # what the tracker was asked to write, as host arrays
        for i, a in zip(indices, imgs.detach().cpu().numpy()):
            seen[os.path.join(folder, 'frame_{:010d}.png'.format(i))] = a.copy()
        if u8 is not None:          # the frame loop's own uint8 conversion must be the tracker's (eval_utils.py:83)
            want = torch.round(torch.clamp(imgs, 0.0, 1.0) * 255).to(torch.uint8).cpu().numpy()
            assert np.array_equal(np.asarray(u8), want)
        return orig(self, folder, indices, imgs, u8)
    monkeypatch.setattr(em.EvalMetricsTracker, '_save_pngs', spy)
    digests = {}
    # async / sync: the library's native writer pool (round 6), and the same encoder with a wait behind every call; pil: round 5's
    # pool of PIL writers (EVREAL_PNG_WRITER=pil) -- another container around the same pixels
    for mode in ('async', 'sync', 'pil'):
        root = tmp_path / mode
        os.makedirs(root)
        model_path = str(root / 'firenet.pth')
        torch.save(ckpt, model_path)
        g2 = json.loads(json.dumps(g))
        g2['cfgs'] = {'k3k': dict(g['cfgs']['k3k'], save_images=True)}
        _write_tree(str(root), g2, model_path)
        monkeypatch.chdir(root)
        if mode == 'sync':
            monkeypatch.setenv('EVREAL_PNG_THREADS', '0')
        if mode == 'pil':
            monkeypatch.delenv('EVREAL_PNG_THREADS')
            monkeypatch.setenv('EVREAL_PNG_WRITER', 'pil')
        ev.evaluate(['FireNet'], ['k3k'], ['SYN'], ['mse'])
        out = {}
        for d, _, files in os.walk(root / 'outputs'):
            if 'timestamps.txt' in files:
                idxs = [int(l.split()[0]) for l in open(os.path.join(d, 'timestamps.txt')).read().strip().splitlines()]
                pngs = sorted(f for f in files if f.endswith('.png'))
                assert pngs == ['frame_{:010d}.png'.format(i) for i in idxs] and pngs, d
                im = Image.open(os.path.join(d, pngs[-1])); im.load()
                assert im.mode == 'L' and im.size == (64, 48)
                for f in pngs:
                    out[os.path.relpath(os.path.join(d, f), root)] = hashlib.sha256(open(os.path.join(d, f), 'rb').read()).hexdigest()
                    # independent of the writer: the decoded file is the reference's formula on the frame it was handed
                    # (eval_utils.py:83 after the clip of eval_metrics.py:253): np.round(clip(img, 0, 1) * 255) as uint8
                    rel = os.path.relpath(os.path.join(d, f), root)
                    want = np.round(np.clip(seen[rel], 0.0, 1.0) * 255).astype(np.uint8)
                    assert np.array_equal(np.asarray(Image.open(os.path.join(d, f))), want), rel
        digests[mode] = out
        seen.clear()
    assert digests['async'] and digests['async'] == digests['sync']
    assert set(digests['pil']) == {k.replace('async', 'pil', 1) for k in digests['async']} or len(digests['pil']) == len(digests['async'])


def test_dataset_reader_matches_reference_tables(tmp_path):
    """MemMapDataset mirror: per-item indices/timestamps/dt and voxel grids vs the reference (dataset_windows.json)."""
    from evreal_amd import synth
    from evreal_amd.dataset import MemMapDataset
    from golden_inputs import sha
    g = load_json('dataset_windows.json')
    s = g['seq']
    synth.write_sequence(str(tmp_path), s['seed'], s['n_events'], s['rate_hz'], s['width'], s['height'], s['fps'])
    for name in ['between_frames', 'k_events', 'k_events_slide', 't_seconds', 't_seconds_slide']:
        c = g[name]
        ds = MemMapDataset(str(tmp_path), num_bins=5, voxel_method=dict(c['voxel_method']))
        assert len(ds) == c['length'], name
        mn, mx = ds.get_min_max_t()
        assert (float(mn), float(mx)) == (c['min_t'], c['max_t'])
        assert list(ds.sensor_resolution) == c['sensor_resolution']
        tb = ds.table()
        ok = [i for i, it in enumerate(c['items']) if 'raises' not in it]
        for i, it in enumerate(c['items']):
            if 'raises' in it:
                assert not tb['valid'][i]
                with pytest.raises(ValueError):
                    ds[i]
                continue
            assert (int(tb['idx0'][i]), int(tb['idx1'][i]), int(tb['event_count'][i])) == (it['idx0'], it['idx1'], it['event_count']), (name, i)
            assert tb['dt'][i] == it['dt'] and tb['voxel_timestamp'][i] == it['voxel_timestamp'], (name, i)
            assert tb['frame_timestamp'][i] == it['frame_timestamp'], (name, i)
        grid, _ = ds.voxel_batch(ok)                       # every window of the sequence in ONE launch
        frames = ds.frames(tb['frame_index'][ok]).cpu().numpy()
        grid = grid.cpu().numpy()
        for j, i in enumerate(ok):
            assert sha(grid[j]) == c['items'][i]['voxel_sha'], (name, i)       # bit-exact vs the reference
            assert sha(frames[j]) == c['items'][i]['frame_sha'], (name, i)
        item = ds[ok[-1]]
        assert item['event_count'] == c['items'][ok[-1]]['event_count'] and item['events'].shape == (5, s['height'], s['width'])


def test_bad_sequence_does_not_discard_the_healthy_ones_before_it(tmp_path, monkeypatch, capsys):
    """One sequence whose files do not validate (here: polarities stored as -1/+1 -> 255 after the uint8 cast) sits in the
    middle of a dataset evaluated with batch_sequences = 8.  The reference evaluates and counts every sequence before the
    failing one and stops the dataset there (eval.py:360-375); so must the batched loop: the sequences before it write
    their files and enter dataset_metrics, the ones after it do not run."""
    from evreal_amd import eval as ev, synth
    monkeypatch.setenv('EVREAL_BATCH_SEQUENCES', '8')
    g = load_json('eval_loop.json')
    w = load_npz('firenet_weights.npz')
    ckpt = {'state_dict': {k: torch.from_numpy(w[k]) for k in w.files},
            'config': {'model': {'num_bins': 5, 'skip_type': 'no_skip', 'recurrent_block_type': 'convgru',
                                 'base_num_channels': 16, 'num_residual_blocks': 2,
                                 'recurrent_blocks': {'resblock': [0]}, 'kernel_size': 3,
                                 'final_activation': 'none', 'norm': 'none', 'BN_momentum': 0.01}}}
    model_path = str(tmp_path / 'firenet.pth')
    torch.save(ckpt, model_path)
    g2 = {'cfgs': {'k3k': dict(g['cfgs']['k3k'], save_images=False)},
          'seqs': {f's{i}': [40 + i, 30000, 1.0e6, 64, 48, 200.0, None, None] for i in range(4)}}
    _write_tree(str(tmp_path), g2, model_path)
    bad = tmp_path / 'data' / 'SYN' / 's2' / 'events_p.npy'
    p = np.load(bad)
    np.save(bad, (p.astype(np.int8) * 2 - 1))                       # -1 / +1 instead of 0 / 1
    monkeypatch.chdir(tmp_path)
    res = ev.evaluate(['FireNet'], ['k3k'], ['SYN'], ['mse'])
    out = capsys.readouterr().out
    assert 'Exception while evaluating method FireNet on SYN dataset' in out and 'events_p.npy must hold 0/1' in out
    base = tmp_path / 'outputs' / 'k3k' / 'SYN'
    for ok_seq in ('s0', 's1'):
        assert (base / ok_seq / 'FireNet' / 'timestamps.txt').read_text().strip(), ok_seq
        assert (base / ok_seq / 'FireNet' / 'mse.txt').read_text().strip(), ok_seq
    assert not (base / 's3' / 'FireNet' / 'timestamps.txt').exists()          # the reference's dataset loop stops at the failure
    dm = res['k3k'][0][0]
    n01 = sum(len((base / q / 'FireNet' / 'mse.txt').read_text().strip().splitlines()) for q in ('s0', 's1'))
    assert dm.get_count('mse') == n01 > 0

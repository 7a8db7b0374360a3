#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Run only in the build container (needs /root/reference; nothing here travels to the GPU box
except the .npz/.json outputs):   python tests/golden/make_golden.py

The reference (ercanburak/EVREAL) is imported from /root/reference with in-memory stubs for
the third-party packages that are absent offline and that the hot path does not need
(torchvision, cv2, yachalk, ffmpeg, skimage, pyiqa).  Every vector is an (input, output) pair
of a reference function; inputs are either stored or regenerated from seeds by
evreal_amd.synth / evreal_amd.weights (digests are stored so drift is detected).
"""
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

for name in ['torchvision', 'torchvision.transforms', 'cv2', 'yachalk', 'ffmpeg', 'skimage',
             'skimage.metrics', 'pyiqa']:
    sys.modules[name] = types.ModuleType(name)
sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
sys.modules['yachalk'].chalk = types.SimpleNamespace(
    cyan=types.SimpleNamespace(bold=str), red=types.SimpleNamespace(bold=str),
    green=types.SimpleNamespace(bold=str), yellow=types.SimpleNamespace(bold=str), underline=str)
sys.modules['skimage.metrics'].mean_squared_error = None
sys.modules['skimage.metrics'].structural_similarity = None
sys.modules['pyiqa'].list_models = lambda: []

os.chdir(REF)
from utils.event_utils import events_to_voxel_torch          # noqa: E402
from utils.util import CropParameters                        # noqa: E402
from dataset import MemMapDataset                            # noqa: E402
import model as ref_model                                    # noqa: E402
import eval as ref_eval                                      # noqa: E402

from evreal_amd import synth, weights                        # noqa: E402

torch.manual_seed(0)
_orig_load = torch.load
torch.load = lambda f, map_location=None, **kw: _orig_load(f, map_location=map_location, weights_only=False)
torch.set_num_threads(1)   # deterministic reduction order for the fp32 goldens


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save_npz(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print('wrote', name, {k: getattr(v, 'shape', None) for k, v in arrs.items()})


def save_json(name, obj):
    with open(os.path.join(HERE, name), 'w') as f:
        json.dump(obj, f, indent=1)
    print('wrote', name)


def ref_voxel(xs, ys, ts, ps, B, H, W):
    return events_to_voxel_torch(torch.from_numpy(xs), torch.from_numpy(ys), torch.from_numpy(ts),
                                 torch.from_numpy(ps), B, sensor_size=(H, W)).numpy()


# ---------------------------------------------------------------- 1. voxelizer
def gen_events(seed, n, W, H, burst=False, same_ts=False, weights_p=False):
    rng = np.random.default_rng(seed)
    t64 = np.sort(rng.uniform(0, n * 1e-6 + 1e-3, n))
    if same_ts:
        t64[:] = t64[0]
    x = rng.integers(0, W, n); y = rng.integers(0, H, n)
    if burst and n > 8:           # duplicate-pixel bursts: many events on few pixels
        hot = rng.integers(0, n, n // 2)
        x[hot] = x[hot[0]] if n < 64 else rng.integers(0, 3, len(hot))
        y[hot] = y[hot[0]] if n < 64 else rng.integers(0, 2, len(hot))
    p = rng.integers(0, 2, n) * 2.0 - 1.0
    if weights_p:
        p = rng.normal(size=n)
    xs = x.astype(np.float32); ys = y.astype(np.float32)
    ts = (t64 - t64[0]).astype(np.float32); ps = p.astype(np.float32)
    return xs, ys, ts, ps


def make_voxel():
    small, meta = {}, []
    cases = [  # name, seed, n, W, H, B, flags
        ('n1', 1, 1, 48, 32, 5, {}),
        ('n2', 2, 2, 48, 32, 5, {}),
        ('n3_same_ts', 3, 3, 48, 32, 5, dict(same_ts=True)),
        ('n5_same_ts', 4, 5, 48, 32, 5, dict(same_ts=True)),
        ('n7_same_ts', 5, 7, 48, 32, 5, dict(same_ts=True)),
        ('n100', 6, 100, 48, 32, 5, {}),
        ('n5000_burst', 7, 5000, 48, 32, 5, dict(burst=True)),
        ('n3000_b3', 8, 3000, 48, 32, 3, {}),
        ('n3000_b10_weights', 9, 3000, 48, 32, 10, dict(weights_p=True)),
        ('n2000_b1', 10, 2000, 48, 32, 1, {}),
    ]
    for name, seed, n, W, H, B, fl in cases:
        xs, ys, ts, ps = gen_events(seed, n, W, H, **fl)
        v = ref_voxel(xs, ys, ts, ps, B, H, W)
        for k, a in zip('xytp', (xs, ys, ts, ps)):
            small[f'{name}.{k}'] = a
        small[f'{name}.voxel'] = v
        meta.append(dict(name=name, n=n, W=W, H=H, B=B))
    small_meta = json.dumps(meta)
    save_npz('voxel_small.npz', meta=np.frombuffer(small_meta.encode(), dtype=np.uint8), **small)

    large = []
    for name, seed, n, W, H, B, fl in [
            ('ecd_15k', 20, 15000, 240, 180, 5, {}),
            ('davis346_15k', 21, 15000, 346, 260, 5, {}),
            ('davis346_50k_burst', 22, 50000, 346, 260, 5, dict(burst=True)),
            ('vga_15k', 23, 15000, 640, 480, 5, {}),
            ('vga_200k', 24, 200000, 640, 480, 5, {}),
            ('bsergb_50k', 25, 50000, 970, 625, 5, {})]:
        xs, ys, ts, ps = gen_events(seed, n, W, H, **fl)
        v = ref_voxel(xs, ys, ts, ps, B, H, W)
        large.append(dict(name=name, seed=seed, n=n, W=W, H=H, B=B, flags=fl,
                          in_sha=sha(np.stack([xs, ys, ts, ps])), out_sha=sha(v),
                          nnz=int((v != 0).sum()), sum=float(v.astype(np.float64).sum()),
                          abs_sum=float(np.abs(v).astype(np.float64).sum())))
    save_json('voxel_large.json', large)


# ---------------------------------------------------------------- 2. dataset windows
def make_dataset():
    out = {}
    with tempfile.TemporaryDirectory() as d:
        seq = synth.write_sequence(d, seed=31, n_events=40000, rate_hz=2.0e5, width=48, height=32, fps=50.0)
        out['seq'] = dict(seed=31, n_events=40000, rate_hz=2.0e5, width=48, height=32, fps=50.0,
                          t_sha=sha(seq['t']), xy_sha=sha(seq['xy']), p_sha=sha(seq['p']),
                          images_sha=sha(seq['images']))
        methods = {
            'between_frames': {'method': 'between_frames'},
            'k_events': {'method': 'k_events', 'k': 3000, 'sliding_window_w': 0},
            'k_events_slide': {'method': 'k_events', 'k': 3000, 'sliding_window_w': 1000},
            't_seconds': {'method': 't_seconds', 't': 0.013, 'sliding_window_t': 0.0},
            't_seconds_slide': {'method': 't_seconds', 't': 0.02, 'sliding_window_t': 0.005},
        }
        for name, vm in methods.items():
            ds = MemMapDataset(d, num_bins=5, voxel_method=dict(vm))
            items = []
            for i in range(len(ds)):
                try:
                    if vm['method'] == 'between_frames':
                        prev = ds.frames_to_use[i - 1] if i > 0 else 0
                        idx0 = ds.get_event_indices(prev)[1]; idx1 = ds.get_event_indices(ds.frames_to_use[i])[1]
                    else:
                        idx0, idx1 = ds.get_event_indices(i)
                    it = ds[i]
                except ValueError:
                    # reference quirk: with sliding_window_w > 0 the table runs past num_events and
                    # get_event_indices raises (dataset.py:196-197)
                    items.append(dict(raises='ValueError'))
                    continue
                items.append(dict(idx0=int(idx0), idx1=int(idx1), event_count=int(it['event_count']),
                                  dt=float(it['dt']), voxel_timestamp=float(it['voxel_timestamp']),
                                  frame_timestamp=float(it['frame_timestamp']),
                                  voxel_sha=sha(it['events'].numpy()), frame_sha=sha(it['frame'].numpy())))
            mn, mx = ds.get_min_max_t()
            out[name] = dict(voxel_method=vm, length=len(ds), min_t=float(mn), max_t=float(mx),
                             sensor_resolution=list(ds.sensor_resolution), items=items)
    save_json('dataset_windows.json', out)


# ---------------------------------------------------------------- 3/4/7/8. small helpers
def make_helpers():
    rng = np.random.default_rng(41)
    cases = {}
    xs, ys, ts, ps = gen_events(42, 15000, 346, 260)
    v = ref_voxel(xs, ys, ts, ps, 5, 260, 346)[None]
    cases['vox15k'] = v
    cases['zeros'] = np.zeros((1, 5, 8, 12), np.float32)
    one = np.zeros((1, 5, 8, 12), np.float32); one[0, 2, 3, 4] = 0.75
    cases['single'] = one
    cases['dense'] = rng.normal(size=(1, 5, 16, 24)).astype(np.float32)
    arrs = {}
    for k, a in cases.items():
        arrs[k + '.in'] = a
        arrs[k + '.out'] = ref_eval.normalize_event_tensor(torch.from_numpy(a.copy())).numpy()
    save_npz('normalize.npz', **{k: v for k, v in arrs.items() if not k.startswith('vox15k.in')},
             **{'vox15k.seed': np.array(42)})

    table = []
    for (W, H, enc) in [(346, 260, 3), (240, 180, 4), (240, 180, 0), (240, 180, 3), (640, 480, 3),
                        (970, 625, 3), (485, 312, 3), (48, 32, 3), (100, 50, 2)]:
        c = CropParameters(W, H, enc)
        x = torch.zeros(1, 1, H, W)
        table.append(dict(W=W, H=H, num_encoders=enc, width_crop=c.width_crop_size, height_crop=c.height_crop_size,
                          pad=[c.padding_left, c.padding_right, c.padding_top, c.padding_bottom],
                          crop=[c.ix0, c.ix1, c.iy0, c.iy1], padded_shape=list(c.pad(x).shape[-2:])))
    save_json('crop_table.json', table)

    imgs = {}
    for k, a in [('unit', rng.random((100, 130)).astype(np.float32)),
                 ('wide', (rng.normal(size=(60, 80)) * 3).astype(np.float32)),
                 ('small', rng.random((7, 9)).astype(np.float32)),
                 ('ties', np.round(rng.random((64, 64)) * 8).astype(np.float32) / 8)]:
        imgs[k + '.in'] = a
        for norm in ['robust', 'standard', 'exprobust']:
            o = ref_eval.post_process_normalization(a.copy(), norm)
            imgs[f'{k}.{norm}'] = o
            assert o.dtype == np.float32, o.dtype
    save_npz('robust_norm.npz', **imgs)

    mt_cases = []
    for upd in [[('mse', 0.05, 10), ('mse', 0.10, 30)],
                [('mse', 0.05, 0), ('mse', 0.2, 3), ('ssim', -1, 4), ('ssim', 0.5, 4)],
                [('lpips', 0.31, 7)]]:
        mt = ref_eval.MetricTracker()
        for k, v, c in upd:
            mt.update(k, v, c)
        mt_cases.append(dict(updates=upd, data=mt.data_dict))
    save_json('metric_tracker.json', mt_cases)


# ---------------------------------------------------------------- 5. FireNet / FireNet+ (real weights)
def voxel_sequence(seed, n_frames, B, H, W, density=0.07):
    """Voxel-like sparse model inputs straight from a seed (evreal_amd.synth.sparse_voxels)."""
    return synth.sparse_voxels(seed, n_frames, B, H, W, density)


def make_firenet():
    for nm, tag in [('FireNet', 'firenet'), ('FireNet+', 'firenetplus')]:
        path = f'{REF}/pretrained/{nm}/model.pth'
        m = ref_eval.get_model_from_checkpoint_path(nm, path)
        sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
        save_npz(f'{tag}_weights.npz', **sd)
        H, W = 90, 120
        crop = CropParameters(W, H, m.num_encoders)
        vox = voxel_sequence(51, 5, 5, H, W)
        m.reset_states()
        outs = []
        with torch.no_grad():
            for f in range(len(vox)):
                x = crop.pad(torch.from_numpy(vox[f:f + 1]))
                outs.append(crop.crop(m(x)['image']).numpy())
        states = m.states if nm == 'FireNet' else m._states
        save_npz(f'{tag}_seq.npz', voxel_sha=np.array(sha(vox)), voxel_args=np.array([51, 5, 5, H, W]),
                 images=np.concatenate(outs),
                 state0_sub=states[0].numpy()[:, :, ::3, ::3].copy(), state1_sub=states[1].numpy()[:, :, ::3, ::3].copy(),
                 state0_sha=np.array(sha(states[0].numpy())),
                 num_encoders=np.array(m.num_encoders), weights_sha=np.array(weights.state_dict_digest(sd)))


# ---------------------------------------------------------------- 6. E2VID layouts (synthetic weights)
def make_e2vid():
    only = os.environ.get('E2VID_ONLY')        # regenerate one layout without touching the others
    for tag, kw in [('e2vid_bn', weights.E2VID_KWARGS), ('e2vid_plus', weights.E2VID_PLUS_KWARGS),
                    ('e2vid_hyper', dict(num_bins=5, base_num_channels=32, num_encoders=3, num_residual_blocks=2,
                                         kernel_size=5, norm=None, use_upsample_conv=True, recurrent_block_type='convlstm',
                                         skip_type='sum', final_activation='none', use_dynamic_decoder=True)),
                    ('e2vid_gru_tiny', dict(num_bins=5, base_num_channels=32, num_encoders=2, num_residual_blocks=1,
                                            kernel_size=5, norm=None, use_upsample_conv=False,
                                            recurrent_block_type='convgru', skip_type='sum',
                                            final_activation='sigmoid')),
                    ('e2vid_in', dict(num_bins=5, base_num_channels=32, num_encoders=3, num_residual_blocks=2,
                                      kernel_size=5, norm='IN', use_upsample_conv=False, recurrent_block_type='convlstm',
                                      skip_type='sum', final_activation='sigmoid')),
                    # the dynamic decoder beside IN layers: DynamicUpsampleLayer itself carries no norm (submodules.py:100-127)
                    ('e2vid_hyper_in', dict(num_bins=5, base_num_channels=32, num_encoders=3, num_residual_blocks=2,
                                            kernel_size=5, norm='IN', use_upsample_conv=True,
                                            recurrent_block_type='convlstm', skip_type='sum', final_activation='none',
                                            use_dynamic_decoder=True))]:
        if only and tag != only:
            continue
        schema = weights.unet_recurrent_schema(**kw)
        m = ref_model.E2VIDRecurrent(dict(kw))
        ref_sd = m.state_dict()
        fixed = {k: v.numpy() for k, v in ref_sd.items() if k.endswith('.bases')}   # Fourier-Bessel table (data)
        sd = weights.synth_state_dict(schema, seed=7, fixed=fixed)
        assert list(ref_sd.keys()) == list(sd.keys()), (set(ref_sd) ^ set(sd))
        for k in ref_sd:
            assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        m.eval()
        H, W = 64, 96
        vox = voxel_sequence(61, 4, 5, H, W)
        outs, taps = [], {}
        hooks = []
        if tag not in ('e2vid_gru_tiny',):
            u = m.unetrecurrent
            def tap(name):
                def hook(mod, inp, out):
                    taps.setdefault(name, (out[0] if isinstance(out, tuple) else out).detach().numpy().copy())
                    return None
                return hook
            hooks.append(u.head.register_forward_hook(tap('head')))
            hooks.append(u.encoders[0].conv.register_forward_hook(tap('enc0.conv')))
            hooks.append(u.encoders[0].register_forward_hook(tap('enc0.h')))
            hooks.append(u.encoders[2].register_forward_hook(tap('enc2.h')))
            hooks.append(u.resblocks[1].register_forward_hook(tap('res1')))
            hooks.append(u.decoders[0].register_forward_hook(tap('dec0')))
            hooks.append(u.decoders[2].register_forward_hook(tap('dec2')))
        with torch.no_grad():
            for f in range(len(vox)):
                outs.append(m(torch.from_numpy(vox[f:f + 1]))['image'].numpy())
                for h in hooks:
                    h.remove()
                hooks = []
        st = m.unetrecurrent.states
        extra = {}
        if kw.get('norm') == 'IN':
            # Conditioning of the layout: the residual blocks' true InstanceNorm2d divides by the deviation of an 8 x 12 map, which
            # magnifies fp32 rounding.  The reference class evaluated in float64 on the same inputs says by how much ITS OWN fp32
            # result is uncertain at every tap; the parity tests allow a small multiple of that, never less than their usual gate.
            import copy
            m64 = copy.deepcopy(m).double()
            m64.reset_states()
            taps64, hooks64 = {}, []
            u64 = m64.unetrecurrent
            def tap64(name):
                def hook(mod, inp, out):
                    taps64.setdefault(name, (out[0] if isinstance(out, tuple) else out).detach().numpy().copy())
                return hook
            for name, mod in (('head', u64.head), ('enc0.conv', u64.encoders[0].conv), ('enc0.h', u64.encoders[0]),
                              ('enc2.h', u64.encoders[2]), ('res1', u64.resblocks[1]), ('dec0', u64.decoders[0]),
                              ('dec2', u64.decoders[2])):
                hooks64.append(mod.register_forward_hook(tap64(name)))
            with torch.no_grad():
                img64 = []
                for f in range(len(vox)):
                    img64.append(m64(torch.from_numpy(vox[f:f + 1]).double())['image'].numpy())
                    for h in hooks64:
                        h.remove()
                    hooks64 = []
            extra['cond.images'] = np.array(np.abs(np.concatenate(img64) - np.concatenate(outs)).max())
            for k, v in taps.items():
                extra['cond.' + k] = np.array(np.abs(taps64[k] - v).max())
            print(tag, 'fp32-vs-fp64 spread of the reference:', {k: float(v) for k, v in extra.items()})
        for i, s in enumerate(st):
            if isinstance(s, tuple):
                extra[f'h{i}_sub'] = s[0].numpy()[:, ::4].copy(); extra[f'c{i}_sub'] = s[1].numpy()[:, ::4].copy()
            else:
                extra[f'h{i}_sub'] = s.numpy()[:, ::4].copy()
        save_npz(f'{tag}_seq.npz', voxel_sha=np.array(sha(vox)), voxel_args=np.array([61, 4, 5, H, W]),
                 images=np.concatenate(outs),
                 kwargs=np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8), seed=np.array(7),
                 weights_sha=np.array(weights.state_dict_digest(sd)),
                 **{'fixed.' + k: v for k, v in fixed.items()},
                 **{'tap.' + k: (v[:, ::4].copy() if v.shape[1] >= 32 else v) for k, v in taps.items()}, **extra)


# ---------------------------------------------------------------- 9. the evaluation loop itself
EVAL_SEQS = {   # name: (seed, n_events, rate_hz, W, H, fps, start_time_s, end_time_s)
    'seqA': (71, 40000, 2.0e5, 64, 48, 50.0, None, None),
    'seqB': (72, 30000, 2.0e5, 64, 48, 50.0, 0.03, 0.12),
}
EVAL_CFGS = {
    'std': {"dataset_kwargs": {"num_bins": 5, "voxel_method": {"method": "between_frames"}, "keep_ratio": 1.0},
            "save_images": False, "histeq": "none", "eval_infer_all": False, "ts_tol_ms": 1.0, "create_video": False},
    'k3k': {"dataset_kwargs": {"num_bins": 5, "voxel_method": {"method": "k_events", "k": 3000, "sliding_window_w": 0},
                               "keep_ratio": 1.0},
            "save_images": False, "histeq": "none", "eval_infer_all": False, "ts_tol_ms": 3.0, "create_video": False},
    't13ms': {"dataset_kwargs": {"num_bins": 5, "voxel_method": {"method": "t_seconds", "t": 0.013, "sliding_window_t": 0},
                                 "keep_ratio": 1.0},
              "save_images": False, "histeq": "none", "eval_infer_all": True, "ts_tol_ms": 4.0, "create_video": False},
}


def write_eval_tree(root, model_path):
    """config/ + data/ tree used by both the reference run (here) and the GPU test."""
    for sub in ('eval', 'method', 'dataset'):
        os.makedirs(os.path.join(root, 'config', sub), exist_ok=True)
    for name, cfg in EVAL_CFGS.items():
        json.dump(cfg, open(os.path.join(root, 'config', 'eval', name + '.json'), 'w'))
    json.dump({"model_name": "FireNet", "model_path": model_path, "event_tensor_normalization": True,
               "post_process_norm": "none"}, open(os.path.join(root, 'config', 'method', 'FireNet.json'), 'w'))
    seqs = {}
    for name, (seed, n, rate, W, H, fps, st, en) in EVAL_SEQS.items():
        synth.write_sequence(os.path.join(root, 'data', 'SYN', name), seed, n, rate, W, H, fps)
        seqs[name] = {} if st is None else {"start_time_s": st, "end_time_s": en}
    json.dump({"root_path": os.path.join(root, 'data', 'SYN'), "sequences": seqs},
              open(os.path.join(root, 'config', 'dataset', 'SYN.json'), 'w'))


def make_eval():
    import contextlib
    import utils.eval_metrics as em
    from oracle import metrics as omet
    # stand-ins for the absent scikit-image (parity unpinned): the oracle restatement
    em.mse = lambda ref, img: omet.mse(img, ref)
    em.ssim = lambda ref, img, **kw: omet.ssim(img, ref, sigma=kw.get('sigma', 1.5), data_range=kw.get('data_range', 1.0))
    sys.modules['cv2'].imwrite = lambda path, img: True

    class NoTimer(contextlib.ContextDecorator):
        def __init__(self, *a, **k): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
    ref_eval.CudaTimer = NoTimer
    captured = {}
    ref_eval.print_scores = lambda all_metrics, methods, dsets, cfg: captured.__setitem__(
        cfg, [[dm.data_dict for dm in mm] for mm in all_metrics])
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        write_eval_tree(d, f'{REF}/pretrained/FireNet/model.pth')
        os.chdir(d)
        sys.path.insert(0, REF)
        try:
            ref_eval.evaluate(['FireNet'], list(EVAL_CFGS), ['SYN'], ['mse', 'ssim'])
        finally:
            os.chdir(cwd)
        files = {}
        for base, _, fs in os.walk(os.path.join(d, 'outputs')):
            for f in fs:
                if f.endswith('.txt'):
                    rel = os.path.relpath(os.path.join(base, f), d)
                    files[rel] = open(os.path.join(base, f)).read()
    save_json('eval_loop.json', dict(seqs=EVAL_SEQS, cfgs=EVAL_CFGS, files=files, scores=captured))


# ---------------------------------------------------------------- 9b. the 'E2VID' registry branch and a pickled-ConfigParser method, end to end
# eval.py:141-144 (checkpoint['model'] kwargs + final_activation = 'sigmoid') with config/method/E2VID.json's
# event_tensor_normalization = true and post_process_norm = 'robust' (eval.py:380-395) INSIDE the frame loop, and
# eval.py:149-151 (checkpoint['config'] is a pickled parse_config.ConfigParser; config.init_obj('arch', model_arch)) with
# config/method/E2VID+.json's settings.  The E2VID / E2VID+ blobs are absent (.MISSING_LARGE_BLOBS): the checkpoints hold
# evreal_amd.weights' deterministic synthetic weights in the reference's own checkpoint layouts.  70x50 sensor: the
# cropper pads to 72x56 and crops back (utils/util.py:30-59).
E2VID_EVAL_SEQS = {   # name: (seed, n_events, rate_hz, W, H, fps, start_time_s, end_time_s)
    'seqC': (81, 36000, 2.0e5, 70, 50, 50.0, None, None),
    'seqD': (82, 27000, 2.0e5, 70, 50, 50.0, 0.03, 0.10),
}
E2VID_EVAL_CFGS = {k: EVAL_CFGS[k] for k in ('std', 'k3k')}
E2VID_METHODS = {
    'E2VID': {"model_name": "E2VID", "event_tensor_normalization": True, "post_process_norm": "robust"},
    'E2VID+': {"model_name": "E2VID+", "event_tensor_normalization": False, "post_process_norm": "none"},
    # the E2VID checkpoint with ONE intermediate tensor 65536 times larger (weights.rescale_encoder_conv: the same function, bit for
    # bit in fp32) -- activations beyond +-4094: what a split-precision drop-in must still get right
    'E2VID_big': {"model_name": "E2VID", "event_tensor_normalization": True, "post_process_norm": "robust"},
}
E2VID_EVAL_SEEDS = {'E2VID': 21, 'E2VID+': 22, 'E2VID_big': 21}
E2VID_BIG_K = 65536.0


def e2vid_eval_checkpoints(root, config_parser_cls):
    """Writes the two checkpoints in the layouts the reference's registry parses; returns {method: path}."""
    out = {}
    kw = {k: v for k, v in weights.E2VID_KWARGS.items() if k != 'final_activation'}        # eval.py:143 sets it
    sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=E2VID_EVAL_SEEDS['E2VID'])
    out['E2VID'] = os.path.join(root, 'e2vid.pth')
    torch.save({'model': dict(kw), 'state_dict': {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}, out['E2VID'])
    big = weights.rescale_encoder_conv(sd, enc=1, K=E2VID_BIG_K)
    out['E2VID_big'] = os.path.join(root, 'e2vid_big.pth')
    torch.save({'model': dict(kw), 'state_dict': {k: torch.from_numpy(np.asarray(v)) for k, v in big.items()}}, out['E2VID_big'])
    kwp = dict(weights.E2VID_PLUS_KWARGS)
    sdp = weights.synth_state_dict(weights.unet_recurrent_schema(**kwp), seed=E2VID_EVAL_SEEDS['E2VID+'])
    out['E2VID+'] = os.path.join(root, 'e2vid_plus.pth')
    torch.save({'config': config_parser_cls({'arch': {'type': 'E2VIDRecurrent', 'args': {'unet_kwargs': kwp}}}),
                'state_dict': {k: torch.from_numpy(np.asarray(v)) for k, v in sdp.items()}}, out['E2VID+'])
    return out


def make_eval_e2vid():
    import contextlib
    import utils.eval_metrics as em
    from oracle import metrics as omet
    from parse_config import ConfigParser           # the reference's own class: what its checkpoints pickle
    em.mse = lambda ref, img: omet.mse(img, ref)
    em.ssim = lambda ref, img, **kw: omet.ssim(img, ref, sigma=kw.get('sigma', 1.5), data_range=kw.get('data_range', 1.0))
    sys.modules['cv2'].imwrite = lambda path, img: True

    class NoTimer(contextlib.ContextDecorator):
        def __init__(self, *a, **k): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
    ref_eval.CudaTimer = NoTimer
    captured = {}
    ref_eval.print_scores = lambda all_metrics, methods, dsets, cfg: captured.__setitem__(
        cfg, [[dm.data_dict for dm in mm] for mm in all_metrics])
    cwd = os.getcwd()
    methods = list(E2VID_METHODS)
    with tempfile.TemporaryDirectory() as d:
        for sub in ('eval', 'method', 'dataset'):
            os.makedirs(os.path.join(d, 'config', sub), exist_ok=True)
        for name, cfg in E2VID_EVAL_CFGS.items():
            json.dump(cfg, open(os.path.join(d, 'config', 'eval', name + '.json'), 'w'))
        paths = e2vid_eval_checkpoints(d, ConfigParser)
        for mname, mcfg in E2VID_METHODS.items():
            json.dump(dict(mcfg, model_path=paths[mname]), open(os.path.join(d, 'config', 'method', mname + '.json'), 'w'))
        seqs = {}
        for name, (seed, n, rate, W, H, fps, st, en) in E2VID_EVAL_SEQS.items():
            synth.write_sequence(os.path.join(d, 'data', 'SYN', name), seed, n, rate, W, H, fps)
            seqs[name] = {} if st is None else {"start_time_s": st, "end_time_s": en}
        json.dump({"root_path": os.path.join(d, 'data', 'SYN'), "sequences": seqs},
                  open(os.path.join(d, 'config', 'dataset', 'SYN.json'), 'w'))
        os.chdir(d)
        try:
            ref_eval.evaluate(methods, list(E2VID_EVAL_CFGS), ['SYN'], ['mse', 'ssim'])
        finally:
            os.chdir(cwd)
        files = {}
        for base, _, fs in os.walk(os.path.join(d, 'outputs')):
            for f in fs:
                if f.endswith('.txt'):
                    rel = os.path.relpath(os.path.join(base, f), d)
                    files[rel] = open(os.path.join(base, f)).read()
    assert files and all(captured[c][i][0] for c in captured for i in range(len(methods))), 'the reference run scored nothing'
    # the rescaled network IS the same function in the reference's fp32: its files are the E2VID files, byte for byte
    for rel, txt in files.items():
        if '/E2VID_big/' in rel:
            assert txt == files[rel.replace('/E2VID_big/', '/E2VID/')], rel
    save_json('eval_loop_e2vid.json', dict(seqs=E2VID_EVAL_SEQS, cfgs=E2VID_EVAL_CFGS, methods=E2VID_METHODS, method_order=methods,
                                           seeds=E2VID_EVAL_SEEDS, big_k=E2VID_BIG_K, files=files, scores=captured))


# ---------------------------------------------------------------- 10. ColorNet streams
COLOR_EDGE_TOL = 1e-3      # in uint8 codes: 4e-6 of the image range


def make_color():
    import model.model as mm
    kw = dict(weights.E2VID_PLUS_KWARGS)
    sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=9)
    base = ref_model.E2VIDRecurrent(dict(kw))
    base.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    base.eval()
    captured = []
    mm.merge_channels_into_color_image = lambda ch: (captured.append({k: v.copy() for k, v in ch.items()}),
                                                     np.zeros(ch['grayscale'].shape + (3,), np.uint8))[1]
    mm.transforms.functional = types.SimpleNamespace(to_tensor=lambda a: torch.zeros(3, *a.shape[:2]))
    # model.py:101 truncates: np.clip(img * 255, 0, 255).astype(np.uint8).  A truncation is discontinuous: two fp32 evaluations of the
    # same network that differ in the last bits (a summation order, a thread count) disagree by one code wherever img * 255 sits on
    # an integer edge.  Record the reference's OWN float argument of that clip, to store where those pixels are.
    floats = []

    class _NpSpy:
        def __getattr__(self, name):
            return getattr(np, name)

        def clip(self, a, lo, hi):
            floats.append(np.array(a, dtype=np.float32, copy=True))
            return np.clip(a, lo, hi)
    mm.np = _NpSpy()
    net = mm.ColorNet(base)
    H, W, F = 96, 128, 3
    vox = synth.sparse_voxels(91, F, 5, H, W)
    try:
        with torch.no_grad():
            for f in range(F):
                net(torch.from_numpy(vox[f:f + 1]))
    finally:
        mm.np = np
    out = {}
    names = list(net.channels)          # R, G, B, W, grayscale: the order ColorNet.forward visits them (model.py:54-58)
    assert len(floats) == F * len(names)
    for f, ch in enumerate(captured):
        for ci, k in enumerate(names):
            v = ch[k]
            out[f'f{f}.{k}'] = v
            x = floats[f * len(names) + ci]
            assert np.array_equal(np.clip(x, 0, 255).astype(np.uint8), v)
            # pixels whose value the reference itself computed within COLOR_EDGE_TOL of a truncation edge (an integer in (0, 255]):
            # the only ones where an implementation that matches the floats to 1e-4 / 255 may land on the neighbouring code
            near = (np.abs(x - np.rint(x)) < COLOR_EDGE_TOL) & (x > 0.5) & (x < 255.5)
            out[f'f{f}.{k}.edge'] = np.packbits(near)
    save_npz('colornet_seq.npz', voxel_sha=np.array(sha(vox)), voxel_args=np.array([91, F, 5, H, W]), seed=np.array(9),
             edge_tol=np.array(COLOR_EDGE_TOL),
             kwargs=np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8),
             weights_sha=np.array(weights.state_dict_digest(sd)), **out)


def make_metrics_published():
    """Known-answer vectors for MSE and SSIM from the PUBLISHED definitions, independent of oracle/metrics.py:

    SSIM: Z. Wang, A. C. Bovik, H. R. Sheikh, E. P. Simoncelli, "Image quality assessment: from error visibility to
    structural similarity", IEEE TIP 13(4), 2004, and the authors' ssim_index.m: an 11x11 circular-symmetric Gaussian
    window with sigma = 1.5 samples, normalised to unit sum; local means, variances and covariance by 'valid' filtering
    (no padding: the map is (H-10) x (W-10)); K1 = 0.01, K2 = 0.03, L = dynamic range;
    SSIM = mean over the map of ((2 mu_x mu_y + C1)(2 sigma_xy + C2)) / ((mu_x^2 + mu_y^2 + C1)(sigma_x^2 + sigma_y^2 + C2)).
    Evaluated here in float64 with an explicit double loop over the window (no scipy filter, no separability).
    The reference's call -- structural_similarity(gaussian_weights=True, sigma=1.5, use_sample_covariance=False,
    data_range=1.0), utils/eval_metrics.py:96 -- is scikit-image's implementation of exactly this configuration (its
    docstring: "to match the implementation of Wang et al."), so these values pin the restatement to the published
    algorithm; scikit-image's own float32 rounding stays unpinned (it is not installed here).
    MSE: mean((x - y)^2) in float64."""
    rng = np.random.default_rng(2004)
    H, W = 48, 64
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    cases = {}
    base = 0.5 + 0.25 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + 0.15 * np.sin((xx + 2 * yy) / 11.0)
    cases['smooth_vs_noisy'] = (np.clip(base, 0, 1), np.clip(base + rng.normal(0, 0.05, (H, W)), 0, 1))
    cases['noise_vs_noise'] = (rng.random((H, W)), rng.random((H, W)))
    cases['shifted'] = (np.clip(base, 0, 1), np.clip(np.roll(base, 2, axis=1) * 0.9 + 0.03, 0, 1))
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    win = np.outer(g, g); win /= win.sum()
    out, meta = {}, []
    for name, (x, y) in cases.items():
        x = x.astype(np.float32); y = y.astype(np.float32)             # the tracker hands float32 images to the metric
        X, Y = x.astype(np.float64), y.astype(np.float64)
        C1, C2 = (0.01 * 1.0) ** 2, (0.03 * 1.0) ** 2
        acc = 0.0
        for i in range(H - 10):
            for j in range(W - 10):
                px, py = X[i:i + 11, j:j + 11], Y[i:i + 11, j:j + 11]
                mx, my = (win * px).sum(), (win * py).sum()
                sx = (win * px * px).sum() - mx * mx; sy = (win * py * py).sum() - my * my
                sxy = (win * px * py).sum() - mx * my
                acc += ((2 * mx * my + C1) * (2 * sxy + C2)) / ((mx * mx + my * my + C1) * (sx + sy + C2))
        out[name + '.x'] = x; out[name + '.y'] = y
        meta.append({'name': name, 'ssim': acc / ((H - 10) * (W - 10)), 'mse': float(((X - Y) ** 2).mean())})
    save_npz('metrics_published.npz', meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **out)
    print(meta)


def make_etnet():
    """ET-Net (reference class EITR) with deterministic synthetic weights: 3 frames of one 64x96 sequence (96 tokens per
    scale) -> images, final ConvLSTM states; once per `norm` the reference's ConvLayers accept (None, 'BN', 'IN';
    u_trans.py:16-52 passes it to the head, the encoders, the decoders and the prediction layer)."""
    for norm, tag, seed in ((None, 'etnet', 17), ('BN', 'etnet_bn', 18), ('IN', 'etnet_in', 19)):
        sd = weights.synth_state_dict(weights.etnet_schema(norm=norm), seed=seed)
        net = ref_model.EITR({'num_bins': 5, 'norm': norm})
        assert list(net.state_dict().keys()) == list(sd.keys()), norm
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        net.eval()
        F, H, W = 3, 64, 96
        vox = synth.sparse_voxels(171, F, 5, H, W, density=0.15)
        imgs = []
        with torch.no_grad():
            for f in range(F):
                imgs.append(net(torch.from_numpy(vox[f:f + 1]))['image'].numpy())
        st = net._states
        out = {f'h{i}_sub': st[i][0].numpy()[:, ::4] for i in range(3)}
        out.update({f'c{i}_sub': st[i][1].numpy()[:, ::4] for i in range(3)})
        save_npz(f'{tag}_seq.npz', voxel_sha=np.array(sha(vox)), voxel_args=np.array([171, F, 5, H, W]), seed=np.array(seed),
                 weights_sha=np.array(weights.state_dict_digest(sd)), images=np.concatenate(imgs), **out)


def make_spade():
    """SPADE-E2VID (reference class Unet6, exported as SpadeE2vid) with deterministic synthetic weights: 4 frames of
    one 64x96 sequence -> images, final hidden states, and the 3-channel prev_recs of the last frame."""
    sd = weights.synth_state_dict(weights.spade_e2vid_schema(), seed=13)
    net = ref_model.SpadeE2vid()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.eval()
    F, H, W = 4, 64, 96
    vox = synth.sparse_voxels(131, F, 5, H, W, density=0.15)
    imgs = []
    with torch.no_grad():
        for f in range(F):
            imgs.append(net(torch.from_numpy(vox[f:f + 1].copy()))['image'].numpy())
    out = {f'h{i}_sub': net.states[i][0].numpy()[:, ::4] for i in range(4)}
    out.update({f'c{i}_sub': net.states[i][1].numpy()[:, ::4] for i in range(4)})
    save_npz('spade_seq.npz', voxel_sha=np.array(sha(vox)), voxel_args=np.array([131, F, 5, H, W]), seed=np.array(13),
             weights_sha=np.array(weights.state_dict_digest(sd)), images=np.concatenate(imgs), prev_recs=net.prev_recs.numpy(), **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['voxel', 'dataset', 'helpers', 'firenet', 'e2vid', 'eval', 'eval_e2vid', 'color', 'spade', 'metrics', 'etnet']
    for w in which:
        {'voxel': make_voxel, 'dataset': make_dataset, 'helpers': make_helpers,
         'firenet': make_firenet, 'e2vid': make_e2vid, 'eval': make_eval, 'eval_e2vid': make_eval_e2vid, 'color': make_color, 'spade': make_spade, 'metrics': make_metrics_published, 'etnet': make_etnet}[w]()

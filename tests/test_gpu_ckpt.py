"""Trained-checkpoint hook: `EVREAL_MODEL_CKPT=<file> EVREAL_MODEL_METHOD=<name> python bench.py --config ckpt` loads a
checkpoint through the drop-in loader (evreal_amd.eval.get_model_from_checkpoint_path = eval.py:124-158 of the reference), runs the bench
step on it and replays frames through a CPU oracle built from the same state_dict.  The only trained checkpoints available offline are
the shipped FireNet / FireNet+ models (tests/golden/firenet*_weights.npz hold their arrays): they go through the hook here, in the
reference's own checkpoint layouts; a user's E2VID / E2VID+ / HyperE2VID file is picked up from the environment (skipped when absent)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import load_npz

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_ckpt(path, method, extra_env=None):
    env = dict(os.environ, EVREAL_MODEL_CKPT=str(path), EVREAL_MODEL_METHOD=method, **(extra_env or {}))
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--config', 'ckpt', '--sub', '--steps', '4', '--warmup', '1', '--n-seq', '4',
           '--parity-frames', '3']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])


def _check(d, gate):
    sp = d['score_parity']
    assert d['config']['weights'].startswith('file:') and d['config']['name'] == 'ckpt'
    assert sp['frames'] == 3 and sp['image_gate'] == gate and sp['image_gate_ok'], sp
    assert sp['all_3sf'], sp
    assert d['config']['range_guard']['runs_beyond_exact_range'] == 0
    assert d['value'] > 0 and d['roofline']['frac'] > 0


def test_shipped_firenet_checkpoint_through_the_hook(tmp_path):
    w = load_npz('firenet_weights.npz')
    ckpt = {'state_dict': {k: torch.from_numpy(w[k]) for k in w.files},
            'config': {'model': {'num_bins': 5, 'skip_type': 'no_skip', 'recurrent_block_type': 'convgru', 'base_num_channels': 16,
                                 'num_residual_blocks': 2, 'recurrent_blocks': {'resblock': [0]}, 'kernel_size': 3,
                                 'final_activation': 'none', 'norm': 'none', 'BN_momentum': 0.01}}}
    torch.save(ckpt, tmp_path / 'firenet.pth')
    d = _bench_ckpt(tmp_path / 'firenet.pth', 'FireNet')
    _check(d, 1e-5)
    assert d['dtype'] == 'f16x3' and d['config']['arithmetic_mode'] == 'h3'


def test_shipped_firenet_plus_checkpoint_through_the_hook(tmp_path):
    """FireNet+ checkpoints carry config['arch'] = {'type': 'FireNet', 'args': ...} (parse_config of the reference; a plain dict unpickles alike)."""
    w = load_npz('firenetplus_weights.npz')
    ckpt = {'state_dict': {k: torch.from_numpy(w[k]) for k in w.files},
            'config': {'arch': {'type': 'FireNet', 'args': {'num_bins': 5, 'base_num_channels': 16, 'kernel_size': 3}}}}
    torch.save(ckpt, tmp_path / 'firenetplus.pth')
    d = _bench_ckpt(tmp_path / 'firenetplus.pth', 'FireNet+')
    _check(d, 1e-5)
    # the fast arithmetic on trained weights: FireNet's ConvGRU epilogue writes 4-channel runs, which P6 cannot take, so EVR_ARITH=mx6 narrows
    # to f16 + MX-fp8 on the zero-padded 32-channel kernels (EVR_FIRENET_PAD32=1) -- the line says which arithmetic ran
    d = _bench_ckpt(tmp_path / 'firenetplus.pth', 'FireNet+', {'EVR_ARITH': 'mx6', 'EVR_FIRENET_PAD32': '1'})
    _check(d, 1e-4)
    assert d['config']['arithmetic_mode'] == 'mx' and d['dtype'] == 'f16+mxfp8'


def test_user_supplied_checkpoint():
    path = os.environ.get('EVREAL_MODEL_CKPT')
    if not path or not os.path.exists(path):
        pytest.skip("EVREAL_MODEL_CKPT names no file (trained E2VID-family checkpoints cannot be downloaded here)")
    d = _bench_ckpt(path, os.environ.get('EVREAL_MODEL_METHOD', 'E2VID'))
    _check(d, 1e-5)

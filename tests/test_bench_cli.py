"""bench.py's launch logic and side-block arithmetic (no GPU): `--gpus N` must either launch N ranks itself or agree
with the launcher's WORLD_SIZE; score_parity's 3-significant-figure rule."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_resolve_world():
    assert bench.resolve_world(1, {}) == ('inprocess', 1)
    assert bench.resolve_world(8, {}) == ('relaunch', 8)                  # python bench.py --gpus 8: self-launch
    assert bench.resolve_world(8, {'WORLD_SIZE': '8'}) == ('ranked', 8)   # the driver's torch.distributed.run launch
    assert bench.resolve_world(1, {'WORLD_SIZE': '1'}) == ('inprocess', 1)
    with pytest.raises(SystemExit):
        bench.resolve_world(8, {'WORLD_SIZE': '2'})
    with pytest.raises(SystemExit):
        bench.resolve_world(1, {'WORLD_SIZE': '4'})


def test_launch_command_is_one_rank_per_gpu_on_loopback():
    cmd = bench.launch_command(4, ['--gpus', '4', '--steps', '7', '--warmup', '2'], port=29999)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29999'
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '7', '--warmup', '2']   # the ranks see the same flags
    free = bench.launch_command(2, [])
    assert 1024 < int(free[free.index('--master-port') + 1]) < 65536


def test_score_parity_three_significant_figures():
    rng = np.random.default_rng(0)
    img = rng.random((4, 8, 8)).astype(np.float32)
    cpu = [(img[i], [0.075291, 0.023561, 0.25398]) for i in range(4)]
    gpu = [(img[i] + 3e-6, [0.075292, 0.023561, 0.25399]) for i in range(4)]
    p = bench.score_parity(gpu, cpu, ['mse', 'ssim', 'lpips'])
    assert p['frames'] == 4 and p['all_3sf'] and p['image_gate_ok'] and p['image_max_abs_err'] < 1e-5
    assert p['mse']['3sf'] and p['mse']['rel_err'] < 2e-5
    bad = [(img[i] + 1e-3, [0.0761, 0.0236, 0.254]) for i in range(4)]
    q = bench.score_parity(bad, cpu, ['mse', 'ssim', 'lpips'])
    assert not q['mse']['3sf'] and not q['all_3sf'] and not q['image_gate_ok']
    assert bench.sig3(0.0752905513) == 0.0753 and bench.sig3(1234.5) == 1230.0


def test_traffic_is_null_when_profiles_are_stale(monkeypatch):
    import json
    d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    v, note = bench.measured_traffic(d.get('arith', 'mx'))
    if d.get('source_sha') == bench.source_sha():
        assert v == d['convlstm_bytes_per_launch']
    else:
        assert v is None and 're-run' in note
    other = 'mx' if d.get('arith', 'mx') != 'mx' else 'mx6'
    v, note = bench.measured_traffic(other)          # passes taken in another arithmetic never count
    assert v is None and 'arithmetic' in note


def test_evaluation_gate_default_follows_the_batch_size():
    """pipeline.HotPath's gate rule (host logic): after the first residual block, from 48 sequences per step on after the second;
    EVR_EVAL_GATE overrides, 'none' disables."""
    from evreal_amd.pipeline import default_eval_gate
    assert default_eval_gate(1, {}) == 'res0.conv2'
    assert default_eval_gate(47, {}) == 'res0.conv2'
    assert default_eval_gate(48, {}) == 'res1.conv2'
    assert default_eval_gate(64, {}) == 'res1.conv2'
    assert default_eval_gate(64, {'EVR_EVAL_GATE': 'dec0'}) == 'dec0'
    assert default_eval_gate(8, {'EVR_EVAL_GATE': 'none'}) == 'none'
    assert default_eval_gate(8, {'EVR_EVAL_GATE': ''}) == 'res0.conv2'

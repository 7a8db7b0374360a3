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


def _canned_full_object(n_gpus=1):
    """A full bench object with every block main() builds, each padded to (more than) the size round 3's line carried."""
    long = "x" * 900
    rl = {"bound": "mfma", "achieved": 441.6, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.1766, "traffic": 1893191939, "traffic_source": long,
          "kernel": "conv3x3_wide_kernel<LSTM=true, WN> " + long, "share_of_bracketed_time": None, "arithmetic": long, "mfma_issue_tflops": 1324.8,
          "mfma_issue_frac": 0.5299, "streams": long, "single_stream": {"avg_launch_us": 1960.1, "achieved": 447.4}, "gflop_per_launch": 876.979,
          "avg_launch_us": 1985.9, "launches": 300, "layers": {f"layer{i}": {"us": 1.0, "tflops": 2.0} for i in range(60)}}
    sp = {"frames": 24, "sequences": [0, 37], "image_max_abs_err": 8.9e-7, "image_max_abs_err_per_frame_max5": [1e-7] * 5, "image_gate": 1e-5,
          "image_gate_ok": True, "oracle": long, "all_3sf": True}
    for nm in ('mse', 'ssim', 'lpips'):
        sp[nm] = {"gpu": 0.0752905513, "oracle": 0.0752905561, "rel_err": 6.4e-8, "worst_frame_rel_err": 2e-7, "3sf": True}
    brief = {"value": 7307.12, "ms_per_step": 8.759, "dtype": "f16+mxfp6", "mevents_per_s": 109.6, "model_tflops": 460.1,
             "steady_state": {"seconds": 2.2, "steps": 260, "value": 7418.0, "ms_per_step": 8.6}, "roofline": dict(rl), "score_parity": dict(sp),
             "scores": {"mse": 0.07, "ssim": 0.02, "lpips": 0.25, "count": 1280}, "workload": long, "sequences_per_gpu": 64, "gflop_per_frame": 62.983,
             "cpu_frames_per_s": 10.7}
    out = {"metric": "reconstructed frames/sec + Mevents/sec voxelized, E2VID 346x260 B=5", "value": 5540.12, "unit": "frames/s", "n_gpus": n_gpus,
           "steps": 100, "warmup": 3, "ms_per_step": 11.55, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16x3",
           "data": "synthetic", "mevents_per_s": 83.1, "model_tflops": 348.9, "rccl_ranks": n_gpus if n_gpus > 1 else 0,
           "config": {"workload": long, "name": "e2vid", "sequences_per_gpu": 64, "events_per_window": 15000, "sensor": [346, 260], "bins": 5,
                      "gflop_per_frame": 62.983, "lpips_gflop_per_frame": 4.671, "sharding": "sequences across GPUs", "lpips_weights": "synthetic(seed=0)",
                      "arithmetic_mode": "h3", "weights": "synthetic(seeded)", "unique_steps": 40, "range_guard": {"runs_beyond_exact_range": 0, "layer": ""},
                      "scores": {"mse": 0.07529056528315684, "ssim": 0.02356263208735744, "lpips": 0.25398426154585535, "count": 6400}},
           "roofline": rl, "steady_state": {"seconds": 2.3, "steps": 200, "value": 5580.0, "ms_per_step": 11.4},
           "per_rank": None if n_gpus == 1 else {"frames_per_s_min": 5500.0, "frames_per_s_max": 5560.0, "ranks": n_gpus},
           "roofline_voxelizer": {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "bytes_per_window": 1994200, "kernels": long,
                                  "in_step": {"windows": 426.7, "frac": 0.3878, "note": long}, "standalone_512": {"frac": 0.47}, "achieved": 3765.0, "frac": 0.4706},
           "small_batch": {f"n_seq_{n}": {"value": 1201.0 * n ** 0.5, "ms_per_step": 0.83} for n in (1, 4, 8, 16, 32)} | {"note": long},
           "cpu_baseline": {"value": 10.726, "unit": "frames/s", "cores": 32, "kind": "port", "sample": long, "ms_per_frame": {"forward": 65.0}},
           "score_parity": sp, "fast": brief, "fp8_cross_terms": brief, "fp32_exact": brief, "large_batch": {"n_seq_128": brief},
           "sensor_640x480": dict(brief, roofline_voxelizer={"achieved": 4915.0, "frac": 0.61}),
           "configs": {"1 (E2VID, CPU PyTorch path)": long, "2 (E2VID 346x260, MSE+SSIM+LPIPS)": "the headline line",
                       "3 (FireNet 240x180, k_events)": brief, "4 (HyperE2VID 346x260, 4 sequences)": brief,
                       "5 (ColorNet E2VID+ 970x624, 50k events/window)": brief, "extra (E2VID+ / SSL-E2VID layout 346x260, 64 sequences)": brief,
                       "extra (ET-Net 346x260, 8 sequences)": {"error": "TimeoutExpired: " + long}, "extra (SPADE-E2VID 346x260, 8 sequences)": brief},
           "user_checkpoint": brief,
           "eval_cli": {"save_images_off": {"value": 3100.0, "frame_loop": {"value": 4741.0, "note": long}},
                        "save_images_on": {"value": 2500.0, "frame_loop": {"value": 3217.0, "note": long}}, "what": long}}
    return out


@pytest.mark.parametrize('n_gpus', [1, 8])
def test_driver_line_is_small_and_complete(n_gpus):
    """Round 3's line was 24.7 KB and the driver (which keeps ~8 KB of stdout) could not parse it: the LAST stdout line must stay below
    4 KB whatever the side blocks hold, and carry the contract's keys with `roofline` and `cpu_baseline` objects."""
    import json
    full = _canned_full_object(n_gpus)
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full, 'gpurun_out/bench_full.json')
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 4096, len(text)
    assert json.loads(text) == line
    for k in bench.REQUIRED_KEYS:
        assert k in line, k
    assert line['value'] == 5540.12 and line['dtype'] == 'f16x3' and line['n_gpus'] == n_gpus and line['vs_baseline'] is None
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(line['roofline']) and line['roofline']['frac'] == 0.1766
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(line['cpu_baseline']) and line['cpu_baseline']['kind'] == 'port'
    assert 'workload' in line['config'] and 'model' not in line['config']
    assert line['score_parity']['image_max_abs_err'] == 8.9e-7 and line['score_parity']['sequences'] == [0, 37]
    assert line['fast']['value'] == 7307.1 and line['fast']['dtype'] == 'f16+mxfp6'
    assert line['configs']['firenet_3']['value'] == 7307.1 and 'error' in line['configs']['etnet']
    assert line['full'] == 'gpurun_out/bench_full.json'
    assert (line.get('per_rank') is None) == (n_gpus == 1)
    # every optional block ten times larger: the required keys survive, the line still fits
    huge = _canned_full_object(n_gpus)
    huge['small_batch'] = {f"n_seq_{n}": {"value": float(n)} for n in range(1, 400)}
    line2 = bench.compact_line(huge, None)
    assert len(json.dumps(line2)) < bench.LINE_LIMIT and all(k in line2 for k in bench.REQUIRED_KEYS)


def _agg_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    K, n_seq = 5, 3
    sc = np.full((K, n_seq, 3), 0.1 * (rank + 1)) * np.array([1.0, 2.0, 3.0])
    tot, elapsed, per_rank = bench.aggregate(sc, 0.5 * (rank + 1), n_seq, K, dist, 'cpu')
    q.put((rank, tot.tolist(), elapsed, per_rank))
    dist.barrier()
    dist.destroy_process_group()


def test_aggregate_under_world2_gloo():
    """bench.py's end-of-run exchange (N > 1): totals = MetricTracker's sums over all ranks' sequences, time = the slowest rank's,
    every rank's own rate reported; without a process group it is the identity."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_agg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60); assert p.exitcode == 0
    for _, tot, elapsed, per_rank in res:
        t = np.array(tot)
        assert t.shape == (1, 4) and t[0, 3] == 2 * 3 * 5
        np.testing.assert_allclose(t[0, :3] / t[0, 3], np.array([0.15, 0.30, 0.45]), rtol=1e-12)     # mean of 0.1 and 0.2, x (1, 2, 3)
        assert elapsed == 1.0                                                                        # the slower rank
        assert per_rank == {"frames_per_s_min": 15.0, "frames_per_s_max": 30.0, "ranks": 2}
    tot, elapsed, per_rank = bench.aggregate(np.ones((4, 2, 3)), 0.25, 2, 4)
    assert per_rank is None and elapsed == 0.25 and tot.tolist() == [[8.0, 8.0, 8.0, 8.0]]

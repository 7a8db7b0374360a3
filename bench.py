#!/usr/bin/env python3
"""Headline benchmark: reconstructed frames/s (+ Mevents/s voxelized) of the E2VID hot path at 346x260,
5 bins, 15k events/window on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n-seq S] [--sensor WxH]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself (re-exec under
torch.distributed.run, one process per GPU, RCCL); under a launcher the flag must equal WORLD_SIZE.

A step = one frame for each of S independent synthetic sequences per GPU (sequences shard across GPUs,
SURVEY 8e): raw events (resident in HBM) -> voxel grid -> event-tensor normalization -> pad -> E2VID
forward (split f16 + MX-fp8 MFMA, fp32 accumulate; EVR_FP32=1: exact fp32 MFMA) -> crop -> robust percentile
normalization -> clip -> MSE + SSIM + LPIPS against the reference frame (LPIPS = AlexNet v0.1 structure on
synthetic weights unless EVREAL_LPIPS_WEIGHTS names a real state_dict: they cannot be downloaded here).
The evaluation half of a frame (robust norm, MSE/SSIM, LPIPS) runs on a second HIP stream and overlaps the
reconstruction of the next frame (`--no-overlap`: one stream); all of a step's work is inside the timed region.

Rank 0 prints ONE JSON line (contract in the task statement).  Besides `roofline` (dominant kernel: the ConvLSTM gate
convolutions) and `cpu_baseline` it carries, all measured OUTSIDE the timed region of the same run:
  roofline_voxelizer  HIP-event time of the tensorizer launches (in the step and standalone at S and 512 windows)
  score_parity        the first frames of sequence 0 replayed on the GPU and through the CPU oracle: per-frame image
                      error, mean MSE/SSIM/LPIPS of both, relative error, agreement to 3 significant figures
  fp32_exact          the same steps in a sub-process with EVR_FP32=1 (exact fp32 MFMA arithmetic)
  sensor_640x480      the same workload on 640x480 streams (north_star's second sensor size), sub-process
  small_batch         1 and 4 sequences per GPU (the reference's regime is batch 1)
  steady_state        >= 2 s of back-to-back steps (the timed region of a 20-step run is 0.25 s)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BINS, K_EVENTS = 5, 15000
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: v_mfma_f32_32x32x16_bf16 dense peak (micro-benchmark ceiling 2382)
PEAK_HBM_GBS = 8000.0             # same guide: HBM3E spec (6.3 TB/s measured with a float4 copy)
TRAFFIC_SOURCES = ['evreal_amd/csrc/conv.hip', 'evreal_amd/csrc/conv.h', 'evreal_amd/csrc/model.cpp',
                   'evreal_amd/csrc/packed.h']


# ------------------------------------------------------------------------------------------------ launch
def launch_command(gpus, argv, port=None):
    """argv of the self-launch: one rank per GPU under torch.distributed.run on 127.0.0.1."""
    if port is None:
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={gpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def resolve_world(args_gpus, environ):
    """-> ('inprocess' | 'relaunch' | 'ranked', world).  Fails loudly when --gpus contradicts the launcher."""
    ws = environ.get('WORLD_SIZE')
    if ws is None:
        return ('inprocess', 1) if args_gpus <= 1 else ('relaunch', args_gpus)
    if int(ws) != args_gpus:
        raise SystemExit(f"bench.py: --gpus {args_gpus} but the launcher set WORLD_SIZE={ws}; pass --gpus {ws}")
    return ('ranked' if int(ws) > 1 else 'inprocess', int(ws))


# ------------------------------------------------------------------------------------------------ inputs
def build_inputs(rank, n_seq, n_steps, device, W_, H_):
    """Step-major resident event arrays: window (step, seq) = 15k consecutive events of sequence `seq`."""
    from evreal_amd import synth
    k = K_EVENTS
    xy = np.empty((n_steps, n_seq, k, 2), np.int16)
    ts = np.empty((n_steps, n_seq, k), np.float64)
    pol = np.empty((n_steps, n_seq, k), np.uint8)
    refs = np.empty((n_seq, H_, W_), np.float32)
    for s in range(n_seq):
        seed = rank * n_seq + s
        t, x, y, p = synth.poisson_events(seed, n_steps * k, 1.0e6, W_, H_)
        xy[:, s, :, 0] = x.reshape(n_steps, k); xy[:, s, :, 1] = y.reshape(n_steps, k)
        ts[:, s] = t.reshape(n_steps, k); pol[:, s] = p.reshape(n_steps, k)
        refs[s] = synth.smooth_frames(seed, 1, W_, H_)[0, :, :, 0].astype(np.float32) / 255
    offs = (np.arange(n_steps)[:, None] * (n_seq * k) + np.arange(n_seq + 1)[None, :] * k).astype(np.int64)
    d = lambda a: torch.from_numpy(a).to(device)
    return d(xy.reshape(-1, 2)), d(ts.reshape(-1)), d(pol.reshape(-1)), d(offs), d(refs), (xy, ts, pol, refs)


# ------------------------------------------------------------------------------------------------ CPU leg
def cpu_baseline(host_inputs, sd, kw, n_frames, W_, H_, budget_s=25.0, lpips_sd=None, keep_frames=0):
    """Oracle ("port") timed on the host cores, batch 1 like the reference: C voxelizer (1 thread) + numpy
    normalization + torch-CPU E2VID forward + numpy percentile + MSE + scipy SSIM + torch-CPU LPIPS.  BOUNDED: the torch
    thread count is the faster of {8, 32} (capped by the core count; all 256 threads of the GPU box's host run this
    batch-1 network ~100x slower), then sequence 0 runs from a state reset, window 0 onwards, until `n_frames` or
    ~`budget_s` seconds are used.  The first `keep_frames` frames' images and scores are returned for score_parity."""
    import ctypes
    from oracle import model as omod, prepost as op, metrics as omet, lpips as olp
    from evreal_amd import synth
    xy, ts, pol, refs = host_inputs
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'liboracle.so'))
    okw = {k: kw[k] for k in ['num_bins', 'base_num_channels', 'num_encoders', 'num_residual_blocks', 'kernel_size',
                              'norm', 'use_upsample_conv', 'recurrent_block_type', 'final_activation']}
    o = omod.UNetRecurrentOracle({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, **okw)
    crop = op.CropParams(W_, H_, kw['num_encoders'])
    offs = np.array([0, K_EVENTS], dtype=np.int64)
    out = np.empty((1, BINS, H_, W_), np.float32)
    f = lambda a: a.ctypes.data_as(ctypes.c_void_p)

    def frame(w, keep=None):
        t0 = time.perf_counter()
        xs, ys, tf, ps = synth.window_events_f32(ts[w, 0], xy[w, 0], pol[w, 0], 0, K_EVENTS)
        lib.oracle_voxelize(f(xs), f(ys), f(tf), f(ps), f(offs), 1, BINS, H_, W_, f(out))
        t1 = time.perf_counter()
        v = op.normalize_event_tensor(out)
        t2 = time.perf_counter()
        with torch.no_grad():
            img = crop.crop(o(torch.from_numpy(crop.pad(v))).numpy())[0, 0]
        t3 = time.perf_counter()
        img = op.post_process_normalization(img, 'robust')
        t4 = time.perf_counter()
        a, b = omet.clip01(img), omet.clip01(refs[0])
        sc = [omet.mse(a, b), omet.ssim(a, b)]
        if lpips_sd is not None:
            with torch.no_grad():
                sc.append(float(olp.lpips(lpips_sd, a[None], b[None])[0]))
        t5 = time.perf_counter()
        if keep is not None:
            keep.append((img.copy(), sc))
        return (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)

    t_start = time.perf_counter()
    best, best_t = None, None
    for nt in sorted({min(8, os.cpu_count()), min(32, os.cpu_count())}):
        torch.set_num_threads(nt)
        frame(0)                                   # warm-up at this thread count
        dt = sum(frame(1 % xy.shape[0]))
        if best_t is None or dt < best_t:
            best, best_t = nt, dt
    torch.set_num_threads(best)
    o.reset_states()                               # the timed/kept frames: sequence 0 from its first window
    times = {'voxel': 0.0, 'norm': 0.0, 'forward': 0.0, 'post': 0.0, 'metrics': 0.0}
    kept, done = [], 0
    while done < max(n_frames, keep_frames) and (done < keep_frames or (time.perf_counter() - t_start) < budget_s):
        for k, dt in zip(times, frame(done % xy.shape[0], kept if done < keep_frames else None)):
            times[k] += dt
        done += 1
    total = sum(times.values())
    res = {"value": round(done / total, 3), "unit": "frames/s", "cores": best, "kind": "port",
           "sample": f"{done} frames of one {W_}x{H_} sequence, batch 1, E2VID forward on {best} torch threads of the "
                     f"{os.cpu_count()}-core host (C voxelizer and numpy/scipy stages 1 thread), torch {torch.__version__}",
           "ms_per_frame": {k: round(1e3 * v / max(done, 1), 3) for k, v in times.items()},
           "mevents_per_s_voxelizer": round(K_EVENTS * done / max(times['voxel'], 1e-9) / 1e6, 2)}
    return res, kept


def sig3(x):
    return float(f"{x:.3g}")


def score_parity(gpu_frames, cpu_frames, names):
    """gpu_frames / cpu_frames: [(image [H,W], [scores...])] for the same windows of sequence 0."""
    n = min(len(gpu_frames), len(cpu_frames))
    if n == 0:
        return None
    img_err = [float(np.abs(gpu_frames[i][0] - cpu_frames[i][0]).max()) for i in range(n)]
    out = {"frames": n, "sequence": 0, "image_max_abs_err": max(img_err), "image_max_abs_err_per_frame_max5": sorted(img_err)[-5:],
           "image_gate_1e-4": bool(max(img_err) < 1e-4),
           "oracle": "oracle/ (torch-CPU fp32 restatement pinned against the reference classes; MSE/SSIM/LPIPS arithmetic "
                     "restated from scikit-image / pyiqa: parity unpinned, see DESIGN.md section 3)"}
    ok = True
    for j, nm in enumerate(names):
        g = float(np.mean([gpu_frames[i][1][j] for i in range(n)]))
        c = float(np.mean([cpu_frames[i][1][j] for i in range(n)]))
        worst = max(abs(gpu_frames[i][1][j] - cpu_frames[i][1][j]) / max(abs(cpu_frames[i][1][j]), 1e-30) for i in range(n))
        same = sig3(g) == sig3(c) or abs(g - c) <= 5e-4 * abs(c)
        ok = ok and same
        out[nm] = {"gpu": g, "oracle": c, "rel_err": abs(g - c) / max(abs(c), 1e-30), "worst_frame_rel_err": worst, "3sf": bool(same)}
    out["all_3sf"] = bool(ok)
    return out


# ------------------------------------------------------------------------------------------------ helpers
def sub_run(extra_args, env_extra, steps, warmup, timeout=420):
    """The same bench in a sub-process (outside this process's timed region); returns its parsed JSON line."""
    env = dict(os.environ); env.update(env_extra)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(steps), '--warmup', str(warmup),
           '--sub'] + extra_args
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        line = [l for l in p.stdout.splitlines() if l.startswith('{')][-1]
        return json.loads(line)
    except Exception as e:      # a side block must never take the headline down
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def source_sha():
    h = hashlib.sha256()
    for p in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, p), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic():
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes -- only if they were
    taken on THIS source (profiles/pmc_traffic.json stores the sha of the kernel sources); otherwise null."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not os.path.exists(path):
        return None, "no PMC pass committed"
    d = json.load(open(path))
    if d.get('source_sha') != source_sha():
        return None, (f"PMC passes in profiles/ were taken on kernel sources {d.get('source_sha')}, this build is "
                      f"{source_sha()}: re-run tools/profile_round.sh (last value: {d.get('convlstm_bytes_per_launch')})")
    return d.get('convlstm_bytes_per_launch'), f"rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, {d.get('profile', 'profiles/')}"


def time_launches(fn, reps, device):
    """Average ms of fn() by HIP events on the current stream (the stream fn launches on)."""
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--n-seq', type=int, default=64, help='independent sequences advanced together per GPU')
    ap.add_argument('--sensor', default='346x260', help='sensor WxH of the synthetic streams (346x260 | 640x480)')
    ap.add_argument('--cpu-frames', type=int, default=200, help='frames of the CPU baseline (0 disables)')
    ap.add_argument('--parity-frames', type=int, default=12, help='frames of sequence 0 replayed through the oracle')
    ap.add_argument('--profile-filter', default='rec', help='layers bracketed with HIP events (roofline block)')
    ap.add_argument('--no-overlap', action='store_true', help='evaluation kernels on the reconstruction stream (no second HIP stream)')
    ap.add_argument('--sub', action='store_true', help='side run: headline + roofline only (no CPU leg, no sub-runs)')
    args = ap.parse_args()
    W_, H_ = [int(v) for v in args.sensor.lower().split('x')]

    mode, world = resolve_world(args.gpus, os.environ)
    if mode == 'relaunch':
        if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPUs are visible")
        cmd = launch_command(args.gpus, sys.argv[1:])
        sys.stdout.flush()
        os.execvpe(cmd[0], cmd, dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or os.environ.get('EVR_FORCE_DIST'):      # EVR_FORCE_DIST=1: exercise the RCCL path on one rank
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    side = (world == 1) and not args.sub                   # the side blocks run on the single-GPU default run only

    from evreal_amd import model, weights
    from evreal_amd.pipeline import HotPath
    from evreal_amd.dist import reduce_metric_sums
    from evreal_amd.lpips import LPIPS
    from evreal_amd.voxel import Voxelizer

    kw = dict(weights.E2VID_KWARGS)
    sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=0)
    net = model.E2VIDRecurrent(kw)
    net.load_state_dict(sd)
    n_seq, K, Wm = args.n_seq, args.steps, args.warmup
    xy, ts, pol, offs, refs, host_inputs = build_inputs(rank, n_seq, K + Wm, device, W_, H_)
    lpips_path = os.environ.get('EVREAL_LPIPS_WEIGHTS')
    if lpips_path:      # a real pyiqa/lpips AlexNet-v0.1 state_dict supplied by the user
        lpips_sd = {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in
                    torch.load(lpips_path, map_location='cpu', weights_only=False).items()}
        lpips_tag = "file:" + hashlib.sha256(open(lpips_path, 'rb').read()).hexdigest()[:16]
    else:
        lpips_sd = weights.synth_lpips_state_dict(seed=0)
        lpips_tag = "synthetic(seed=0)"
    lp = LPIPS(lpips_sd)
    hp = HotPath(net, BINS, (H_, W_), n_seq, event_tensor_normalization=True, post_process_norm='robust',
                 metrics=('mse', 'ssim', 'lpips'), device=str(device), lpips=lp, overlap=not args.no_overlap)
    scores = torch.zeros((K + Wm, n_seq, 3), dtype=torch.float64, device=device)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(Wm):
        hp.step_raw(xy, ts, pol, offs[s], refs, scores[s])
    net.profile(args.profile_filter)
    hp.time_voxelizer(True)
    barrier()
    t0 = time.perf_counter()
    for s in range(Wm, Wm + K):
        hp.step_raw(xy, ts, pol, offs[s], refs, scores[s])
    barrier()
    elapsed = time.perf_counter() - t0
    prof = net.profile_read()
    net.profile(None)
    vox_ms_in_step = hp.time_voxelizer(False)
    sc = scores[Wm:].cpu().numpy()                       # [K, n_seq, 3] before anything below rewrites the buffer

    # ---- everything below is outside the timed region ----
    scratch = torch.zeros((n_seq, 3), dtype=torch.float64, device=device)
    # the same layers once more with the evaluation kernels on the SAME stream, so the dominant kernel's duration is
    # also known without the second stream's kernels sharing the chip with it
    prof_single = None
    if hp.overlap:
        hp.overlap = False
        net.profile(args.profile_filter)
        for s in range(Wm, min(Wm + 3, Wm + K)):
            hp.step_raw(xy, ts, pol, offs[s], refs, scratch)
        barrier()
        prof_single = net.profile_read()
        net.profile(None)
        hp.overlap = True

    steady = None
    if side or args.sub:
        cycles = max(1, int(np.ceil(2.2 / max(elapsed, 1e-3))))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(cycles):
            for s in range(Wm, Wm + K):
                hp.step_raw(xy, ts, pol, offs[s], refs, scratch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        steady = {"seconds": round(dt, 3), "steps": cycles * K, "value": round(n_seq * cycles * K / dt, 2),
                  "ms_per_step": round(1e3 * dt / (cycles * K), 4)}

    # metric aggregation exactly as MetricTracker.update (eval.py:259-266): sum(mean*count), count
    seq_mean = sc.mean(axis=0)                           # per sequence
    sums = np.array([[seq_mean[:, 0].sum() * K, seq_mean[:, 1].sum() * K, seq_mean[:, 2].sum() * K, n_seq * K]],
                    dtype=np.float64)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tot = reduce_metric_sums(torch.from_numpy(sums).to(device), dist)
    elapsed = float(tmax.item())

    if rank == 0:
        frames = n_seq * K * world
        fps = frames / elapsed
        flops_step = net.flops_per_step()
        lstm = [p for p in prof if '.rec' in p['name']]
        rl_flops = sum(p['flops_per_launch'] * p['launches'] for p in lstm)
        rl_ms = sum(p['ms'] for p in lstm)
        rl_launches = sum(p['launches'] for p in lstm)
        achieved = rl_flops / (rl_ms * 1e-3) / 1e12 if rl_ms > 0 else 0.0
        # arithmetic mode of the 32-channel-chunk convolutions (model.cpp finish_conv): default split f16 + MX-fp8 (csrc/conv.h)
        # (f16 main product + one MX-scaled fp8 MFMA for the two cross terms per 32 k, fp32 accumulate: ~2^-16 relative per
        # product term); EVR_FP32=1 selects the exact fp32 MFMA.  `achieved` counts ALGORITHMIC (direct convolution)
        # flops in both modes; in split mode the matrix pipe is busy for 2x that in f16-rate cycles (the fp8 MFMA covers
        # its 64 k in the time of 32 f16 k), reported as mfma_issue_*; `peak` stays the f16/bf16 dense peak.
        x3 = not os.environ.get('EVR_FP32')
        peak = PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_F32_MFMA_TFLOPS
        traffic, traffic_note = measured_traffic() if (x3 and n_seq == 64 and (W_, H_) == (346, 260)) else (None, "not the profiled configuration")
        out = {
            "metric": "reconstructed frames/sec + Mevents/sec voxelized, E2VID %dx%d B=5" % (W_, H_),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16+mxfp8" if x3 else "f32", "data": "synthetic",
            "mevents_per_s": round(frames * K_EVENTS / elapsed / 1e6, 2),
            "model_tflops": round(flops_step * K * world / elapsed / 1e12, 2),
            "rccl_ranks": world if dist is not None else 0,
            "config": {"workload": "E2VID (synthetic weights, BN folded) on synthetic %dx%d Poisson events, 5 bins, "
                                   "15k events/window (k_events), %d sequences per GPU advanced together; per frame: "
                                   "voxelize(raw)+event-tensor norm+pad+forward+crop+robust norm+clip+MSE+SSIM+LPIPS "
                                   "(LPIPS: AlexNet-v0.1 structure, weights %s)" % (W_, H_, n_seq, lpips_tag),
                       "sequences_per_gpu": n_seq, "events_per_window": K_EVENTS, "sensor": [W_, H_], "bins": BINS,
                       "gflop_per_frame": round(flops_step / n_seq / 1e9, 3),
                       "lpips_gflop_per_frame": round(lp.flops() / n_seq / 1e9, 3), "sharding": "sequences across GPUs",
                       "lpips_weights": lpips_tag,
                       "scores": {"mse": tot[0, 0] / tot[0, 3], "ssim": tot[0, 1] / tot[0, 3], "lpips": tot[0, 2] / tot[0, 3],
                                  "count": int(tot[0, 3])}},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": ("conv3x3_wide_kernel<LSTM=true, WN> (ConvLSTM gate convolutions: 256 x 128 tiles, two blocks per CU, up to 256 input channels; 256 x 256 tiles above)" if x3 else
                                    "conv_igemm_kernel<32,4,4,LSTM=true,REGSTAGE,X3=0> (ConvLSTM gate convolutions)"),
                         "arithmetic": ("split: x = hi + lo8*2^-12, w = hi + wlo8*2^-(e+12) (f16 hi, fp8 e4m3 residuals); per 32 k "
                                        "acc += hi_w*hi_x on 2 x v_mfma_f32_32x32x16_f16 + (w8*lo8 + wlo8*x8) on 1 x "
                                        "v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulate; activations stored PACKED (f16 hi | "
                                        "fp8 lo8 | fp8 x8 per 16 channels) by the producer" if x3 else
                                        "v_mfma_f32_32x32x2_f32 (exact fp32 fma chain)"),
                         "mfma_issue_tflops": round(achieved * (2 if x3 else 1), 2),
                         "mfma_issue_frac": round(achieved * (2 if x3 else 1) / peak, 4),
                         "streams": ("2: reconstruction | evaluation (robust norm, MSE/SSIM, LPIPS of the previous frame) -- "
                                     "the timed launches share the chip with the evaluation kernels"
                                     if prof_single is not None else "1"),
                         "single_stream": (None if prof_single is None else (lambda l: {
                             "avg_launch_us": round(1e3 * sum(p['ms'] for p in l) / max(sum(p['launches'] for p in l), 1), 2),
                             "achieved": round(sum(p['flops_per_launch'] * p['launches'] for p in l) / (sum(p['ms'] for p in l) * 1e-3) / 1e12, 2)})(
                             [p for p in prof_single if '.rec' in p['name']])),
                         "gflop_per_launch": round(rl_flops / max(rl_launches, 1) / 1e9, 3),
                         "avg_launch_us": round(1e3 * rl_ms / max(rl_launches, 1), 2), "launches": rl_launches,
                         "layers": {p['name']: {"us": round(1e3 * p['ms'] / p['launches'], 2),
                                                "tflops": round(p['flops_per_launch'] * p['launches'] / (p['ms'] * 1e-3) / 1e12, 2)}
                                    for p in prof}},
            "steady_state": steady,
        }

        # ---- tensorizer roofline: algorithmic bytes 13 N + 4 B H W per window (raw form) ----
        vz = Voxelizer(str(device))
        bytes_win = 13 * K_EVENTS + 4 * BINS * H_ * W_
        rv = {"bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s", "bytes_per_window": bytes_win,
              "kernels": "every launch of one evr_voxelize_raw call (statistics for the event-tensor normalization included)"}
        if vox_ms_in_step:
            g = n_seq * bytes_win / (vox_ms_in_step * 1e-3) / 1e9
            rv["in_step"] = {"windows": n_seq, "us": round(1e3 * vox_ms_in_step, 2), "achieved": round(g, 1), "frac": round(g / PEAK_HBM_GBS, 4),
                             "note": "inside the timed region, sharing the chip with the evaluation stream"}
        n_win_avail = (K + Wm) * n_seq
        for nw in sorted({n_seq, min(512, n_win_avail)}):
            o_ = torch.arange(nw + 1, dtype=torch.int64, device=device) * K_EVENTS
            buf = torch.empty((nw, BINS, H_, W_), dtype=torch.float32, device=device)
            st = torch.zeros((nw, 3), dtype=torch.float64, device=device)
            ms = time_launches(lambda: vz.voxelize_raw(xy, ts, pol, o_, BINS, (H_, W_), out=buf, stats=st), 20, device)
            g = nw * bytes_win / (ms * 1e-3) / 1e9
            rv[f"standalone_{nw}"] = {"windows": nw, "us": round(1e3 * ms, 2), "achieved": round(g, 1), "frac": round(g / PEAK_HBM_GBS, 4),
                                      "mevents_per_s": round(nw * K_EVENTS / (ms * 1e-3) / 1e6, 1)}
            del buf, st
        best = max((v for k, v in rv.items() if isinstance(v, dict)), key=lambda v: v["frac"])
        rv["achieved"], rv["frac"] = best["achieved"], best["frac"]
        out["roofline_voxelizer"] = rv

        if side:
            # ---- small batches (the reference's regime is one sequence at a time) ----
            sb = {}
            for ns in (1, 4):
                if ns >= n_seq:
                    continue
                h2 = HotPath(net, BINS, (H_, W_), ns, event_tensor_normalization=True, post_process_norm='robust',
                             metrics=('mse', 'ssim', 'lpips'), device=str(device), lpips=lp, overlap=not args.no_overlap)
                sc2 = torch.zeros((ns, 3), dtype=torch.float64, device=device)
                ofs = [offs[s][:ns + 1].contiguous() for s in range(K + Wm)]
                for s in range(Wm):
                    h2.step_raw(xy, ts, pol, ofs[s], refs[:ns], sc2)
                torch.cuda.synchronize()
                reps = 4
                t1 = time.perf_counter()
                for _ in range(reps):
                    for s in range(Wm, Wm + K):
                        h2.step_raw(xy, ts, pol, ofs[s], refs[:ns], sc2)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                sb[f"n_seq_{ns}"] = {"value": round(ns * K * reps / dt, 1), "ms_per_step": round(1e3 * dt / (K * reps), 4)}
                del h2
            sb["note"] = ("the headline advances %d sequences per GPU in lock-step; evreal_amd.eval --batch-sequences S does "
                          "the same for the sequences of a dataset" % n_seq)
            out["small_batch"] = sb

            # ---- score parity: first frames of sequence 0, GPU replay vs the CPU oracle ----
            F = min(args.parity_frames, K + Wm)
            hp.overlap_saved, hp.overlap = hp.overlap, False
            net.reset_states()
            gpu_frames = []
            for s in range(F):
                img, scs = hp.step_raw(xy, ts, pol, offs[s], refs, scratch)
                torch.cuda.synchronize()
                gpu_frames.append((img[0, 0].cpu().numpy().copy(), [float(v) for v in scratch[0].cpu().numpy()]))
            hp.overlap = hp.overlap_saved
            cb, cpu_frames = (None, [])
            if args.cpu_frames > 0:
                cb, cpu_frames = cpu_baseline(host_inputs, sd, kw, args.cpu_frames, W_, H_, lpips_sd=lpips_sd, keep_frames=F)
            out["cpu_baseline"] = cb
            out["score_parity"] = score_parity(gpu_frames, cpu_frames, ['mse', 'ssim', 'lpips'])
        else:
            out["cpu_baseline"] = None

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0 and side:
        # release this process's device memory before the sub-runs allocate theirs
        del hp, net, lp, xy, ts, pol, offs, refs, scores
        torch.cuda.empty_cache()
        fp = sub_run(['--n-seq', str(n_seq), '--sensor', args.sensor], {'EVR_FP32': '1'}, K, Wm)
        out["fp32_exact"] = ({k: fp.get(k) for k in ('value', 'ms_per_step', 'dtype', 'steady_state')} |
                             {"roofline": {k: fp.get('roofline', {}).get(k) for k in ('achieved', 'peak', 'frac', 'avg_launch_us', 'kernel')},
                              "scores": fp.get('config', {}).get('scores')}) if 'error' not in fp else fp
        if (W_, H_) == (346, 260):
            big = sub_run(['--n-seq', str(n_seq), '--sensor', '640x480'], {}, K, Wm)
            out["sensor_640x480"] = ({k: big.get(k) for k in ('value', 'ms_per_step', 'dtype', 'mevents_per_s', 'model_tflops', 'steady_state')} |
                                     {"roofline": {k: big.get('roofline', {}).get(k) for k in ('achieved', 'peak', 'frac', 'avg_launch_us')},
                                      "roofline_voxelizer": {k: big.get('roofline_voxelizer', {}).get(k) for k in ('achieved', 'frac')},
                                      "gflop_per_frame": big.get('config', {}).get('gflop_per_frame'),
                                      "sequences_per_gpu": n_seq}) if 'error' not in big else big
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so the JSON stays the LAST stdout line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Headline benchmark: reconstructed frames/s (+ Mevents/s voxelized) of the E2VID hot path at 346x260,
5 bins, 15k events/window on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n-seq S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one frame for each of S independent synthetic sequences per GPU (sequences shard across GPUs,
SURVEY 8e): raw events (resident in HBM) -> voxel grid -> event-tensor normalization -> pad -> E2VID
forward (split-bf16 x3 MFMA, fp32 accumulate; EVR_FP32=1: exact fp32 MFMA) -> crop -> robust percentile normalization -> clip -> MSE + SSIM + LPIPS against the
reference frame (LPIPS = AlexNet v0.1 structure on synthetic weights: the real ones cannot be downloaded here).
The evaluation half of a frame (robust norm, MSE/SSIM, LPIPS) runs on a second HIP stream and overlaps the
reconstruction of the next frame (`--no-overlap`: one stream); all of a step's work is inside the timed region either way.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_, H_, BINS, K_EVENTS = 346, 260, 5, 15000
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: v_mfma_f32_32x32x16_bf16 dense peak (micro-benchmark ceiling 2382)
PEAK_HBM_GBS = 8000.0


def build_inputs(rank, n_seq, n_steps, device):
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


def cpu_baseline(host_inputs, sd, kw, n_frames, budget_s=25.0, lpips_sd=None):
    """Oracle ("port") timed on the host cores, batch 1 like the reference: C voxelizer (1 thread) + numpy
    normalization + torch-CPU E2VID forward + numpy percentile + MSE + scipy SSIM + torch-CPU LPIPS.  BOUNDED: the torch thread
    count is the faster of {8, 32} (capped by the core count; all 256 threads of the GPU box's host run this
    batch-1 network ~100x slower), then frames run until `n_frames` or ~`budget_s` seconds are used."""
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

    def frame(w):
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
        omet.mse(a, b); omet.ssim(a, b)
        if lpips_sd is not None:
            with torch.no_grad():
                olp.lpips(lpips_sd, a[None], b[None])
        t5 = time.perf_counter()
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
    times = {'voxel': 0.0, 'norm': 0.0, 'forward': 0.0, 'post': 0.0, 'metrics': 0.0}
    done = 0
    while done < n_frames and (time.perf_counter() - t_start) < budget_s:
        for k, dt in zip(times, frame((done + 2) % xy.shape[0])):
            times[k] += dt
        done += 1
    total = sum(times.values())
    return {"value": round(done / total, 3), "unit": "frames/s", "cores": best, "kind": "port",
            "sample": f"{done} frames of one 346x260 sequence, batch 1, E2VID forward on {best} torch threads of the "
                      f"{os.cpu_count()}-core host (C voxelizer and numpy/scipy stages 1 thread), torch {torch.__version__}",
            "ms_per_frame": {k: round(1e3 * v / max(done, 1), 3) for k, v in times.items()},
            "mevents_per_s_voxelizer": round(K_EVENTS * done / max(times['voxel'], 1e-9) / 1e6, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--n-seq', type=int, default=64, help='independent sequences advanced together per GPU')
    ap.add_argument('--cpu-frames', type=int, default=200, help='frames of the CPU baseline (0 disables)')
    ap.add_argument('--profile-filter', default='rec', help='layers bracketed with HIP events (roofline block)')
    ap.add_argument('--no-overlap', action='store_true', help='evaluation kernels on the reconstruction stream (no second HIP stream)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
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

    from evreal_amd import model, weights
    from evreal_amd.pipeline import HotPath
    from evreal_amd.dist import reduce_metric_sums
    from evreal_amd.lpips import LPIPS

    kw = dict(weights.E2VID_KWARGS)
    sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=0)
    net = model.E2VIDRecurrent(kw)
    net.load_state_dict(sd)
    n_seq, K, Wm = args.n_seq, args.steps, args.warmup
    xy, ts, pol, offs, refs, host_inputs = build_inputs(rank, n_seq, K + Wm, device)
    lp = LPIPS(weights.synth_lpips_state_dict(seed=0))
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
    barrier()
    t0 = time.perf_counter()
    for s in range(Wm, Wm + K):
        hp.step_raw(xy, ts, pol, offs[s], refs, scores[s])
    barrier()
    elapsed = time.perf_counter() - t0
    prof = net.profile_read()
    net.profile(None)
    # outside the timed region: the same layers once more with the evaluation kernels on the SAME stream, so the
    # dominant kernel's duration is also known without the second stream's kernels sharing the chip with it
    prof_single = None
    if hp.overlap:
        hp.overlap = False
        net.profile(args.profile_filter)
        for s in range(Wm, min(Wm + 3, Wm + K)):
            hp.step_raw(xy, ts, pol, offs[s], refs, scores[s].clone())
        barrier()
        prof_single = net.profile_read()
        net.profile(None)
        hp.overlap = True

    # metric aggregation exactly as MetricTracker.update (eval.py:259-266): sum(mean*count), count
    sc = scores[Wm:].cpu().numpy()                       # [K, n_seq, 3]
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
        # arithmetic mode of the 32-channel-chunk convolutions (model.cpp finish_conv): default split-bf16
        # (x = hi + lo, three bf16 MFMA products, fp32 accumulate: fp32-equivalent to ~1e-6 relative);
        # EVR_FP32=1 selects the exact fp32 MFMA.  `achieved` counts ALGORITHMIC (direct convolution) flops in
        # both modes; in split mode the matrix cores execute 3x that, reported as mfma_issue_*.
        x3 = not os.environ.get('EVR_FP32')
        peak = PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_F32_MFMA_TFLOPS
        pmc = None
        pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path)).get('conv_igemm_lstm_bytes_per_launch')
        out = {
            "metric": "reconstructed frames/sec + Mevents/sec voxelized, E2VID 346x260 B=5",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16x3" if x3 else "f32", "data": "synthetic",
            "mevents_per_s": round(frames * K_EVENTS / elapsed / 1e6, 2),
            "model_tflops": round(flops_step * K * world / elapsed / 1e12, 2),
            "config": {"workload": "E2VID (synthetic weights, BN folded) on synthetic 346x260 Poisson events, 5 bins, "
                                   "15k events/window (k_events), %d sequences per GPU advanced together; per frame: "
                                   "voxelize(raw)+event-tensor norm+pad+forward+crop+robust norm+clip+MSE+SSIM+LPIPS "
                                   "(LPIPS: AlexNet-v0.1 structure, synthetic weights)" % n_seq,
                       "sequences_per_gpu": n_seq, "events_per_window": K_EVENTS, "sensor": [W_, H_], "bins": BINS,
                       "gflop_per_frame": round(flops_step / n_seq / 1e9, 3),
                       "lpips_gflop_per_frame": round(lp.flops() / n_seq / 1e9, 3), "sharding": "sequences across GPUs",
                       "scores": {"mse": tot[0, 0] / tot[0, 3], "ssim": tot[0, 1] / tot[0, 3], "lpips": tot[0, 2] / tot[0, 3],
                                  "count": int(tot[0, 3])}},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": pmc,
                         "kernel": ("conv3x3_band_kernel<WM=4,RING=2,LSTM=true> (ConvLSTM gate convolutions)" if x3 else
                                    "conv_igemm_kernel<32,4,4,LSTM=true,REGSTAGE,X3=0> (ConvLSTM gate convolutions)"),
                         "arithmetic": ("split bf16: x=hi+lo, w=hi+lo, acc += lo*hi + hi*lo + hi*hi on "
                                        "v_mfma_f32_32x32x16_bf16, fp32 accumulate; activations stored PACKED (bf16 hi|lo "
                                        "per 8 channels) by the producer" if x3 else
                                        "v_mfma_f32_32x32x2_f32 (exact fp32 fma chain)"),
                         "mfma_issue_tflops": round(achieved * (3 if x3 else 1), 2),
                         "mfma_issue_frac": round(achieved * (3 if x3 else 1) / peak, 4),
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
        }
        if world == 1 and args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(host_inputs, sd, kw, args.cpu_frames,
                                               lpips_sd=weights.synth_lpips_state_dict(seed=0))
        else:
            out["cpu_baseline"] = None
        line = json.dumps(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so the JSON stays the LAST stdout line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(line, flush=True)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Experiment: the 64 sequences of a GPU as G independent groups, each on its own pair of HIP streams, so one group's
low-occupancy layers (decoders, residual blocks, evaluation) run under another group's ConvLSTM layers.

    python tools/groups_bench.py [groups] [n_seq_total] [steps]
"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from evreal_amd import model, weights
from evreal_amd.pipeline import HotPath
from evreal_amd.lpips import LPIPS

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
Wm = 3
WS, HS = 346, 260
dev = torch.device('cuda', 0)
kw = dict(weights.E2VID_KWARGS)
sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=0)
n = N // G
groups = []
for g in range(G):
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        net = model.E2VIDRecurrent(kw); net.load_state_dict(sd)
        xy, ts, pol, offs, refs, _ = bench.build_inputs(g, n, K + Wm, dev, WS, HS, 15000)
        lp = LPIPS(weights.synth_lpips_state_dict(seed=0))
        hp = HotPath(net, bench.BINS, (HS, WS), n, event_tensor_normalization=True, post_process_norm='robust',
                     metrics=('mse', 'ssim', 'lpips'), device=str(dev), lpips=lp, overlap=True)
        scores = torch.zeros((K + Wm, n, 3), dtype=torch.float64, device=dev)
    groups.append((st, hp, xy, ts, pol, offs, refs, scores))
torch.cuda.synchronize()
def run(s0, s1):
    for s in range(s0, s1):
        for st, hp, xy, ts, pol, offs, refs, scores in groups:
            with torch.cuda.stream(st):
                hp.step_raw(xy, ts, pol, offs[s], refs, scores[s])
    for st, hp, *_ in groups:      # the newest frame's evaluation is held back until a successor is enqueued (HotPath.flush)
        with torch.cuda.stream(st):
            hp.flush()
run(0, Wm)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(Wm, Wm + K)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f'groups {G} x {n} sequences: {N * K / el:.1f} frames/s, {1e3 * el / K:.3f} ms per step of {N}')

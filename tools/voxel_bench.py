#!/usr/bin/env python3
"""Standalone tensorizer throughput: many 15k-event windows per launch, raw event form, vs the HBM roofline.

    python tools/voxel_bench.py [--windows 64 512 2048] [--events 15000] [--sensor 346x260]
Algorithmic bytes per window (raw form): 13*N + 4*B*H*W.  One JSON line per (windows, stats on/off).
EVR_VOX_ACC_KB=<KiB> changes the LDS budget of the cells (range size / workgroups per CU).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evreal_amd.voxel import Voxelizer   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--windows', type=int, nargs='+', default=[64, 512, 2048])
ap.add_argument('--events', type=int, default=15000)
ap.add_argument('--sensor', default='346x260')
ap.add_argument('--iters', type=int, default=20)
a = ap.parse_args()
W, H = [int(v) for v in a.sensor.split('x')]
B, n = 5, a.events
rng = np.random.default_rng(0)
nmax = max(a.windows)
N = n * nmax
xy = torch.from_numpy(np.stack([rng.integers(0, W, N), rng.integers(0, H, N)], 1).astype(np.int16)).cuda()
ts = torch.from_numpy(np.sort(rng.uniform(0, N * 1e-6, N))).cuda()
pol = torch.from_numpy(rng.integers(0, 2, N).astype(np.uint8)).cuda()
vz = Voxelizer()
for nw in a.windows:
    offs = torch.arange(nw + 1, dtype=torch.int64, device='cuda') * n
    out = torch.empty((nw, B, H, W), dtype=torch.float32, device='cuda')
    st = torch.zeros((nw, 3), dtype=torch.float64, device='cuda')
    for stats in (None, st):
        for _ in range(3):
            vz.voxelize_raw(xy, ts, pol, offs, B, (H, W), out=out, stats=stats)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            vz.voxelize_raw(xy, ts, pol, offs, B, (H, W), out=out, stats=stats)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        alg = nw * (13 * n + 4 * B * H * W)
        print(json.dumps({"windows": nw, "events_per_window": n, "sensor": a.sensor, "stats": stats is not None,
                          "acc_kb": os.environ.get('EVR_VOX_ACC_KB', 'default'), "us": round(1e3 * ms, 1),
                          "mevents_per_s": round(nw * n / ms / 1e3, 1), "algorithmic_GBs": round(alg / ms / 1e6, 1),
                          "frac_of_8TBs": round(alg / ms / 1e6 / 8000, 4)}), flush=True)
    del out, st

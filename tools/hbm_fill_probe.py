#!/usr/bin/env python3
"""Calibration for the tensorizer's roofline: what this MI355X sustains for PURE WRITES (the tensorizer writes 9x the
bytes it reads).  Times torch fill / zero / copy kernels over buffers of the voxel-grid sizes."""
import json
import torch

def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps

for mb in (115, 920, 3680):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, dtype=torch.float32, device='cuda'); b = torch.empty_like(a)
    r = {"MB": mb}
    ms = t(lambda: a.zero_()); r["zero_GBs"] = round(mb / ms, 1)
    ms = t(lambda: a.fill_(1.5)); r["fill_GBs"] = round(mb / ms, 1)
    ms = t(lambda: b.copy_(a)); r["copy_rw_GBs"] = round(2 * mb / ms, 1)
    ms = t(lambda: a.sum()); r["read_sum_GBs"] = round(mb / ms, 1)
    print(json.dumps(r), flush=True)
    del a, b

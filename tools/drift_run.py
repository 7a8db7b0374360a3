#!/usr/bin/env python3
"""Long-recurrence drift at the bench's dispatch: E2VID 346x260, 64 sequences advanced together (the wide ConvLSTM kernels,
twin-form decoders, matrix-core head -- what `python bench.py` times), N frames; sequences 0 and 37 are also run through
the CPU oracle and compared every `--every`-th frame (image max abs error; final ConvLSTM states).

    python tools/drift_run.py [--frames 1000] [--every 100] [--out gpurun_out/drift.json]      (EVR_ARITH=h3 / EVR_FP32=1: other modes)

Windows cycle through a 50-step pool per sequence (the event stream of 1000 steps x 64 sequences would be 12 GB)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--every', type=int, default=100)
    ap.add_argument('--pool', type=int, default=50)
    ap.add_argument('--out', default='gpurun_out/drift.json')
    args = ap.parse_args()
    from evreal_amd import synth
    from evreal_amd.pipeline import HotPath
    from oracle import prepost as op, voxel as ov
    dev = torch.device('cuda', 0)
    wl = bench.Workload('e2vid')
    n_seq, k, W_, H_ = 64, wl.k, wl.W, wl.H
    xy, ts, pol, offs, refs, host = bench.build_inputs(0, n_seq, args.pool, dev, W_, H_, k)
    hp = HotPath(wl.net, 5, (H_, W_), n_seq, event_tensor_normalization=True, post_process_norm='none', metrics=(), device=str(dev))
    watch = [0, 37]
    oracles = [wl.make_oracle() for _ in watch]
    crop = op.CropParams(W_, H_, wl.enc)
    torch.set_num_threads(min(32, os.cpu_count()))
    hxy, hts, hpol, _ = host
    rows, worst = [], 0.0
    t0 = time.time()
    for f in range(args.frames):
        w = f % args.pool
        img, _ = hp.step_raw(xy, ts, pol, offs[w])
        check = (f + 1) % args.every == 0 or f == 0
        got = img[watch, 0].cpu().numpy() if check else None
        for j, s in enumerate(watch):
            xs, ys, tf, ps = synth.window_events_f32(hts[w, s], hxy[w, s], hpol[w, s], 0, k)
            v = op.normalize_event_tensor(ov.events_to_voxel(xs, ys, tf, ps, 5, (H_, W_))[None])
            with torch.no_grad():
                want = crop.crop(oracles[j](torch.from_numpy(crop.pad(v))).numpy())[0, 0]
            if check:
                e = float(np.abs(got[j] - want).max())
                worst = max(worst, e)
                rows.append({"frame": f + 1, "sequence": s, "image_max_abs_err": e})
        if check:
            print(f"frame {f + 1}: " + ", ".join(f"seq {r['sequence']} {r['image_max_abs_err']:.2e}" for r in rows[-len(watch):]), flush=True)
    states = {}
    for j, s in enumerate(watch):
        for i in range(3):
            h = wl.net.read_tensor(f'h{i}').cpu().numpy().reshape(n_seq, *oracles[j].states[i][0].shape[1:])[s]
            c = wl.net.read_tensor(f'c{i}').cpu().numpy().reshape(n_seq, *oracles[j].states[i][1].shape[1:])[s]
            states[f"seq{s}.h{i}"] = float(np.abs(h - oracles[j].states[i][0].numpy()[0]).max())
            states[f"seq{s}.c{i}"] = float(np.abs(c - oracles[j].states[i][1].numpy()[0]).max())
    runs, layer = wl.net.saturation()
    out = {"what": "E2VID 346x260, 64 sequences in lock-step (bench.py's dispatch), fused event-tensor normalization, sigmoid output",
           "arithmetic": bench.arith_name(), "frames": args.frames, "checked_every": args.every, "watched_sequences": watch,
           "worst_image_max_abs_err": worst, "gate": 1e-4, "ok": bool(worst < 1e-4), "checks": rows,
           "final_state_max_abs_err": states, "range_guard_runs": runs, "seconds": round(time.time() - t0, 1)}
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(out, open(args.out, 'w'), indent=1)
    print(json.dumps({k_: out[k_] for k_ in ('arithmetic', 'frames', 'worst_image_max_abs_err', 'ok', 'final_state_max_abs_err', 'seconds')}))


if __name__ == '__main__':
    main()

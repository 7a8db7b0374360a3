#!/usr/bin/env python3
"""Latency experiment: the whole two-frame step captured in one HIP graph (torch.cuda.CUDAGraph) vs eager launches.

    python tools/graph_bench.py [n_seq] [steps]
"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from evreal_amd import model, weights
from evreal_amd.pipeline import HotPath
from evreal_amd.lpips import LPIPS

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device('cuda', 0)
kw = dict(weights.E2VID_KWARGS)
net = model.E2VIDRecurrent(kw); net.load_state_dict(weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=0))
xy, ts, pol, offs, refs, _ = bench.build_inputs(0, n_seq, K + 4, dev)
lp = LPIPS(weights.synth_lpips_state_dict(seed=0))
hp = HotPath(net, bench.BINS, (bench.H_, bench.W_), n_seq, event_tensor_normalization=True, post_process_norm='robust',
             metrics=('mse', 'ssim', 'lpips'), device=str(dev), lpips=lp, overlap=False)
scores = torch.zeros((K + 4, n_seq, 3), dtype=torch.float64, device=dev)
for s in range(4):
    hp.step_raw(xy, ts, pol, offs[s], refs, scores[s])
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(4, 4 + K):
    hp.step_raw(xy, ts, pol, offs[s], refs, scores[s])
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / K
ref_scores = scores[4:4 + K].clone()

# graph: two frames (both ping-pong parities of the model) per replay, static argument buffers
offs_st = torch.zeros((2, n_seq + 1), dtype=torch.int64, device=dev)
sc_st = torch.zeros((2, n_seq, 3), dtype=torch.float64, device=dev)
net.reset_states()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for s in range(4):      # warm-up on the capture stream, states back in step with the eager run
        offs_st[0].copy_(offs[s]); hp.step_raw(xy, ts, pol, offs_st[0], refs, sc_st[0])
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        hp.step_raw(xy, ts, pol, offs_st[0], refs, sc_st[0])
        hp.step_raw(xy, ts, pol, offs_st[1], refs, sc_st[1])
    out = torch.zeros_like(scores)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(4, 4 + K, 2):
        offs_st.copy_(offs[s:s + 2]); g.replay(); out[s:s + 2].copy_(sc_st)
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / K
same = torch.equal(out[4:4 + K], ref_scores)
print(f'n_seq {n_seq}: eager {1e3 * eager:.3f} ms/step ({n_seq / eager:.0f} frames/s), graph {1e3 * graph:.3f} ms/step ({n_seq / graph:.0f} frames/s), identical scores: {same}')

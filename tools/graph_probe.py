"""Does replaying the one-sequence step as a HIP graph help?    The whole single-stream step -- tensorizer, network,
robust normalisation, MSE/SSIM, LPIPS: ~60 launches -- is captured once (torch.cuda.CUDAGraph over the library's launches on torch's
stream) and replayed; eager = the same calls issued one by one.      python tools/graph_probe.py [n_seq ...]"""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from evreal_amd.pipeline import HotPath
from evreal_amd.lpips import LPIPS
from evreal_amd import weights

dev = torch.device('cuda', 0)
for n_seq in [int(a) for a in sys.argv[1:]] or [1, 4]:
    wl = bench.Workload('e2vid')
    xy, ts, pol, offs, refs, host = bench.build_inputs(0, n_seq, 8, dev, wl.W, wl.H, wl.k)
    lp = LPIPS(weights.synth_lpips_state_dict(seed=0))
    hp = HotPath(wl.net, 5, (wl.H, wl.W), n_seq, event_tensor_normalization=True, post_process_norm='robust',
                 metrics=('mse', 'ssim', 'lpips'), device=str(dev), lpips=lp, overlap=False)
    scores = torch.zeros((n_seq, 3), dtype=torch.float64, device=dev)
    step = lambda: hp.step_raw(xy, ts, pol, offs[0], refs, scores, n_window_events=n_seq * wl.k)
    for _ in range(20): step()
    torch.cuda.synchronize()
    K = 400
    t0 = time.perf_counter()
    for _ in range(K): step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / K
    ref_scores = scores.clone()
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            for _ in range(3): step()          # (warm on the capture stream: plans, attributes, workspaces)
            side.synchronize()
            with torch.cuda.graph(g, stream=side):
                step(); step()                 # two frames: both ping-pong parities of the recurrent state
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K // 2): g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / K
        print(f'n_seq {n_seq}: eager single-stream {1e6 * eager:.0f} us/step ({n_seq / eager:.0f} frames/s), HIP-graph replay {1e6 * graph:.0f} us/step '
              f'({n_seq / graph:.0f} frames/s)')
    except Exception as e:
        print(f'n_seq {n_seq}: eager single-stream {1e6 * eager:.0f} us/step; graph capture failed: {type(e).__name__}: {str(e)[:300]}')

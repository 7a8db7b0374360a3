import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from evreal_amd.pipeline import HotPath
from evreal_amd.lpips import LPIPS
from evreal_amd import weights
dev = torch.device('cuda', 0)
for n_seq in (1, 4):
    wl = bench.Workload('e2vid')
    xy, ts, pol, offs, refs, host = bench.build_inputs(0, n_seq, 40, dev, wl.W, wl.H, wl.k)
    lp = LPIPS(weights.synth_lpips_state_dict(seed=0))
    hp = HotPath(wl.net, 5, (wl.H, wl.W), n_seq, event_tensor_normalization=True, post_process_norm='robust', metrics=('mse', 'ssim', 'lpips'), device=str(dev), lpips=lp, overlap=True)
    scores = torch.zeros((n_seq, 3), dtype=torch.float64, device=dev)
    def run(K):
        for s in range(K):
            hp.step_raw(xy, ts, pol, offs[s % 40], refs, scores, n_window_events=n_seq * wl.k)
        hp.flush()
    run(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(400); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'n_seq {n_seq}: host enqueue {1e6 * (t1 - t0) / 400:.0f} us/step, total {1e6 * (t2 - t0) / 400:.0f} us/step')

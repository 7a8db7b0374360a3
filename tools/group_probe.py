"""Do G concurrent groups of S/G sequences (each group its own model handle and streams, stepped in turn by one host thread) beat ONE
step over S sequences at small batches?  The launches of a small batch leave most CUs idle (tile rounds, tails, launch boundaries);
a second group's launches could fill them.      python tools/group_probe.py [S ...]"""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from evreal_amd.pipeline import HotPath
from evreal_amd.lpips import LPIPS
from evreal_amd import weights

dev = torch.device('cuda', 0)


def make(n_seq, seed_rank):
    wl = bench.Workload('e2vid')
    xy, ts, pol, offs, refs, host = bench.build_inputs(seed_rank, n_seq, 8, dev, wl.W, wl.H, wl.k)
    lp = LPIPS(weights.synth_lpips_state_dict(seed=0))
    hp = HotPath(wl.net, 5, (wl.H, wl.W), n_seq, event_tensor_normalization=True, post_process_norm='robust',
                 metrics=('mse', 'ssim', 'lpips'), device=str(dev), lpips=lp, overlap=True)
    scores = torch.zeros((n_seq, 3), dtype=torch.float64, device=dev)
    stream = torch.cuda.Stream(device=dev)

    def step(i):
        with torch.cuda.stream(stream):
            hp.step_raw(xy, ts, pol, offs[i % 8], refs, scores, n_window_events=n_seq * wl.k)
    return step, hp, wl


def rate(steps_fns, n_frames_per_round, K=300):
    for i in range(20):
        for f in steps_fns: f(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        for f in steps_fns: f(i)
    torch.cuda.synchronize()
    return n_frames_per_round * K / (time.perf_counter() - t0)


def rate_threads(steps_fns, n_frames_per_round, K=300):
    """every group stepped by its own host thread (the library calls release the GIL)"""
    import threading
    for i in range(20):
        for f in steps_fns: f(i)
    torch.cuda.synchronize()
    def run(f):
        for i in range(K): f(i)
    th = [threading.Thread(target=run, args=(f,)) for f in steps_fns]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return n_frames_per_round * K / (time.perf_counter() - t0)


for S in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    one = make(S, 0)
    r1 = rate([one[0]], S)
    del one
    out = [f'S={S}: one step over {S}: {r1:.0f} frames/s']
    for G in (2, 4, 8):
        if S % G or S // G < 1: continue
        groups = [make(S // G, g) for g in range(G)]
        rg = rate([g[0] for g in groups], S)
        rt = rate_threads([g[0] for g in groups], S)
        out.append(f'{G} groups of {S // G}: {rg:.0f} (one host thread) / {rt:.0f} (a thread per group)')
        del groups
    print(';  '.join(out), flush=True)

#!/usr/bin/env python3
"""Where the drop-in call's set-up time goes: bench.py's eval_cli tree (8 sequences x 160 frames, 346x260) under cProfile.
    python tools/eval_cli_profile.py [out.txt]"""
import cProfile, io, os, pstats, sys, time, contextlib, json, shutil, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from evreal_amd import eval as ev, synth, weights

n_seq, frames, W_, H_ = 8, 160, 346, 260
kw = dict(weights.E2VID_KWARGS)
sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=0)
tmp = tempfile.mkdtemp(prefix='evr_cli_')
for sub in ('eval', 'method', 'dataset'):
    os.makedirs(os.path.join(tmp, 'config', sub))
os.makedirs(os.path.join(tmp, 'pretrained'))
torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in weights.synth_lpips_state_dict(seed=0).items()}, os.path.join(tmp, 'pretrained', 'lpips_alex.pth'))
torch.save({'model': {k: v for k, v in kw.items() if k != 'final_activation'}, 'state_dict': {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}, os.path.join(tmp, 'e2vid.pth'))
json.dump({"model_name": "E2VID", "model_path": os.path.join(tmp, 'e2vid.pth'), "event_tensor_normalization": True, "post_process_norm": "robust"}, open(os.path.join(tmp, 'config/method/E2VID.json'), 'w'))
json.dump({"dataset_kwargs": {"num_bins": 5, "voxel_method": {"method": "between_frames"}, "keep_ratio": 1.0}, "save_images": False, "histeq": "none", "eval_infer_all": False, "ts_tol_ms": 1.0, "create_video": False, "batch_sequences": int(os.environ.get("EVR_PROFILE_BS", n_seq))}, open(os.path.join(tmp, 'config/eval/std.json'), 'w'))
seqs = {}
for s in range(n_seq):
    synth.write_sequence(os.path.join(tmp, 'data', 'SYN', f's{s}'), 100 + s, (frames + 1) * 15000, 1.0e6, W_, H_, 1.0e6 / 15000)
    seqs[f's{s}'] = {}
json.dump({"root_path": os.path.join(tmp, 'data', 'SYN'), "sequences": seqs}, open(os.path.join(tmp, 'config/dataset/SYN.json'), 'w'))
os.chdir(tmp)
out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else sys.stdout
for rep in range(3):
    shutil.rmtree('outputs', ignore_errors=True)
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr.enable()
    with contextlib.redirect_stdout(io.StringIO()):
        ev.evaluate(['E2VID'], ['std'], ['SYN'], ['mse', 'ssim', 'lpips'])
    torch.cuda.synchronize()
    pr.disable()
    dt = time.perf_counter() - t0
    print(f'--- pass {rep}: {dt:.3f} s, TIMINGS {ev.TIMINGS[-1]}', file=out)
    if rep == 2:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(70); print(s.getvalue(), file=out)
shutil.rmtree(tmp, ignore_errors=True)

#!/usr/bin/env python3
"""Per-tap error of one golden E2VID sequence (first frame): where a numerical difference enters the network."""
import json, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from evreal_amd import model, synth, weights
tag = sys.argv[1] if len(sys.argv) > 1 else 'e2vid_bn'
z = np.load(os.path.join(ROOT, 'tests', 'golden', f'{tag}_seq.npz'))
kw = json.loads(bytes(z['kwargs']).decode())
fixed = {k[6:]: z[k] for k in z.files if k.startswith('fixed.')}
sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=int(z['seed']), fixed=fixed)
m = model.E2VIDRecurrent(kw); m.debug_taps = True; m.load_state_dict(sd)
seed, F, B, H, W = [int(v) for v in z['voxel_args']]
vox = synth.sparse_voxels(seed, F, B, H, W)
m.reset_states()
img = m(torch.from_numpy(vox[0:1]).cuda())['image'].cpu().numpy()
print('image', np.abs(img - z['images'][0:1]).max())
for k in [k for k in z.files if k.startswith('tap.')]:
    name = k[4:]
    dname = {'enc0.h': 'h0', 'enc2.h': 'h2'}.get(name, name)
    try:
        got = m.read_tensor(dname).cpu().numpy()
    except Exception as e:
        print(name, 'ERR', e); continue
    want = z[k]
    got = got.reshape(1, -1, want.shape[2], want.shape[3])
    got = got[:, ::4] if got.shape[1] >= 32 else got
    d = np.abs(got - want)
    print(f'{name:12s} max|d| {d.max():.3e}  rel {d.max() / (np.abs(want).max() + 1e-30):.3e}  shape {want.shape}')
if len(sys.argv) > 2:
    name = sys.argv[2]
    want = z['tap.' + name]
    got = m.read_tensor({'enc0.h': 'h0', 'enc2.h': 'h2'}.get(name, name)).cpu().numpy().reshape(1, -1, want.shape[2], want.shape[3])[:, ::4]
    d = np.abs(got - want)[0]
    print('per (sub)channel max err:', ' '.join(f'{v:.1e}' for v in d.reshape(d.shape[0], -1).max(1)))
    print('per row max err:', ' '.join(f'{v:.1e}' for v in d.max(axis=(0, 2))))
    print('per col max err:', ' '.join(f'{v:.1e}' for v in d.max(axis=(0, 1))))

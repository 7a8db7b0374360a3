"""Diagnostic: per-tap error of an E2VID golden layout (python tools/tap_diag.py <tag>)."""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
from conftest import load_npz
from evreal_amd import model, synth, weights

tag = sys.argv[1]
z = load_npz(f'{tag}_seq.npz')
kw = json.loads(bytes(z['kwargs']).decode())
fixed = {k[6:]: z[k] for k in z.files if k.startswith('fixed.')}
sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=int(z['seed']), fixed=fixed)
m = model.E2VIDRecurrent(kw); m.debug_taps = True; m.load_state_dict(sd)
seed, F, B, H, W = [int(v) for v in z['voxel_args']]
vox = synth.sparse_voxels(seed, F, B, H, W)
m.reset_states()
for f in range(F):
    img = m(torch.from_numpy(vox[f:f + 1]).cuda())['image'].cpu().numpy()
    print('frame', f, 'img err', np.abs(img - z['images'][f:f + 1]).max())
    if f == 0:
        for k in [k for k in z.files if k.startswith('tap.')]:
            name = k[4:]
            dname = {'enc0.h': 'h0', 'enc2.h': 'h2'}.get(name, name)
            got = m.read_tensor(dname).cpu().numpy(); want = z[k]
            got = got.reshape(1, -1, want.shape[2], want.shape[3]); got = got[:, ::4] if got.shape[1] >= 32 else got
            d = np.abs(got - want)
            print(' ', k, 'max err', d.max(), 'max |want|', np.abs(want).max(), 'worst ch', np.unravel_index(d.argmax(), d.shape))

"""Diagnostic: the bilinear x2 (x + skip) step of an upsample-conv layout against numpy (python tools/up_diag.py)."""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
from conftest import load_npz
from evreal_amd import model, synth, weights
z = load_npz('e2vid_plus_seq.npz')
kw = json.loads(bytes(z['kwargs']).decode())
sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=int(z['seed']))
m = model.E2VIDRecurrent(kw); m.debug_taps = True; m.load_state_dict(sd)
seed, F, B, H, W = [int(v) for v in z['voxel_args']]
vox = synth.sparse_voxels(seed, F, B, H, W)
m.reset_states()
m(torch.from_numpy(vox[0:1]).cuda())
print('arith', m.arith)
hh, ww = H // 8, W // 8
res = m.read_tensor('res1').cpu().numpy().reshape(1, -1, hh, ww); h2 = m.read_tensor('h2').cpu().numpy().reshape(1, -1, hh, ww); up = m.read_tensor('dec0.up').cpu().numpy()
x = torch.from_numpy(res + h2)
want = torch.nn.functional.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False).numpy()
up = up.reshape(want.shape)
d = np.abs(up - want)
print('up err', d.max(), 'max|want|', np.abs(want).max(), 'worst', np.unravel_index(d.argmax(), d.shape))
print('per-channel max err (first 32):', np.round(d.max(axis=(0, 2, 3))[:32], 5))
dec0 = m.read_tensor('dec0').cpu().numpy(); print('dec0 shape', dec0.shape, 'tap err', np.abs(dec0.reshape(1, -1, 16, 24)[:, ::4] - z['tap.dec0']).max())

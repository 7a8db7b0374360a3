"""Pin the network oracles (torch-CPU functional restatements) to the reference goldens:
FireNet / FireNet+ with the REAL shipped weights, E2VID layouts with synthetic weights."""
import json

import numpy as np
import torch

from oracle import model as omod
from oracle.prepost import CropParams
from evreal_amd import synth, weights
from conftest import load_npz
from golden_inputs import sha

TOL = dict(rtol=1e-5, atol=1e-5)


def _sd(npz):
    return {k: torch.from_numpy(npz[k]) for k in npz.files}


def _run_fire(tag, cls):
    z = load_npz(f'{tag}_seq.npz')
    w = load_npz(f'{tag}_weights.npz')
    assert weights.state_dict_digest({k: w[k] for k in w.files}) == str(z['weights_sha'])
    m = cls(_sd(w))
    assert m.num_encoders == int(z['num_encoders'])
    seed, F, B, H, W = [int(v) for v in z['voxel_args']]
    vox = synth.sparse_voxels(seed, F, B, H, W)
    assert sha(vox) == str(z['voxel_sha'])
    crop = CropParams(W, H, m.num_encoders)
    torch.set_num_threads(1)
    for f in range(F):
        x = torch.from_numpy(crop.pad(vox[f:f + 1]))
        img = crop.crop(m(x).numpy())
        np.testing.assert_allclose(img, z['images'][f:f + 1], **TOL)
    np.testing.assert_allclose(m.states[0].numpy()[:, :, ::3, ::3], z['state0_sub'], **TOL)
    np.testing.assert_allclose(m.states[1].numpy()[:, :, ::3, ::3], z['state1_sub'], **TOL)


def test_firenet_legacy_real_weights():
    _run_fire('firenet', omod.FireNetLegacyOracle)


def test_firenet_plus_real_weights():
    _run_fire('firenetplus', omod.FireNetOracle)


def _run_e2vid(tag):
    z = load_npz(f'{tag}_seq.npz')
    kw = json.loads(bytes(z['kwargs']).decode())
    fixed = {k[6:]: z[k] for k in z.files if k.startswith('fixed.')}
    sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=int(z['seed']), fixed=fixed)
    assert weights.state_dict_digest(sd) == str(z['weights_sha'])
    okw = {k: kw[k] for k in ['num_bins', 'base_num_channels', 'num_encoders', 'num_residual_blocks', 'kernel_size',
                              'norm', 'use_upsample_conv', 'recurrent_block_type', 'final_activation']}
    okw['use_dynamic_decoder'] = kw.get('use_dynamic_decoder', False)
    m = omod.UNetRecurrentOracle({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, **okw)
    seed, F, B, H, W = [int(v) for v in z['voxel_args']]
    vox = synth.sparse_voxels(seed, F, B, H, W)
    assert sha(vox) == str(z['voxel_sha'])
    torch.set_num_threads(1)
    for f in range(F):
        taps = {} if f == 0 else None
        img = m(torch.from_numpy(vox[f:f + 1]), taps).numpy()
        np.testing.assert_allclose(img, z['images'][f:f + 1], **TOL)
        if f == 0:
            for k in [k for k in z.files if k.startswith('tap.')]:
                name = k[4:]
                got = taps['enc0.h' if name == 'enc0.h' else name] if name in taps else None
                if name == 'enc0.h':
                    got = m.states[0][0] if isinstance(m.states[0], tuple) else m.states[0]
                if name == 'enc2.h':
                    got = m.states[2][0]
                got = got.numpy()
                got = got[:, ::4] if got.shape[1] >= 32 else got
                np.testing.assert_allclose(got, z[k], rtol=1e-4, atol=1e-5, err_msg=k)
    for i, s in enumerate(m.states):
        h = s[0] if isinstance(s, tuple) else s
        np.testing.assert_allclose(h.numpy()[:, ::4], z[f'h{i}_sub'], rtol=1e-4, atol=1e-5)
        if isinstance(s, tuple):
            np.testing.assert_allclose(s[1].numpy()[:, ::4], z[f'c{i}_sub'], rtol=1e-4, atol=1e-5)


def test_e2vid_bn_layout():
    _run_e2vid('e2vid_bn')


def test_e2vid_plus_layout():
    _run_e2vid('e2vid_plus')


def test_e2vid_gru_tiny():
    _run_e2vid('e2vid_gru_tiny')


def test_e2vid_hyper_dynamic_decoder():
    _run_e2vid('e2vid_hyper')


def test_e2vid_instance_norm_layout():
    """norm='IN' branch of the oracle (running-statistics InstanceNorm in the conv layers, a true InstanceNorm2d in the
    residual blocks; model/submodules.py:22-23,160-162) against the reference class's golden."""
    _run_e2vid('e2vid_in')


def test_e2vid_hyper_instance_norm_layout():
    """The dynamic decoder beside norm='IN' layers (DynamicUpsampleLayer carries no norm itself, submodules.py:100-127)."""
    _run_e2vid('e2vid_hyper_in')


def test_spade_e2vid_oracle_golden():
    """SpadeE2vidOracle against the reference class SpadeE2vid (model/spade_e2v.py:113-179): images, the 3-channel
    prev_recs and every ConvLSTM state of the 4-frame golden."""
    z = load_npz('spade_seq.npz')
    sd = weights.synth_state_dict(weights.spade_e2vid_schema(), seed=int(z['seed']))
    assert weights.state_dict_digest(sd) == str(z['weights_sha'])
    m = omod.SpadeE2vidOracle({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    seed, F, B, H, W = [int(v) for v in z['voxel_args']]
    vox = synth.sparse_voxels(seed, F, B, H, W, density=0.15)
    assert sha(vox) == str(z['voxel_sha'])
    torch.set_num_threads(1)
    for f in range(F):
        img = m(torch.from_numpy(vox[f:f + 1])).numpy()
        np.testing.assert_allclose(img, z['images'][f:f + 1], **TOL, err_msg=f'frame {f}')
    np.testing.assert_allclose(m.prev_recs.numpy(), z['prev_recs'], **TOL)
    for i, (h, c) in enumerate(m.states):
        np.testing.assert_allclose(h.numpy()[:, ::4], z[f'h{i}_sub'], rtol=1e-4, atol=1e-5, err_msg=f'h{i}')
        np.testing.assert_allclose(c.numpy()[:, ::4], z[f'c{i}_sub'], rtol=1e-4, atol=1e-5, err_msg=f'c{i}')


import pytest


@pytest.mark.parametrize('norm,tag', [(None, 'etnet'), ('BN', 'etnet_bn'), ('IN', 'etnet_in')])
def test_etnet_oracle_golden(norm, tag):
    """ETNetOracle against the reference class EITR (model/eitr/eitr.py:4-16, u_trans.py:13-123): images and encoder
    states of the 3-frame goldens, for every `norm` the reference's ConvLayers accept."""
    z = load_npz(f'{tag}_seq.npz')
    sd = weights.synth_state_dict(weights.etnet_schema(norm=norm), seed=int(z['seed']))
    assert weights.state_dict_digest(sd) == str(z['weights_sha'])
    m = omod.ETNetOracle({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if np.asarray(v).dtype.kind == 'f'}, norm=norm)
    seed, F, B, H, W = [int(v) for v in z['voxel_args']]
    vox = synth.sparse_voxels(seed, F, B, H, W, density=0.15)
    assert sha(vox) == str(z['voxel_sha'])
    torch.set_num_threads(1)
    for f in range(F):
        img = m(torch.from_numpy(vox[f:f + 1])).numpy()
        np.testing.assert_allclose(img, z['images'][f:f + 1], **TOL, err_msg=f'frame {f}')
    for i, (h, c) in enumerate(m.states):
        np.testing.assert_allclose(h.numpy()[:, ::4], z[f'h{i}_sub'], rtol=1e-4, atol=1e-5, err_msg=f'h{i}')
        np.testing.assert_allclose(c.numpy()[:, ::4], z[f'c{i}_sub'], rtol=1e-4, atol=1e-5, err_msg=f'c{i}')

"""GPU parity: the recurrent networks (HIP implicit-GEMM convolutions through the C ABI) vs the oracle
and the reference goldens.  Tolerance: 1e-4 absolute per pixel on images (north_star), 1e-4 relative /
1e-5 absolute on states -- fp32 everywhere, only the summation order differs from the reference."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_npz
from golden_inputs import sha

pytestmark = pytest.mark.gpu
# the default arithmetic (three f16 products, fp32-grade) is gated at 1e-5 per pixel; the opt-in fast modes (EVR_ARITH=mx6|mx) and the
# exact-fp32 mode (whose own summation order differs from the oracle's) at north_star's 1e-4
ARITH = 'fp32' if (os.environ.get('EVR_FP32') or os.environ.get('EVR_ARITH') == 'fp32') else (os.environ.get('EVR_ARITH') or 'h3')
IMG_ATOL = float(os.environ.get('EVR_TEST_IMG_ATOL', '1e-5' if ARITH == 'h3' else '1e-4'))


def _fire(tag, cls_name):
    from evreal_amd import model, synth, weights
    z = load_npz(f'{tag}_seq.npz')
    w = load_npz(f'{tag}_weights.npz')
    sd = {k: w[k] for k in w.files}
    assert weights.state_dict_digest(sd) == str(z['weights_sha'])
    if cls_name == 'FireNet_legacy':
        m = model.FireNet_legacy(unet_kwargs=dict(num_bins=5, skip_type='no_skip', recurrent_block_type='convgru',
                                                  base_num_channels=16, num_residual_blocks=2,
                                                  recurrent_blocks={'resblock': [0]}, kernel_size=3, norm='none'))
    else:
        m = model.FireNet(num_bins=5, base_num_channels=16, kernel_size=3)
    m.load_state_dict(sd)
    assert m.num_encoders == int(z['num_encoders'])
    seed, F, B, H, W = [int(v) for v in z['voxel_args']]
    vox = synth.sparse_voxels(seed, F, B, H, W)
    assert sha(vox) == str(z['voxel_sha'])
    m.reset_states()
    for f in range(F):
        img = m(torch.from_numpy(vox[f:f + 1]).cuda())['image'].cpu().numpy()
        np.testing.assert_allclose(img, z['images'][f:f + 1], rtol=0, atol=IMG_ATOL, err_msg=f'frame {f}')
    # states live on the padded grid; the golden stores a ::3 subsample of the reference's padded state
    hp, wp = ((H + 15) // 16 * 16, (W + 15) // 16 * 16) if cls_name == 'FireNet_legacy' else (H, W)
    # (EVR_FIRENET_PAD32=1 in the default split mode: the states themselves are stored PACKED, ~2^-16 relative per product term)
    split_mx = os.environ.get('EVR_FIRENET_PAD32', '0') not in ('', '0') and ARITH in ('mx', 'mx6')
    for i in range(2):
        h = m.read_tensor(f'h{i}').cpu().numpy().reshape(1, 16, hp, wp)
        np.testing.assert_allclose(h[:, :, ::3, ::3], z[f'state{i}_sub'], rtol=1e-4, atol=2e-4 if split_mx else 1e-5)


def test_firenet_legacy_real_weights():
    _fire('firenet', 'FireNet_legacy')


def test_firenet_plus_real_weights():
    _fire('firenetplus', 'FireNet')


def _e2vid(tag, n_seq=1):
    from evreal_amd import model, synth, weights
    z = load_npz(f'{tag}_seq.npz')
    kw = json.loads(bytes(z['kwargs']).decode())
    fixed = {k[6:]: z[k] for k in z.files if k.startswith('fixed.')}
    sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=int(z['seed']), fixed=fixed)
    assert weights.state_dict_digest(sd) == str(z['weights_sha'])
    m = model.E2VIDRecurrent(kw)
    m.debug_taps = (n_seq == 1)      # taps of the last decoder need its (otherwise never stored) output
    m.load_state_dict(sd)
    seed, F, B, H, W = [int(v) for v in z['voxel_args']]
    vox = synth.sparse_voxels(seed, F, B, H, W)
    assert sha(vox) == str(z['voxel_sha'])
    m.reset_states()
    for f in range(F):
        x = torch.from_numpy(vox[f:f + 1]).cuda()
        if n_seq > 1:   # replicate the sequence: every replica must reproduce the golden
            x = x.repeat(n_seq, 1, 1, 1)
        img = m(x)['image'].cpu().numpy()
        # (a tightened gate of the fp32-grade modes never goes below 8x the reference's own fp32-vs-float64 image spread)
        img_atol = max(IMG_ATOL, 8.0 * float(z['cond.images'])) if 'cond.images' in z.files else IMG_ATOL
        for s in range(n_seq):
            np.testing.assert_allclose(img[s:s + 1], z['images'][f:f + 1], rtol=0, atol=img_atol, err_msg=f'frame {f} seq {s}')
        if f == 0 and n_seq == 1:
            for k in [k for k in z.files if k.startswith('tap.')]:
                name = k[4:]
                if name in ('res1', 'dec0', 'dec2') and not kw['use_upsample_conv']:
                    continue          # those buffers hold the fused skip-sum; covered by the later layers
                dname = {'enc0.h': 'h0', 'enc2.h': 'h2'}.get(name, name)
                got = m.read_tensor(dname).cpu().numpy()
                want = z[k]
                got = got.reshape(1, -1, want.shape[2], want.shape[3])
                got = got[:, ::4] if got.shape[1] >= 32 else got
                # norm='IN' goldens carry the reference's own fp32-vs-float64 spread per tap (make_golden.py): behind a true
                # InstanceNorm2d over an 8 x 12 map a 1e-7 difference in the ConvLSTM state becomes 1e-4, for the reference as
                # for us -- those taps get a multiple of that spread; the IMAGE gate above is the same for every layout
                cond = float(z['cond.' + name]) if ('cond.' + name) in z.files else 0.0
                np.testing.assert_allclose(got, want, rtol=1e-4, atol=max(2e-5, 64.0 * cond), err_msg=k)
    for i in range(kw['num_encoders']):
        want = z[f'h{i}_sub']
        h = m.read_tensor(f'h{i}').cpu().numpy().reshape(n_seq, -1, want.shape[2], want.shape[3])
        np.testing.assert_allclose(h[:1, ::4], want, rtol=1e-4, atol=2e-5, err_msg=f'h{i}')
        if f'c{i}_sub' in z.files:
            c = m.read_tensor(f'c{i}').cpu().numpy().reshape(n_seq, -1, want.shape[2], want.shape[3])
            np.testing.assert_allclose(c[:1, ::4], z[f'c{i}_sub'], rtol=1e-4, atol=2e-5, err_msg=f'c{i}')


def test_e2vid_bn_layout():
    _e2vid('e2vid_bn')


def test_e2vid_plus_layout():
    _e2vid('e2vid_plus')


def test_e2vid_gru_tiny():
    _e2vid('e2vid_gru_tiny')


def test_e2vid_instance_norm_layout():
    """norm='IN' (model/submodules.py:22-23,160-162): running-statistics InstanceNorm folded into the conv layers, a true
    InstanceNorm2d kernel inside the residual blocks -- against the reference class."""
    _e2vid('e2vid_in')


def test_e2vid_hyper_dynamic_decoder():
    _e2vid('e2vid_hyper')


def test_e2vid_hyper_instance_norm_layout():
    """use_dynamic_decoder with norm='IN': IN-folded encoders / decoders 1-2, true InstanceNorm2d residual blocks feeding the
    dynamic decoder (submodules.py:100-127 carries no norm)."""
    _e2vid('e2vid_hyper_in')
    _e2vid('e2vid_hyper_in', n_seq=2)


def test_e2vid_hyper_batched_sequences():
    _e2vid('e2vid_hyper', n_seq=2)


def test_e2vid_bn_batched_sequences():
    _e2vid('e2vid_bn', n_seq=3)


def test_e2vid_full_size_vs_oracle_with_padding_and_norm():
    """346x260 (pads to 352x264), fused event-tensor normalization, 3 frames, against the torch-CPU oracle."""
    from evreal_amd import model, synth, weights
    from evreal_amd.voxel import Voxelizer
    from oracle import model as omod, prepost as op, voxel as ov
    from golden_inputs import gen_events
    kw = dict(weights.E2VID_KWARGS)
    sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=3)
    m = model.E2VIDRecurrent(kw); m.load_state_dict(sd)
    okw = {k: kw[k] for k in ['num_bins', 'base_num_channels', 'num_encoders', 'num_residual_blocks', 'kernel_size',
                              'norm', 'use_upsample_conv', 'recurrent_block_type', 'final_activation']}
    o = omod.UNetRecurrentOracle({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, **okw)
    H, W = 260, 346
    crop = op.CropParams(W, H, 3)
    vz = Voxelizer()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    m.reset_states()
    for f in range(3):
        x, y, t, p = gen_events(900 + f, 15000, W, H)
        st = torch.zeros((1, 3), dtype=torch.float64, device='cuda')
        g = vz.voxelize(d(x), d(y), d(t), d(p), d(np.array([0, 15000], dtype=np.int64)), 5, (H, W), stats=st)
        img = m(g, stats=st)['image'].cpu().numpy()
        v = op.normalize_event_tensor(ov.events_to_voxel(x, y, t, p, 5, (H, W))[None])
        want = crop.crop(o(torch.from_numpy(crop.pad(v))).numpy())
        np.testing.assert_allclose(img, want, rtol=0, atol=IMG_ATOL, err_msg=f'frame {f}')


def test_e2vid_full_size_batch_takes_the_band_kernels():
    """The same 346x260 comparison with 8 sequences advanced together: at this size the launches clear the band
    kernels' fill threshold by themselves (no EVR_BAND_MIN), so the dispatch bench.py times is the one checked here --
    ConvLSTM / residual / transposed layers on conv3x3_band_kernel, the 64-column encoder on the programmed band
    kernel, the head on the matrix cores.  Every replica must match the oracle's single-sequence result."""
    from evreal_amd import model, weights
    from evreal_amd.voxel import Voxelizer
    from oracle import model as omod, prepost as op, voxel as ov
    from golden_inputs import gen_events
    kw = dict(weights.E2VID_KWARGS)
    sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=5)
    m = model.E2VIDRecurrent(kw); m.load_state_dict(sd)
    okw = {k: kw[k] for k in ['num_bins', 'base_num_channels', 'num_encoders', 'num_residual_blocks', 'kernel_size',
                              'norm', 'use_upsample_conv', 'recurrent_block_type', 'final_activation']}
    o = omod.UNetRecurrentOracle({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, **okw)
    H, W, NS = 260, 346, 8
    crop = op.CropParams(W, H, 3)
    vz = Voxelizer()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    m.reset_states()
    for f in range(2):
        x, y, t, p = gen_events(700 + f, 15000, W, H)
        st = torch.zeros((1, 3), dtype=torch.float64, device='cuda')
        g = vz.voxelize(d(x), d(y), d(t), d(p), d(np.array([0, 15000], dtype=np.int64)), 5, (H, W), stats=st)
        img = m(g.repeat(NS, 1, 1, 1), stats=st.repeat(NS, 1))['image'].cpu().numpy()
        v = op.normalize_event_tensor(ov.events_to_voxel(x, y, t, p, 5, (H, W))[None])
        want = crop.crop(o(torch.from_numpy(crop.pad(v))).numpy())
        for s in range(NS):
            np.testing.assert_allclose(img[s:s + 1], want, rtol=0, atol=IMG_ATOL, err_msg=f'frame {f} seq {s}')


def test_recurrent_drift_30_frames():
    """fp32 MFMA vs torch-CPU over a long recurrence: the 1e-4 per-pixel gate must hold on every frame, and the
    ConvLSTM states must not drift (error compounds through h/c, SURVEY section 7 'hard parts')."""
    from evreal_amd import model, synth, weights
    from oracle import model as omod
    kw = dict(weights.E2VID_KWARGS)
    sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=11)
    m = model.E2VIDRecurrent(kw); m.load_state_dict(sd)
    okw = {k: kw[k] for k in ['num_bins', 'base_num_channels', 'num_encoders', 'num_residual_blocks', 'kernel_size',
                              'norm', 'use_upsample_conv', 'recurrent_block_type', 'final_activation']}
    o = omod.UNetRecurrentOracle({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, **okw)
    F, H, W = 30, 64, 96
    vox = synth.sparse_voxels(123, F, 5, H, W, density=0.1)
    m.reset_states()
    worst = 0.0
    for f in range(F):
        got = m(torch.from_numpy(vox[f:f + 1]).cuda())['image'].cpu().numpy()
        want = o(torch.from_numpy(vox[f:f + 1])).numpy()
        worst = max(worst, float(np.abs(got - want).max()))
        assert worst < IMG_ATOL, (f, worst)
    for i in range(3):
        h = m.read_tensor(f'h{i}').cpu().numpy().reshape(o.states[i][0].shape)
        c = m.read_tensor(f'c{i}').cpu().numpy().reshape(o.states[i][1].shape)
        np.testing.assert_allclose(h, o.states[i][0].numpy(), rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(c, o.states[i][1].numpy(), rtol=2e-4, atol=5e-5)


def test_firenet_real_weights_40_frames_and_reset():
    """FireNet with the shipped checkpoint over 40 frames, then reset_states() must reproduce frame 0 exactly."""
    from evreal_amd import model, synth
    from oracle import model as omod
    from oracle.prepost import CropParams
    w = load_npz('firenet_weights.npz')
    sd = {k: w[k] for k in w.files}
    m = model.FireNet_legacy(unet_kwargs=dict(num_bins=5, recurrent_block_type='convgru', base_num_channels=16,
                                              num_residual_blocks=2, kernel_size=3, norm='none'))
    m.load_state_dict(sd)
    o = omod.FireNetLegacyOracle({k: torch.from_numpy(v) for k, v in sd.items()})
    F, H, W = 40, 90, 120
    crop = CropParams(W, H, 4)
    vox = synth.sparse_voxels(321, F, 5, H, W, density=0.08)
    m.reset_states()
    first = None
    for f in range(F):
        got = m(torch.from_numpy(vox[f:f + 1]).cuda())['image'].cpu().numpy()
        want = crop.crop(o(torch.from_numpy(crop.pad(vox[f:f + 1]))).numpy())
        assert float(np.abs(got - want).max()) < IMG_ATOL, f
        if f == 0:
            first = got
    m.reset_states()
    again = m(torch.from_numpy(vox[0:1]).cuda())['image'].cpu().numpy()
    assert np.array_equal(again, first)          # deterministic kernels + zeroed state


def test_spade_e2vid_golden():
    """SPADE-E2VID (Unet6, model/spade_e2v.py) against the reference class: pixel-shuffle decoders (written through the
    phase-major column groups), SPADE normalisation, the first-frame in-place rewrite of x[:, :3], 3-channel head."""
    from evreal_amd import model, synth, weights
    z = load_npz('spade_seq.npz')
    sd = weights.synth_state_dict(weights.spade_e2vid_schema(), seed=int(z['seed']))
    assert weights.state_dict_digest(sd) == str(z['weights_sha'])
    m = model.SpadeE2vid(); m.load_state_dict(sd)
    assert m.num_encoders == 3
    seed, F, B, H, W = [int(v) for v in z['voxel_args']]
    vox = synth.sparse_voxels(seed, F, B, H, W, density=0.15)
    assert sha(vox) == str(z['voxel_sha'])
    for n_seq in (1, 2):                     # every batch slot is its own sequence (own min/max on the first frame)
        m.reset_states()
        for f in range(F):
            x = torch.from_numpy(vox[f:f + 1]).cuda().repeat(n_seq, 1, 1, 1)
            img = m(x)['image'].cpu().numpy()
            for s in range(n_seq):
                np.testing.assert_allclose(img[s:s + 1], z['images'][f:f + 1], rtol=0, atol=IMG_ATOL, err_msg=f'frame {f} seq {s}')
        for i in range(4):
            want = z[f'h{i}_sub']
            h = m.read_tensor(f'h{i}').cpu().numpy().reshape(n_seq, -1, want.shape[2], want.shape[3])
            c = m.read_tensor(f'c{i}').cpu().numpy().reshape(n_seq, -1, want.shape[2], want.shape[3])
            np.testing.assert_allclose(h[:1, ::4], want, rtol=1e-4, atol=2e-5, err_msg=f'h{i}')
            np.testing.assert_allclose(c[:1, ::4], z[f'c{i}_sub'], rtol=1e-4, atol=2e-5, err_msg=f'c{i}')
    m.reset_states()                          # reset -> the first-frame path again, bit for bit
    a = m(torch.from_numpy(vox[0:1]).cuda())['image'].cpu().numpy()
    m.reset_states()
    b = m(torch.from_numpy(vox[0:1]).cuda())['image'].cpu().numpy()
    assert np.array_equal(a, b)


def test_spade_e2vid_through_the_method_registry(tmp_path):
    """eval.py:130-133: the 'SPADE-E2VID' checkpoint IS the state_dict; num_encoders = 3 for the cropper."""
    from evreal_amd import eval as ev, model, weights
    sd = weights.synth_state_dict(weights.spade_e2vid_schema(), seed=3)
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, tmp_path / 'spade.pth')
    m = ev.get_model_from_checkpoint_path('SPADE-E2VID', str(tmp_path / 'spade.pth'))
    assert isinstance(m, model.SpadeE2vid) and m.num_encoders == 3
    img = m(torch.zeros((1, 5, 60, 90), device='cuda'))['image']            # pads to 64x96, crops back
    assert img.shape == (1, 1, 60, 90) and bool(torch.isfinite(img).all())


@pytest.mark.parametrize('norm,tag', [(None, 'etnet'), ('BN', 'etnet_bn'), ('IN', 'etnet_in')])
def test_etnet_golden(norm, tag):
    """ET-Net (EITR, model/eitr/) against the reference class: ConvLSTM encoder, patch embeddings, sine positions, nine
    pre-norm encoder layers and six decoder layers (LayerNorm, 8-head attention, FFN), mean of the six token sets,
    bilinear-upsample decoders; for every `norm` the reference's ConvLayers accept (u_trans.py:16-52: BatchNorm, or
    running-statistics InstanceNorm, folded into head / encoders / decoders / prediction layer)."""
    from evreal_amd import model, synth, weights
    z = load_npz(f'{tag}_seq.npz')
    sd = weights.synth_state_dict(weights.etnet_schema(norm=norm), seed=int(z['seed']))
    assert weights.state_dict_digest(sd) == str(z['weights_sha'])
    m = model.EITR({'num_bins': 5, 'norm': norm}); m.load_state_dict(sd)
    seed, F, B, H, W = [int(v) for v in z['voxel_args']]
    vox = synth.sparse_voxels(seed, F, B, H, W, density=0.15)
    assert sha(vox) == str(z['voxel_sha'])
    for n_seq in (1, 2):
        m.reset_states()
        for f in range(F):
            x = torch.from_numpy(vox[f:f + 1]).cuda().repeat(n_seq, 1, 1, 1)
            img = m(x)['image'].cpu().numpy()
            for s in range(n_seq):
                np.testing.assert_allclose(img[s:s + 1], z['images'][f:f + 1], rtol=0, atol=IMG_ATOL, err_msg=f'frame {f} seq {s}')
        for i in range(3):
            want = z[f'h{i}_sub']
            h = m.read_tensor(f'h{i}').cpu().numpy().reshape(n_seq, -1, want.shape[2], want.shape[3])
            np.testing.assert_allclose(h[:1, ::4], want, rtol=1e-4, atol=2e-5, err_msg=f'h{i}')
    # eval.py:197 resets the states before EVERY sequence of a dataset; at an unchanged (n, H, W) the library only zeroes its
    # buffers -- the sine position table must survive that (it once sat among them and every later sequence lost the term)
    for rep in range(2):
        m.reset_states()
        for f in range(F):
            x = torch.from_numpy(vox[f:f + 1]).cuda().repeat(2, 1, 1, 1)
            img = m(x)['image'].cpu().numpy()
            np.testing.assert_allclose(img[:1], z['images'][f:f + 1], rtol=0, atol=IMG_ATOL, err_msg=f'after same-shape reset {rep}, frame {f}')


def test_large_activations_are_reported_not_silent():
    """The packed activation formats have a finite exact range (csrc/packed.h sat_note): PACKED's fp8 pieces (EVR_ARITH=mx) saturate
    from |x| ~ 256 on (the value then keeps only its f16 half, 2^-12 relative), H2 (EVR_ARITH=h3) clamps at +-4094, P6 (the default for
    this layout) scales every 16-channel group by its own maximum and only clamps beyond the half-precision range.  Inputs 30x the
    usual magnitude stay inside and pass the gate with a zero counter; at 300x and beyond every frame EITHER still passes
    1e-4 OR evr_model_saturation reports the excursion (so the degradation is never silent); results stay finite."""
    from evreal_amd import model, weights
    from oracle import model as omod
    kw = dict(weights.E2VID_KWARGS)
    sd = weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=5)
    m = model.E2VIDRecurrent(kw); m.load_state_dict(sd)
    okeys = ['num_bins', 'base_num_channels', 'num_encoders', 'num_residual_blocks', 'kernel_size', 'norm', 'use_upsample_conv',
             'recurrent_block_type', 'final_activation']
    o = omod.UNetRecurrentOracle({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, **{k: kw[k] for k in okeys})
    rng = np.random.default_rng(0)
    exact = bool(os.environ.get('EVR_FP32')) or os.environ.get('EVR_ARITH') == 'fp32'
    for scale in (30.0, 300.0, 1e5):
        m.reset_states(); o.reset_states()
        m.saturation(clear=True)
        worst = 0.0
        for f in range(3):
            v = np.zeros((1, 5, 64, 96), np.float32)
            mk = rng.random(v.shape) < 0.2
            v[mk] = rng.normal(0, 1.5, mk.sum()) * scale
            img = m(torch.from_numpy(v).cuda())['image'].cpu().numpy()
            with torch.no_grad():
                want = o(torch.from_numpy(v)).numpy()
            assert np.isfinite(img).all()
            worst = max(worst, float(np.abs(img - want).max()))
        runs, layer = m.saturation()
        loose = {30.0: 1e-4, 300.0: 1e-3, 1e5: 5e-2}[scale]      # fp32 itself: summation-order differences grow with the magnitude
        if exact:
            assert runs == 0 and worst < loose, (scale, runs, layer, worst)
        elif scale == 30.0:
            assert runs == 0 and worst < 1e-4, (scale, runs, layer, worst)
        else:
            assert worst < 1e-4 or runs > 0, (scale, runs, layer, worst)       # never a silent degradation
            if runs:
                assert layer != ''
            if ARITH in ('mx', 'mx6') and scale < 1e4:
                assert worst < loose, (scale, worst)   # PACKED beyond its range: the f16 half alone still carries 2^-12 (P6 scales per group)
            if scale >= 1e4:
                assert runs > 0, scale                 # inputs beyond the half-precision range itself (+-65504) are clamped: reported


def test_exact_twin_is_the_reference_arithmetic_without_a_range_limit():
    """_HipModel.exact_twin(): the same weights on the library's exact-fp32 kernels, in the same process as the default arithmetic
    (evr_model_desc.reserved[2]).  On a network whose enc1.conv output is 65536 times larger (weights.rescale_encoder_conv: the same
    function in fp32) the default arithmetic reports saturation, the twin reports none and matches the oracle to 1e-4 -- in fact to
    the level the unscaled network reaches -- frame after frame through the recurrence."""
    from evreal_amd import model, synth, weights
    from oracle import model as omod, prepost as op
    kw = dict(weights.E2VID_KWARGS)
    sd = weights.rescale_encoder_conv(weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=21), enc=1, K=65536.0)
    m = model.E2VIDRecurrent(kw); m.load_state_dict(sd)
    twin = m.exact_twin()
    assert twin.arith == 'fp32' and twin.exact_twin() is twin and m.exact_twin() is twin and (twin is m) == (m.arith == 'fp32')
    okeys = ['num_bins', 'base_num_channels', 'num_encoders', 'num_residual_blocks', 'kernel_size', 'norm', 'use_upsample_conv',
             'recurrent_block_type', 'final_activation']
    o = omod.UNetRecurrentOracle({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, **{k: kw[k] for k in okeys})
    vox = synth.sparse_voxels(3, 4, 5, 50, 70, density=0.1)
    crop = op.CropParams(70, 50, 3)
    m.reset_states(); twin.reset_states(); o.reset_states()
    m.saturation(clear=True)
    worst_twin = worst_m = 0.0
    for f in range(4):
        x = torch.from_numpy(vox[f:f + 1]).cuda()
        with torch.no_grad():
            want = crop.crop(o(torch.from_numpy(crop.pad(vox[f:f + 1]))).numpy())
        worst_twin = max(worst_twin, float(np.abs(twin(x)['image'].cpu().numpy() - want).max()))
        if twin is not m:        # (EVR_FP32=1: the model is its own twin -- one step per frame)
            worst_m = max(worst_m, float(np.abs(m(x)['image'].cpu().numpy() - want).max()))
    assert worst_twin < 1e-5, worst_twin
    assert twin.saturation()[0] == 0
    if m.arith in ('h3', 'mx'):      # fixed-range formats: the excursion is reported (and evaluate() re-runs on the twin)
        runs, layer = m.saturation()
        assert runs > 0 and layer == 'enc1.conv', (runs, layer, worst_m)
    elif m.arith == 'mx6':           # P6 scales every 16-channel group by its own maximum: 2e4 is inside the half-precision range
        runs, layer = m.saturation()
        assert runs > 0 or worst_m < 1e-4, (runs, layer, worst_m)


def test_arithmetic_is_narrowed_per_layout():
    """evr_model_arith: the default is the fp32-grade three-f16-product arithmetic for every layout; the opt-in f16 + MX-fp6 arithmetic (EVR_ARITH=mx6) needs a layout whose packed tensors are written as whole 16-channel groups
    (ConvLSTM UNets with transposed or upsample-conv decoders, BN / no norm, the 5-bin k5 32-channel head); every other layout runs
    f16 + MX-fp8 in the same process; explicit modes (EVR_ARITH=mx|h3, EVR_FP32=1) apply to all."""
    from evreal_amd import model, weights
    env = ARITH
    if os.environ.get('EVR_GROUP_STORE', '1') == '0' and env == 'mx6':
        env = 'mx'          # (the 4-channel-piece A/B switch: P6 has no such writer)
    cases = [('e2vid_bn', True), ('e2vid_plus', True), ('e2vid_hyper', False), ('e2vid_gru_tiny', False), ('e2vid_in', False)]
    for tag, eligible in cases:
        z = load_npz(f'{tag}_seq.npz')
        kw = json.loads(bytes(z['kwargs']).decode())
        fixed = {k[6:]: z[k] for k in z.files if k.startswith('fixed.')}
        m = model.E2VIDRecurrent(kw)
        m.load_state_dict(weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=int(z['seed']), fixed=fixed))
        want = env if (env != 'mx6' or eligible) else 'mx'
        assert m.arith == want, (tag, m.arith, want)

"""GPU parity: event-tensor normalization, percentile post-normalization, MSE/SSIM (C ABI) vs oracle/goldens."""
import numpy as np
import pytest
import torch

from conftest import load_npz
from golden_inputs import gen_events

pytestmark = pytest.mark.gpu


def test_normalize_event_tensor_goldens():
    from evreal_amd.prepost import normalize_event_tensor
    from oracle import voxel as ov, prepost as op
    z = load_npz('normalize.npz')
    for k in ['zeros', 'single', 'dense']:
        v = torch.from_numpy(z[k + '.in'].copy()).cuda()
        out = normalize_event_tensor(v).cpu().numpy()
        np.testing.assert_allclose(out, z[k + '.out'], rtol=2e-6, atol=2e-6, err_msg=k)
    x, y, t, p = gen_events(int(z['vox15k.seed']), 15000, 346, 260)
    v = ov.events_to_voxel(x, y, t, p, 5, (260, 346))[None]
    out = normalize_event_tensor(torch.from_numpy(v.copy()).cuda()).cpu().numpy()
    np.testing.assert_allclose(out, z['vox15k.out'], rtol=2e-6, atol=2e-6)
    assert np.array_equal(out == 0, z['vox15k.out'] == 0)
    # batched + stats handed over from the tensorizer give the same numbers
    from evreal_amd.voxel import Voxelizer
    vz = Voxelizer()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    st = torch.zeros((1, 3), dtype=torch.float64, device='cuda')
    g = vz.voxelize(d(x), d(y), d(t), d(p), d(np.array([0, 15000], dtype=np.int64)), 5, (260, 346), stats=st)
    out2 = normalize_event_tensor(g, stats=st).cpu().numpy()
    np.testing.assert_allclose(out2, z['vox15k.out'], rtol=2e-6, atol=2e-6)


def test_post_process_normalization_goldens():
    from evreal_amd.prepost import post_process_normalization
    z = load_npz('robust_norm.npz')
    for k in ['unit', 'wide', 'small', 'ties']:
        for norm in ['robust', 'standard']:
            img = torch.from_numpy(z[k + '.in'].copy()).cuda()
            out = post_process_normalization(img, norm).cpu().numpy()
            assert np.array_equal(out, z[f'{k}.{norm}'], equal_nan=True), (k, norm)   # bit exact
        img = torch.from_numpy(z[k + '.in'].copy()).cuda()
        out = post_process_normalization(img, 'exprobust').cpu().numpy()
        np.testing.assert_allclose(out, z[f'{k}.exprobust'], rtol=2e-6, atol=2e-6)  # expf vs numpy's SIMD exp


def test_post_process_batched_matches_single():
    from evreal_amd.prepost import post_process_normalization
    from oracle import prepost as op
    rng = np.random.default_rng(2)
    a = rng.random((6, 260, 346)).astype(np.float32)
    out = post_process_normalization(torch.from_numpy(a.copy()).cuda(), 'robust').cpu().numpy()
    for i in range(6):
        assert np.array_equal(out[i], op.post_process_normalization(a[i].copy(), 'robust'))


@pytest.mark.parametrize('shape', [(260, 346), (180, 240), (64, 80), (33, 47)])
def test_mse_ssim_vs_oracle(shape):
    from evreal_amd.prepost import Metrics
    from oracle import metrics as om
    rng = np.random.default_rng(shape[0])
    H, W = shape
    n = 5
    yy, xx = np.mgrid[0:H, 0:W]
    ref = np.stack([0.5 + 0.4 * np.sin(xx / (7.0 + i) + i) * np.cos(yy / (9.0 + i)) for i in range(n)]).astype(np.float32)
    img = (ref + 0.15 * rng.standard_normal(ref.shape)).astype(np.float32)   # exceeds [0,1] -> exercises the clip
    img[0] = ref[0]
    m = Metrics()
    out = m(torch.from_numpy(img).cuda(), torch.from_numpy(ref).cuda()).cpu().numpy()
    for i in range(n):
        a, b = om.clip01(img[i]), om.clip01(ref[i])
        assert abs(out[i, 0] - om.mse(a, b)) <= 1e-12 + 1e-9 * om.mse(a, b)
        assert abs(out[i, 1] - om.ssim(a, b)) <= 2e-6, (i, out[i, 1], om.ssim(a, b))
    assert out[0, 0] == 0.0 and abs(out[0, 1] - 1.0) < 1e-6


@pytest.mark.parametrize('shape', [(260, 346), (180, 240), (96, 128)])
def test_lpips_vs_oracle(shape):
    """LPIPS structure (AlexNet v0.1) with deterministic synthetic weights: HIP vs the torch restatement.
    Parity vs pyiqa itself is unpinned (weights unobtainable offline)."""
    from evreal_amd import weights
    from evreal_amd.lpips import LPIPS
    from oracle import lpips as ol
    sd = weights.synth_lpips_state_dict(seed=3)
    H, W = shape
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:H, 0:W]
    ref = np.stack([0.5 + 0.4 * np.sin(xx / (7.0 + i)) * np.cos(yy / (9.0 + i)) for i in range(3)]).astype(np.float32)
    img = np.clip(ref + 0.1 * rng.standard_normal(ref.shape), 0, 1).astype(np.float32)
    img[0] = ref[0]
    m = LPIPS(sd)
    got = m(torch.from_numpy(img).cuda(), torch.from_numpy(ref).cuda()).cpu().numpy()
    want = ol.lpips(sd, img, ref)
    assert got[0] == 0.0
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-7)


def test_lpips_handle_serves_changing_pair_counts():
    """The drop-in's chunks change the number of scored frames at every sequence boundary.  One handle keeps buffers for the largest
    count seen and a small cache of per-count plans (lpips.hip, round 6): every count -- smaller than the capacity, growing it,
    evicting the oldest of the eight cached plans, returning to an evicted one -- must give exactly what a fresh handle gives."""
    from evreal_amd import weights
    from evreal_amd.lpips import LPIPS
    from oracle import lpips as ol
    sd = weights.synth_lpips_state_dict(seed=3)
    H, W = 97, 131
    rng = np.random.default_rng(5)
    ref = rng.random((12, H, W), dtype=np.float32)
    img = np.clip(ref + 0.1 * rng.standard_normal(ref.shape), 0, 1).astype(np.float32)
    d_img, d_ref = torch.from_numpy(img).cuda(), torch.from_numpy(ref).cuda()
    want = LPIPS(sd)(d_img, d_ref).cpu().numpy()                      # a fresh handle, 12 pairs at once
    np.testing.assert_allclose(want[:2], ol.lpips(sd, img[:2], ref[:2]), rtol=2e-4, atol=1e-7)
    fresh = {}
    m = LPIPS(sd)
    for n in [3, 5, 2, 3, 12, 1, 4, 6, 7, 8, 9, 10, 11, 3, 12, 5, 1]:      # 12 distinct counts through 8 plan slots
        o = int(rng.integers(0, 12 - n + 1))
        got = m(d_img[o:o + n], d_ref[o:o + n]).cpu().numpy()
        if n not in fresh:
            fresh[n] = LPIPS(sd)
        np.testing.assert_array_equal(got, fresh[n](d_img[o:o + n], d_ref[o:o + n]).cpu().numpy(), err_msg=f'n={n}')
        np.testing.assert_allclose(got, want[o:o + n], rtol=1e-6, atol=1e-9, err_msg=f'n={n}')
    m2 = LPIPS(sd)      # a different image size on a used handle re-allocates
    a = m2(d_img[:2], d_ref[:2]).cpu().numpy()
    b = m2(d_img[:2, :64, :80].contiguous(), d_ref[:2, :64, :80].contiguous()).cpu().numpy()
    c = m2(d_img[:2], d_ref[:2]).cpu().numpy()
    np.testing.assert_array_equal(a, c)
    np.testing.assert_allclose(b, ol.lpips(sd, img[:2, :64, :80], ref[:2, :64, :80]), rtol=2e-4, atol=1e-7)


def test_lpips_with_user_supplied_weights():
    """When the real AlexNet-v0.1 LPIPS weights are available (EVREAL_LPIPS_WEIGHTS = a torch.save'd pyiqa / lpips state_dict, the
    file evreal_amd.eval picks up), the HIP path and the oracle are compared on THEM -- trained filters instead of the synthetic
    statistics.  Offline the weights cannot be obtained: the test then skips and says so."""
    import os
    from evreal_amd.eval_metrics import LPIPS_WEIGHTS_ENV
    path = os.environ.get(LPIPS_WEIGHTS_ENV, os.path.join('pretrained', 'lpips_alex.pth'))
    if not os.path.exists(path):
        pytest.skip(f'no LPIPS weights at ${LPIPS_WEIGHTS_ENV} / pretrained/lpips_alex.pth (pyiqa downloads them; there is no network here)')
    from evreal_amd.lpips import LPIPS
    from oracle import lpips as ol
    sd = torch.load(path, map_location='cpu', weights_only=False)
    sd = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, 'detach') else v, dtype=np.float32) for k, v in sd.items()}
    H, W = 260, 346
    rng = np.random.default_rng(17)
    yy, xx = np.mgrid[0:H, 0:W]
    ref = np.stack([0.5 + 0.4 * np.sin(xx / (7.0 + i)) * np.cos(yy / (9.0 + i)) for i in range(3)]).astype(np.float32)
    img = np.clip(ref + 0.1 * rng.standard_normal(ref.shape), 0, 1).astype(np.float32)
    m = LPIPS(sd)
    got = m(torch.from_numpy(img).cuda(), torch.from_numpy(ref).cuda()).cpu().numpy()
    want = ol.lpips(sd, img, ref)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-7)


def test_metrics_kernel_against_the_published_definition():
    """evr_metrics vs the Wang et al. known answers (tests/golden/metrics_published.npz, float64, explicit window loops)."""
    import json
    from evreal_amd.prepost import Metrics
    z = load_npz('metrics_published.npz')
    met = Metrics()
    for m in json.loads(bytes(z['meta']).decode()):
        x, y = z[m['name'] + '.x'], z[m['name'] + '.y']
        out = met(torch.from_numpy(x[None]).cuda(), torch.from_numpy(y[None]).cuda(), clip=False).cpu().numpy()[0]
        assert abs(out[0] - m['mse']) < 1e-9 and abs(out[1] - m['ssim']) < 5e-6, (m['name'], out)


def test_device_split_codec_matches_host_bit_for_bit():
    """The PACKED codec as the kernels' epilogues run it (v_cvt_f16_f32, v_cvt_pk_fp8_f32 after the clamps) against the host
    codec, which tests/test_split_codec.py pins to numpy / torch dtypes: f16 RNE incl. subnormals, OCP e4m3 RNE, saturation."""
    import ctypes
    from evreal_amd import lib as _lib
    L = _lib.load()
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.standard_normal(1 << 16) * 10.0 ** rng.integers(-9, 4, 1 << 16),
                        [0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1e-30, 0.5, 255.0, 447.9, 448.0, 449.0, 65504.0, 65519.0, 7e4, 2.0 ** -14]]).astype(np.float32)
    x = np.resize(x, (x.size // 16) * 16)
    want = np.empty_like(x)
    assert L.evr_split_pack(x.ctypes.data_as(ctypes.c_void_p), want.ctypes.data_as(ctypes.c_void_p), x.size) == 0
    d = torch.from_numpy(x).cuda()
    _lib.check(L.evr_split_pack_device(_lib.ptr(d), _lib.ptr(d), x.size, _lib.stream_ptr()), 'evr_split_pack_device')
    got = d.cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))


def test_h2_codec_device_equals_host_bit_for_bit():
    """The H2 codec of the fp32-grade mode as the kernels run it (two v_cvt_f16_f32 per value after the clamp, subnormal halves
    kept) against the host codec that tests/test_split_codec.py pins to numpy's float16."""
    import ctypes
    from evreal_amd import lib as _lib
    L = _lib.load()
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.standard_normal(1 << 16) * 10.0 ** rng.integers(-9, 4, 1 << 16),
                        [0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1e-30, 0.5, 255.0, 4093.9, 4094.0, 4095.0, 65504.0, 7e4, 2.0 ** -14, 2.0 ** -28]]).astype(np.float32)
    x = np.resize(x, (x.size // 16) * 16)
    want = np.empty_like(x)
    assert L.evr_h2_pack(x.ctypes.data_as(ctypes.c_void_p), want.ctypes.data_as(ctypes.c_void_p), x.size) == 0
    d = torch.from_numpy(x).cuda()
    _lib.check(L.evr_h2_pack_device(_lib.ptr(d), _lib.ptr(d), x.size, _lib.stream_ptr()), 'evr_h2_pack_device')
    got = d.cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))


def test_p6_codec_device_equals_host_bit_for_bit():
    """The P6 codec of the f16 + MX-fp6 mode as the kernels run it (v_cvt_f16_f32 + ONE v_cvt_scalef32_2xpk16_fp6_f32 per group with
    the group's own scale) against the host codec that tests/test_split_codec.py pins to numpy."""
    import ctypes
    from evreal_amd import lib as _lib
    L = _lib.load()
    rng = np.random.default_rng(9)
    x = np.concatenate([rng.standard_normal(1 << 16) * 10.0 ** rng.integers(-9, 4, 1 << 16), rng.standard_normal(1 << 15) * 2.0,
                        [0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1e-30, 0.5, 255.0, 7.5, 7.75, 8.0, 65504.0, 7e4, 2.0 ** -14, 2.0 ** -28],
                        np.zeros(16)]).astype(np.float32)
    x = np.resize(x, (x.size // 16) * 16)
    want = np.empty_like(x)
    assert L.evr_p6_pack(x.ctypes.data_as(ctypes.c_void_p), want.ctypes.data_as(ctypes.c_void_p), x.size) == 0
    d = torch.from_numpy(x).cuda()
    _lib.check(L.evr_p6_pack_device(_lib.ptr(d), _lib.ptr(d), x.size, _lib.stream_ptr()), 'evr_p6_pack_device')
    got = d.cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))

"""Self-pinning tests for the six SURVEY rows that third-party arithmetic caps at "parity unpinned" (a22 / 8f-3 ColorNet's Lab merge,
a27 MSE, a28 SSIM, a29 LPIPS, 8f-4 hist-eq): scikit-image, OpenCV and pyiqa are not in the reference tree and cannot be installed in
the build image (no package index).  Every test here compares `oracle/` (CPU, no marker) or the HIP kernels (`-m gpu`) with

  * the REAL package, called exactly as the reference calls it (tests/thirdparty_refs.py cites the lines) -- behind
    `pytest.importorskip`, so the test skips here and starts pinning the moment the package is importable; and
  * the frozen outputs of tests/golden/thirdparty_*.{json,npz} (written by tests/golden/make_thirdparty_golden.py on a box that has
    the packages) -- skipping while those files do not exist.

ROUND 6: this image turned out to carry an Anaconda python3.9 with scikit-image 0.18.3 beside the system interpreter.  The fixtures
thirdparty_metrics.json (MSE, SSIM) and thirdparty_histeq.npz (hist-eq 'global' and 'local') were written by
`/opt/conda/bin/python3.9 tests/golden/make_thirdparty_golden.py` and are committed: rows a27, a28 and two of 8f-4's three modes
are pinned to the real package on every box (oracle: MSE bit for bit, SSIM 5e-8, global 6e-8, local exact), and
`/opt/conda/bin/python3.9 -m pytest tests/test_thirdparty_pins.py -m "not gpu"` runs the live comparisons here.  OpenCV (CLAHE, the
colour merge) and pyiqa (LPIPS) exist nowhere in the image: those tests still skip.

One command on a box with the packages turns the remaining rows green:
    python tests/golden/make_thirdparty_golden.py && python -m pytest tests/test_thirdparty_pins.py -q [-m gpu]

Tolerances are the ones the product's own gates use: MSE 1e-9 absolute, SSIM 5e-6 (the HIP kernel's distance from the oracle), LPIPS
2e-4 relative, hist-eq and the colour merge exact in uint8 except where noted in the test.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import thirdparty_refs as tp
from conftest import GOLDEN


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _fixture(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f'{name} not generated yet (tests/golden/make_thirdparty_golden.py on a box with the third-party packages)')
    return path


def _gpu_metrics(pairs):
    import torch
    from evreal_amd.prepost import Metrics
    met = Metrics()
    out = {}
    for name, img, ref in pairs:
        r = met(torch.from_numpy(img[None]).cuda(), torch.from_numpy(ref[None]).cuda(), clip=False).cpu().numpy()[0]
        out[name] = (float(r[0]), float(r[1]))
    return out


# ---- a27 / a28: MSE, SSIM -----------------------------------------------------------------------------------------------------------
def test_oracle_mse_ssim_vs_skimage():
    pytest.importorskip('skimage.metrics')
    from oracle import metrics as om
    for name, img, ref in tp.image_pairs():
        assert abs(om.mse(img, ref) - tp.skimage_mse(img, ref)) < 1e-9, name
        assert abs(om.ssim(img, ref) - tp.skimage_ssim(img, ref)) < 2e-6, name


def test_oracle_mse_ssim_vs_frozen_skimage():
    fx = json.load(open(_fixture('thirdparty_metrics.json')))
    from oracle import metrics as om
    pairs = {n: (i, r) for n, i, r in tp.image_pairs()}
    for row in fx['rows']:
        img, ref = pairs[row['name']]
        assert _sha(img) == row['img_sha'] and _sha(ref) == row['ref_sha'], 'input generator drifted'
        assert abs(om.mse(img, ref) - row['mse']) < 1e-9 and abs(om.ssim(img, ref) - row['ssim']) < 2e-6, (row['name'], fx['package'])


@pytest.mark.gpu
def test_hip_mse_ssim_vs_skimage():
    pytest.importorskip('skimage.metrics')
    got = _gpu_metrics(tp.image_pairs())
    for name, img, ref in tp.image_pairs():
        assert abs(got[name][0] - tp.skimage_mse(img, ref)) < 1e-9 and abs(got[name][1] - tp.skimage_ssim(img, ref)) < 5e-6, name


@pytest.mark.gpu
def test_hip_mse_ssim_vs_frozen_skimage():
    fx = json.load(open(_fixture('thirdparty_metrics.json')))
    got = _gpu_metrics(tp.image_pairs())
    for row in fx['rows']:
        assert abs(got[row['name']][0] - row['mse']) < 1e-9 and abs(got[row['name']][1] - row['ssim']) < 5e-6, (row['name'], fx['package'])


# ---- 8f-4: histogram equalisation ---------------------------------------------------------------------------------------------------
def _histeq_cases(modes):
    for name, img in tp.histeq_images():
        for mode in modes:
            if mode == 'local' and img.size > 72 * 88:
                continue
            yield name, img, mode


def _u8(a):
    return np.rint(np.asarray(a, np.float64) * 255).astype(np.int64)


@pytest.mark.parametrize('mode,needs', [('global', ['skimage']), ('local', ['skimage']), ('clahe', ['skimage', 'cv2'])])
def test_oracle_histeq_vs_packages(mode, needs):
    for m in needs:
        pytest.importorskip(m)
    from oracle import histeq as oh
    for name, img, md in _histeq_cases([mode]):
        want, got = tp.thirdparty_histeq(img, md), oh.histogram_equalization(img, md)
        if mode == 'global':      # float output: the cdf interpolation in float64, rounded to float32 at the end
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-7, err_msg=f'{name} {md}')
        else:                     # uint8 pipelines: exact
            assert np.array_equal(_u8(got), _u8(want)), (name, md, int(np.abs(_u8(got) - _u8(want)).max()))


def test_oracle_histeq_vs_frozen_packages():
    z = np.load(_fixture('thirdparty_histeq.npz'))
    from oracle import histeq as oh
    imgs = dict(tp.histeq_images())
    for m in json.loads(bytes(z['meta']).decode()):
        img = imgs[m['name']]
        assert _sha(img) == m['img_sha'], 'input generator drifted'
        got, want = oh.histogram_equalization(img, m['mode']), z[f"{m['name']}.{m['mode']}"]
        if m['mode'] == 'global':
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-7)
        else:
            assert np.array_equal(_u8(got), _u8(want)), m


def _hip_histeq(img, mode):
    import torch
    from evreal_amd.prepost import histogram_equalization
    t = torch.from_numpy(img[None].copy()).cuda()
    return histogram_equalization(t, mode).cpu().numpy()[0]


@pytest.mark.gpu
@pytest.mark.parametrize('mode,needs', [('global', ['skimage']), ('local', ['skimage']), ('clahe', ['skimage', 'cv2'])])
def test_hip_histeq_vs_packages(mode, needs):
    for m in needs:
        pytest.importorskip(m)
    for name, img, md in _histeq_cases([mode]):
        want, got = tp.thirdparty_histeq(img, md), _hip_histeq(img, md)
        if mode == 'global':
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-7, err_msg=f'{name} {md}')
        else:
            assert np.abs(_u8(got) - _u8(want)).max() <= (1 if mode == 'clahe' else 0), (name, md)      # (CLAHE: fp32-level ties of the bilinear blend, as tests/test_gpu_tracker.py allows)


@pytest.mark.gpu
def test_hip_histeq_vs_frozen_packages():
    z = np.load(_fixture('thirdparty_histeq.npz'))
    imgs = dict(tp.histeq_images())
    for m in json.loads(bytes(z['meta']).decode()):
        got, want = _hip_histeq(imgs[m['name']], m['mode']), z[f"{m['name']}.{m['mode']}"]
        if m['mode'] == 'global':
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-7)
        else:
            assert np.abs(_u8(got) - _u8(want)).max() <= (1 if m['mode'] == 'clahe' else 0), m


# ---- a22 / 8f-3: ColorNet's colour merge --------------------------------------------------------------------------------------------
# oracle/color.py restates OpenCV's Lab conversion in floating point; OpenCV's 8-bit path uses fixed-point tables, so single-LSB
# differences on a small share of the pixels are expected and everything beyond that is a bug.  The bounds below are the acceptance
# criterion, written before the first run against a real cv2: <= 1 code on every pixel and channel, identical on >= 99 %.
# (the HIP kernel is a second floating-point restatement -- v_exp / v_log power functions -- held to the oracle at <= 2 codes and
# >= 98 % identical by tests/test_gpu_color.py; it gets the same room against OpenCV)
def _merge_close(got, want, what, max_codes=1, same=0.99):
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= max_codes, (what, int(d.max()))
    assert (d == 0).mean() >= same, (what, float((d == 0).mean()))


def test_oracle_color_merge_vs_cv2():
    pytest.importorskip('cv2')
    from oracle import color as oc
    for name, planes, gray in tp.color_inputs():
        _merge_close(oc.merge(planes, gray), tp.cv2_color_merge(planes, gray), name)


def test_oracle_color_merge_vs_frozen_cv2():
    z = np.load(_fixture('thirdparty_color.npz'))
    from oracle import color as oc
    ins = {n: (p, g) for n, p, g in tp.color_inputs()}
    for m in json.loads(bytes(z['meta']).decode()):
        planes, gray = ins[m['name']]
        assert _sha(planes) == m['planes_sha'] and _sha(gray) == m['gray_sha'], 'input generator drifted'
        _merge_close(oc.merge(planes, gray), z[m['name']], (m['name'], m['package']))


def _hip_merge(planes, gray):
    import torch
    from evreal_amd import lib as _lib
    lib = _lib.load()
    H, W = gray.shape
    p = torch.from_numpy(planes[None].copy()).cuda(); g = torch.from_numpy(gray[None, None].copy()).cuda()
    bgr = torch.empty((1, H, W, 3), dtype=torch.uint8, device='cuda')
    _lib.check(lib.evr_color_merge(_lib.ptr(p), _lib.ptr(g), 1, H, W, _lib.ptr(bgr), _lib.stream_ptr()), 'evr_color_merge')
    return bgr.cpu().numpy()[0]


@pytest.mark.gpu
def test_hip_color_merge_vs_cv2():
    pytest.importorskip('cv2')
    for name, planes, gray in tp.color_inputs():
        _merge_close(_hip_merge(planes, gray), tp.cv2_color_merge(planes, gray), name, 2, 0.98)


@pytest.mark.gpu
def test_hip_color_merge_vs_frozen_cv2():
    z = np.load(_fixture('thirdparty_color.npz'))
    ins = {n: (p, g) for n, p, g in tp.color_inputs()}
    for m in json.loads(bytes(z['meta']).decode()):
        _merge_close(_hip_merge(*ins[m['name']]), z[m['name']], (m['name'], m['package']), 2, 0.98)


# ---- a29: LPIPS ---------------------------------------------------------------------------------------------------------------------
def _pyiqa_or_skip():
    pytest.importorskip('pyiqa')
    try:
        return tp.pyiqa_lpips()
    except Exception as e:      # pyiqa downloads its weights on first use: no network, no weights
        pytest.skip(f'pyiqa is importable but its LPIPS weights are not on disk: {e}')


def test_oracle_lpips_vs_pyiqa():
    run, sd = _pyiqa_or_skip()
    from oracle import lpips as ol
    img, ref = tp.lpips_pairs()
    np.testing.assert_allclose(ol.lpips(sd, img, ref), run(img, ref), rtol=2e-5, atol=1e-7)


@pytest.mark.gpu
def test_hip_lpips_vs_pyiqa():
    run, sd = _pyiqa_or_skip()
    import torch
    from evreal_amd.lpips import LPIPS
    img, ref = tp.lpips_pairs()
    got = LPIPS(sd)(torch.from_numpy(img).cuda(), torch.from_numpy(ref).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, run(img, ref), rtol=2e-4, atol=1e-7)


@pytest.mark.gpu
def test_hip_lpips_vs_frozen_pyiqa():
    """Needs the frozen scores AND the weights they were computed with (EVREAL_LPIPS_WEIGHTS / pretrained/lpips_alex.pth: 10 MB, not a
    fixture); the digest in the fixture says whether they are the same weights."""
    fx = json.load(open(_fixture('thirdparty_lpips.json')))
    import torch
    from evreal_amd.eval_metrics import LPIPS_WEIGHTS_ENV
    from evreal_amd.lpips import LPIPS
    path = os.environ.get(LPIPS_WEIGHTS_ENV, os.path.join('pretrained', 'lpips_alex.pth'))
    if not os.path.exists(path):
        pytest.skip(f'no LPIPS weights at ${LPIPS_WEIGHTS_ENV} / pretrained/lpips_alex.pth')
    sd = torch.load(path, map_location='cpu', weights_only=False)
    sd = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, 'detach') else v, dtype=np.float32) for k, v in sd.items()}
    if _sha(np.concatenate([np.asarray(sd[k]).ravel() for k in sorted(sd)])) != fx['weights_sha']:
        pytest.skip('the weights on disk are not the ones the fixture was computed with')
    img, ref = tp.lpips_pairs()
    got = LPIPS(sd)(torch.from_numpy(img).cuda(), torch.from_numpy(ref).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, np.asarray(fx['scores']), rtol=2e-4, atol=1e-7)


# ---- the helpers above must work on the day the packages appear: run them against the oracle now -------------------------------------
@pytest.mark.gpu
def test_pin_helpers_run_against_the_oracle():
    from oracle import color as oc, histeq as oh, metrics as om
    got = _gpu_metrics(tp.image_pairs())
    for name, img, ref in tp.image_pairs():
        assert abs(got[name][0] - om.mse(img, ref)) < 1e-9 and abs(got[name][1] - om.ssim(img, ref)) < 5e-6, name
    for name, img, mode in _histeq_cases(['global', 'local', 'clahe']):
        g, w = _hip_histeq(img, mode), oh.histogram_equalization(img, mode)
        if mode == 'global':
            np.testing.assert_allclose(g, w, rtol=0, atol=2e-7, err_msg=name)
        else:
            assert np.abs(_u8(g) - _u8(w)).max() <= (1 if mode == 'clahe' else 0), (name, mode)
    for name, planes, gray in tp.color_inputs():
        _merge_close(_hip_merge(planes, gray), oc.merge(planes, gray), name, 2, 0.98)

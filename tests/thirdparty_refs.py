"""The reference's THIRD-PARTY arithmetic, called exactly as the reference calls it -- shared by tests/test_thirdparty_pins.py and
tests/golden/make_thirdparty_golden.py.

SURVEY 8c: MSE / SSIM live in scikit-image, LPIPS in pyiqa, the hist-eq modes in scikit-image + OpenCV, ColorNet's Lab merge in
OpenCV.  None of them is in the reference tree or in this image (no package index), so oracle/metrics.py, oracle/histeq.py,
oracle/color.py and oracle/lpips.py restate the published algorithms and are "parity unpinned".  Everything here imports the real
package lazily: on a box that has it, the tests of test_thirdparty_pins.py stop skipping and pin the oracle AND the HIP kernels to it;
`python tests/golden/make_thirdparty_golden.py` additionally freezes the packages' outputs into tests/golden/thirdparty_*.npz so that
the pin travels to boxes without them.  TEST INFRASTRUCTURE ONLY (nothing in evreal_amd/ imports this).
"""
import importlib
import os

import numpy as np


def have(*mods):
    for m in mods:
        try:
            importlib.import_module(m)
        except Exception:
            return False
    return True


# ---- inputs (seeded; the fixtures store their digests, not the arrays) -------------------------------------------------------------
def image_pairs():
    """[(name, img, ref)] float32 in [0, 1]: what the tracker hands to the metrics after its clip (utils/eval_metrics.py:253-255)."""
    out = []
    for name, (H, W), seed in [('davis346', (260, 346), 1), ('davis240', (180, 240), 2), ('vga', (480, 640), 3), ('odd', (97, 131), 4)]:
        rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:H, 0:W]
        ref = (0.5 + 0.35 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + 0.1 * np.sin((xx + 2 * yy) / 31.0)).astype(np.float32)
        img = np.clip(ref + 0.08 * rng.standard_normal(ref.shape), 0, 1).astype(np.float32)
        out.append((name, img, np.clip(ref, 0, 1).astype(np.float32)))
    rng = np.random.default_rng(9)
    out.append(('noise', rng.random((260, 346), dtype=np.float32), rng.random((260, 346), dtype=np.float32)))
    flat = np.full((64, 80), 0.25, np.float32)
    out.append(('constant', flat, flat.copy()))
    return out


def histeq_images():
    out = []
    for name, (H, W), seed in [('davis346', (260, 346), 11), ('small', (72, 88), 12)]:
        rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.clip(0.3 + 0.25 * np.sin(xx / 17.0) * np.cos(yy / 11.0) + 0.05 * rng.standard_normal((H, W)), 0, 1).astype(np.float32)
        out.append((name, img))
    return out


def color_inputs():
    """[(name, planes float32 [4, h2, w2] in R, G, B, W order, gray float32 [H, W])] -- reconstructions before the uint8 truncation."""
    out = []
    for name, (h2, w2), seed in [('small', (36, 48), 21), ('bsergb_half', (312, 485), 22)]:
        rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:h2, 0:w2]
        base = 0.5 + 0.3 * np.sin(xx / 7.0) * np.cos(yy / 5.0)
        planes = np.stack([np.clip(base * s + 0.05 * rng.standard_normal((h2, w2)), -0.1, 1.1) for s in (1.0, 0.8, 0.6, 0.9)]).astype(np.float32)
        YY, XX = np.mgrid[0:2 * h2, 0:2 * w2]
        gray = np.clip(0.5 + 0.35 * np.sin(XX / 14.0) * np.cos(YY / 10.0) + 0.03 * rng.standard_normal((2 * h2, 2 * w2)), -0.1, 1.1).astype(np.float32)
        out.append((name, planes, gray))
    return out


def lpips_pairs():
    H, W = 260, 346
    rng = np.random.default_rng(17)
    yy, xx = np.mgrid[0:H, 0:W]
    ref = np.stack([0.5 + 0.4 * np.sin(xx / (7.0 + i)) * np.cos(yy / (9.0 + i)) for i in range(4)]).astype(np.float32)
    img = np.clip(ref + 0.1 * rng.standard_normal(ref.shape), 0, 1).astype(np.float32)
    return img, np.clip(ref, 0, 1).astype(np.float32)


# ---- the reference's calls ----------------------------------------------------------------------------------------------------------
def skimage_mse(img, ref):
    """MseMetric.calculate, utils/eval_metrics.py:82-84: mse(ref, img)."""
    from skimage.metrics import mean_squared_error
    return float(mean_squared_error(ref, img))


def skimage_ssim(img, ref):
    """SsimMetric.calculate, utils/eval_metrics.py:95-97."""
    from skimage.metrics import structural_similarity
    return float(structural_similarity(ref, img, gaussian_weights=True, sigma=1.5, use_sample_covariance=False, data_range=1.0))


def thirdparty_histeq(img, mode):
    """EvalMetricsTracker.histogram_equalization, utils/eval_metrics.py:326-350 (img: float32 [H, W] in [0, 1])."""
    from skimage.util import img_as_float32, img_as_ubyte
    if mode == 'global':
        from skimage import exposure
        return img_as_float32(exposure.equalize_hist(img))
    if mode == 'local':
        from skimage.filters import rank
        from skimage.morphology import disk
        try:
            return img_as_float32(rank.equalize(img_as_ubyte(img), footprint=disk(55)))
        except TypeError:      # scikit-image < 0.19 calls the same argument `selem` (the reference's spelling needs >= 0.19)
            return img_as_float32(rank.equalize(img_as_ubyte(img), selem=disk(55)))
    if mode == 'clahe':
        import cv2
        clahe = cv2.createCLAHE(clipLimit=2.0, tileGridSize=(8, 8))
        return img_as_float32(clahe.apply(img_as_ubyte(img)))
    raise ValueError(mode)


def _shift_image(X, dx, dy):
    """utils/color_utils.py:5-16 (only the dx, dy >= 0 cases merge_channels_into_color_image uses)."""
    X = np.roll(X, dy, axis=0)
    X = np.roll(X, dx, axis=1)
    if dy > 0:
        X[:dy, :] = np.expand_dims(X[dy, :], axis=0)
    if dx > 0:
        X[:, :dx] = np.expand_dims(X[:, dx], axis=1)
    return X


def cv2_color_merge(planes, gray):
    """ColorNet.forward's tail, model/model.py:100-104, then merge_channels_into_color_image, utils/color_utils.py:53-88.  Uses the
    reference's own function when /root/reference is present (build container), the same OpenCV calls restated otherwise."""
    import cv2
    to_u8 = lambda a: np.clip(a * 255, 0, 255).astype(np.uint8)          # model/model.py:100-101 (truncation)
    ch = {'R': to_u8(planes[0]), 'G': to_u8(planes[1]), 'B': to_u8(planes[2]), 'W': to_u8(planes[3]), 'grayscale': to_u8(gray)}
    ref_dir = '/root/reference'
    if os.path.isdir(ref_dir):
        import importlib.util
        spec = importlib.util.spec_from_file_location('_ref_color_utils', os.path.join(ref_dir, 'utils', 'color_utils.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.merge_channels_into_color_image(ch)
    for c in ['R', 'G', 'W', 'B']:
        ch[c] = cv2.resize(ch[c], dsize=None, fx=2, fy=2, interpolation=cv2.INTER_LINEAR)
    ch['B'] = _shift_image(ch['B'], dx=1, dy=1)
    ch['G'] = _shift_image(ch['G'], dx=1, dy=0)
    ch['W'] = _shift_image(ch['W'], dx=0, dy=1)
    bgr = np.dstack([ch['B'], cv2.addWeighted(src1=ch['G'], alpha=0.5, src2=ch['W'], beta=0.5, gamma=0.0, dtype=cv2.CV_8U), ch['R']])
    lab = cv2.cvtColor(src=bgr, code=cv2.COLOR_BGR2LAB)
    lab[:, :, 0] = ch['grayscale']
    return cv2.cvtColor(src=lab, code=cv2.COLOR_LAB2BGR)


def pyiqa_lpips():
    """pyiqa.create_metric('lpips') as PyIqaMetricFactory.get_metric does (utils/eval_metrics.py:110-125).  Returns (callable on
    float32 [n, H, W] gray pairs, state_dict as numpy) or raises when pyiqa cannot find / fetch its weights."""
    import pyiqa
    import torch
    metric = pyiqa.create_metric('lpips', device='cpu')
    net = metric.net
    sd = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in net.state_dict().items()}

    def run(img, ref):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().unsqueeze(1).repeat(1, 3, 1, 1)      # cv2torch(img, num_ch=3), utils/eval_utils.py:46-54
        with torch.no_grad():
            return metric(t(img), t(ref)).reshape(-1).double().numpy()
    return run, sd

"""Oracle: per-frame metrics and score aggregation.  TEST INFRASTRUCTURE ONLY.

PARITY PINNED (round 6) for mse()/ssim() against the real scikit-image: the arithmetic lives in scikit-image, which is not in the
reference tree (requirements.txt lists it without a version) and not installable for the system interpreter -- but this image
carries an Anaconda python3.9 with scikit-image 0.18.3 (/opt/conda/bin/python3.9), and tests/golden/thirdparty_metrics.json holds
ITS mean_squared_error / structural_similarity for the seeded pairs of tests/thirdparty_refs.py (written by
`/opt/conda/bin/python3.9 tests/golden/make_thirdparty_golden.py`): mse() agrees bit for bit, ssim() to 5e-8 (0.18 filters in
float64, >= 0.19 -- whose float32 semantics this file follows -- in float32).  tests/test_thirdparty_pins.py holds oracle and HIP
kernel to the fixture and, wherever skimage imports, to the live package.  The statement below follows the published algorithm
(Wang et al. 2004) as implemented by scikit-image (skimage/metrics/simple_metrics.py, _structural_similarity.py) with exactly the
kwargs the reference passes at utils/eval_metrics.py:83 and :96.

  mse   mean((ref-img)^2): fp32 difference and square, mean accumulated in fp64.
  ssim  gaussian_weights=True, sigma=1.5, use_sample_covariance=False, data_range=1.0:
        11x11 separable Gaussian (truncate 3.5 -> radius 5) via scipy.ndimage.gaussian_filter
        mode='reflect', fp32 images, K1=0.01, K2=0.03, cov_norm=1, mean (fp64) of the S map
        cropped by 5 px per side.
  clip, BaseMetric mean, MetricTracker   utils/eval_metrics.py:45-71,253-255; eval.py:249-276
"""
import math
import numpy as np
from scipy.ndimage import gaussian_filter

F32 = np.float32


def mse(img, ref):
    a = np.asarray(ref, dtype=F32); b = np.asarray(img, dtype=F32)
    d = (a - b).astype(F32)
    return float(np.mean((d * d).astype(F32), dtype=np.float64))


def ssim(img, ref, sigma=1.5, data_range=1.0):
    X = np.asarray(ref, dtype=F32); Y = np.asarray(img, dtype=F32)
    truncate = 3.5
    r = int(truncate * sigma + 0.5)
    win = 2 * r + 1
    fa = dict(sigma=sigma, truncate=truncate, mode='reflect')
    ux = gaussian_filter(X, **fa); uy = gaussian_filter(Y, **fa)
    uxx = gaussian_filter(X * X, **fa); uyy = gaussian_filter(Y * Y, **fa)
    uxy = gaussian_filter(X * Y, **fa)
    cov_norm = 1.0
    vx = cov_norm * (uxx - ux * ux); vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    C1 = (0.01 * data_range) ** 2; C2 = (0.03 * data_range) ** 2
    A1 = 2 * ux * uy + C1; A2 = 2 * vxy + C2
    B1 = ux ** 2 + uy ** 2 + C1; B2 = vx + vy + C2
    S = (A1 * A2) / (B1 * B2)
    pad = (win - 1) // 2
    return float(S[pad:-pad, pad:-pad].mean(dtype=np.float64))


def clip01(a):
    """utils/eval_metrics.py:253-255."""
    return np.clip(a, 0.0, 1.0)


def mean_score(scores):
    """BaseMetric.update/get_mean_score, utils/eval_metrics.py:45-71: keep finite, mean or -1."""
    s = [x for x in scores if math.isfinite(x) and not math.isnan(x)]
    return -1 if not s else sum(s) / len(s)


class MetricTracker:
    """eval.py:249-276 (count==0 updates are ignored; weighted running average)."""

    def __init__(self):
        self.data = {}

    def update(self, key, value, count=1):
        if count == 0:
            return
        d = self.data.setdefault(key, {'total': 0.0, 'count': 0, 'average': 0.0})
        d['total'] += value * count
        d['count'] += count
        d['average'] = d['total'] / d['count']

    def average(self, key):
        return self.data[key]['average'] if key in self.data else 0.0

    def count(self, key):
        return self.data[key]['count'] if key in self.data else 0

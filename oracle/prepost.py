"""Oracle: pre/post-processing around the network.  TEST INFRASTRUCTURE ONLY.

  normalize_event_tensor      eval.py:398-410
  crop_params / pad / crop    utils/util.py:20-59 (optimal_crop_size, CropParameters)
  post_process_normalization  eval.py:380-395 + utils/eval_utils.py:15-35 (np.percentile,
                              numpy's default 'linear' method; numpy IS the reference's
                              implementation here, so the oracle calls it directly)
"""
from math import ceil, floor
import numpy as np

F32 = np.float32


def normalize_event_tensor(v):
    """eval.py:398-410.  Statistics over non-zeros: mean = sum/nnz,
    std = sqrt(sumsq/nnz - mean^2), std = max(std, 1e-6), out = mask*(v-mean)/std.
    The reference reduces with torch.sum (fp32, order depends on thread count); the oracle
    reduces in fp64 and rounds once -> documented tolerance 2e-6 relative on the output."""
    v = np.asarray(v, dtype=F32)
    nz = v != 0
    n = int(nz.sum())
    if n == 0:
        return v.copy()
    s = F32(v.sum(dtype=np.float64))
    ss = F32((v.astype(np.float64) ** 2).sum())
    nf = F32(n)
    mean = F32(s / nf)
    var = F32(F32(ss / nf) - F32(mean * mean))
    std = F32(np.sqrt(var))
    std = max(std, F32(1e-6))
    out = (nz.astype(F32) * (v - mean).astype(F32)).astype(F32)
    return (out / std).astype(F32)


def optimal_crop_size(max_size, max_subsample_factor):
    """utils/util.py:20-27 with safety_margin=0."""
    f = 2 ** max_subsample_factor
    return int(f * ceil(max_size / f))


class CropParams:
    """utils/util.py:30-59."""

    def __init__(self, width, height, num_encoders):
        self.width, self.height = width, height
        self.width_crop_size = optimal_crop_size(width, num_encoders)
        self.height_crop_size = optimal_crop_size(height, num_encoders)
        self.padding_top = ceil(0.5 * (self.height_crop_size - height))
        self.padding_bottom = floor(0.5 * (self.height_crop_size - height))
        self.padding_left = ceil(0.5 * (self.width_crop_size - width))
        self.padding_right = floor(0.5 * (self.width_crop_size - width))
        cx, cy = floor(self.width_crop_size / 2), floor(self.height_crop_size / 2)
        self.ix0, self.ix1 = cx - floor(width / 2), cx + ceil(width / 2)
        self.iy0, self.iy1 = cy - floor(height / 2), cy + ceil(height / 2)

    def pad(self, x):
        pw = [(0, 0)] * (x.ndim - 2) + [(self.padding_top, self.padding_bottom),
                                       (self.padding_left, self.padding_right)]
        return np.pad(x, pw)

    def crop(self, x):
        return x[..., self.iy0:self.iy1, self.ix0:self.ix1]


def percentile_normalize(img, q_min, q_max):
    """utils/eval_utils.py:15-35 (robust_min/robust_max/normalize)."""
    img = np.asarray(img)
    lo = np.percentile(img.ravel(), q_min)
    hi = np.percentile(img.ravel(), q_max)
    return (img - lo) / (hi - lo)


def post_process_normalization(img, norm):
    """eval.py:380-395."""
    if norm == 'robust':
        return percentile_normalize(img, 1, 99)
    if norm == 'standard':
        return percentile_normalize(img, 0, 100)
    if norm == 'none':
        return img
    if norm == 'exprobust':
        return percentile_normalize(np.exp(img), 1, 99)
    raise ValueError(f"Unrecognized normalization argument: {norm}")

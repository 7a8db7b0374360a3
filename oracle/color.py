"""Oracle: ColorNet stream split and colour merge.  TEST INFRASTRUCTURE ONLY.

  bayer_split   model/model.py:54-57,81-99 (R,G,B,W sub-lattices of the event tensor)
  to_u8         model/model.py:100-101     (np.clip(img*255, 0, 255).astype(np.uint8): truncation)
  merge         utils/color_utils.py:4-88  PARITY UNPINNED: the reference calls cv2.resize / cv2.addWeighted /
                cv2.cvtColor on uint8 images; cv2 is not installed, so this restates OpenCV's documented formulas
                (bilinear with half-pixel centres, round-half-even blend, sRGB/D65 CIE Lab with the 8-bit convention
                L*255/100, a+128, b+128) in floating point -- a few LSB from OpenCV's fixed-point tables are possible.
"""
import numpy as np


def bayer_split(vox):
    """[N,B,H,W] -> [N,4,B,H/2,W/2] in R,G,B,W order."""
    return np.stack([vox[:, :, 0::2, 0::2], vox[:, :, 0::2, 1::2], vox[:, :, 1::2, 1::2], vox[:, :, 1::2, 0::2]], axis=1)


def to_u8(img):
    return np.clip(img * 255, 0, 255).astype(np.uint8)


def _resize2x(p):
    p = p.astype(np.float32)
    h, w = p.shape
    def coords(n_out, n_in):
        f = (np.arange(n_out) + 0.5) * 0.5 - 0.5
        i0 = np.floor(f).astype(int); l = (f - i0).astype(np.float32)
        l[i0 < 0] = 0; i0[i0 < 0] = 0
        l[i0 >= n_in - 1] = 0; i0[i0 >= n_in - 1] = n_in - 1
        return i0, np.minimum(i0 + 1, n_in - 1), l
    y0, y1, ly = coords(2 * h, h); x0, x1, lx = coords(2 * w, w)
    a, b = p[y0][:, x0], p[y0][:, x1]; c, d = p[y1][:, x0], p[y1][:, x1]
    lx = lx[None, :]; ly = ly[:, None]
    v = (1 - ly) * ((1 - lx) * a + lx * b) + ly * ((1 - lx) * c + lx * d)
    return np.floor(v + np.float32(0.5)).astype(np.float32)


def _shift(X, dx, dy):
    H, W = X.shape
    ys = np.maximum(np.arange(H) - dy, 0); xs = np.maximum(np.arange(W) - dx, 0)
    return X[ys][:, xs]


def _f(t):
    return np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16.0 / 116.0).astype(np.float32)


def _finv(t):
    t3 = t * t * t
    return np.where(t3 > 0.008856, t3, (t - 16.0 / 116.0) / 7.787).astype(np.float32)


def merge(planes, gray):
    """planes: float [4,h2,w2] (R,G,B,W reconstructions), gray: float [H,W] -> uint8 BGR [H,W,3]."""
    f32 = np.float32
    up = [_resize2x(to_u8(p)) for p in planes]
    R, G, B, Wc = up[0], _shift(up[1], 1, 0), _shift(up[2], 1, 1), _shift(up[3], 0, 1)
    Gm = np.rint(f32(0.5) * G + f32(0.5) * Wc).astype(f32)
    lin = lambda c: np.where(c <= 0.04045, c / 12.92, np.power((c + 0.055) / 1.055, 2.4)).astype(f32)
    r, g, b = lin(R / f32(255)), lin(Gm / f32(255)), lin(B / f32(255))
    X = (f32(0.412453) * r + f32(0.357580) * g + f32(0.180423) * b) / f32(0.950456)
    Z = (f32(0.019334) * r + f32(0.119193) * g + f32(0.950227) * b) / f32(1.088754)
    Y = f32(0.212671) * r + f32(0.715160) * g + f32(0.072169) * b
    fx, fy, fz = _f(X), _f(Y), _f(Z)
    a8 = np.clip(np.rint(500 * (fx - fy) + 128), 0, 255).astype(f32)
    b8 = np.clip(np.rint(200 * (fy - fz) + 128), 0, 255).astype(f32)
    L = to_u8(gray).astype(f32) * f32(100) / f32(255)
    fy2 = (L + 16) / 116; fx2 = fy2 + (a8 - 128) / 500; fz2 = fy2 - (b8 - 128) / 200
    X = _finv(fx2) * f32(0.950456); Y = _finv(fy2); Z = _finv(fz2) * f32(1.088754)
    ro = f32(3.240479) * X - f32(1.537150) * Y - f32(0.498535) * Z
    go = f32(-0.969256) * X + f32(1.875991) * Y + f32(0.041556) * Z
    bo = f32(0.055648) * X - f32(0.204043) * Y + f32(1.057311) * Z
    def to8(c):
        c = np.clip(c, 0, 1)
        s = np.where(c <= 0.0031308, 12.92 * c, 1.055 * np.power(c, 1 / 2.4) - 0.055)
        return np.clip(np.rint(s * 255), 0, 255).astype(np.uint8)
    return np.stack([to8(bo), to8(go), to8(ro)], axis=-1)

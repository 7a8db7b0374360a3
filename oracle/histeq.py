"""Oracle: the tracker's histogram-equalisation modes (utils/eval_metrics.py:326-350).  TEST INFRASTRUCTURE ONLY.

'global' and 'local': PARITY PINNED (round 6) to scikit-image 0.18.3 (/opt/conda/bin/python3.9 of this image; fixture
tests/golden/thirdparty_histeq.npz, tests/test_thirdparty_pins.py): equalize_global within one float32 ulp (6e-8), equalize_local
exact.  'clahe': PARITY UNPINNED (OpenCV is nowhere in the image).  The reference calls scikit-image (exposure.equalize_hist,
filters.rank.equalize, img_as_ubyte / img_as_float32) and OpenCV (createCLAHE); requirements.txt pins no versions.  This restates
their published algorithms on numpy:
  equalize_global  skimage/exposure/exposure.py: histogram(image, 256) over [min, max] (np.histogram), bin centres,
                   cdf = cumsum / total (float32), np.interp(image, centres, cdf) -> float32
  to_u8 / to_f32   skimage/util/dtype.py: rint(x * 255) clipped; u8 * float32(1/255)
  equalize_local   skimage/filters/rank/generic_cy.pyx _kernel_equalize over footprint disk(55): out = uint8(255 *
                   #{footprint pixels inside the image with value <= centre} / #{footprint pixels inside the image})
  clahe            opencv/modules/imgproc/src/clahe.cpp (clipLimit 2.0, 8x8 tiles): BORDER_REFLECT_101 extension to whole
                   tiles, per-tile clipped histogram with uniform redistribution (+ residual every `step` bins), LUT =
                   round(cumsum * 255 / tileArea), bilinear blend of the four neighbouring tile LUTs, round
"""
import numpy as np


def equalize_global(img):
    img = np.asarray(img, np.float32)
    hist, edges = np.histogram(img.flatten(), bins=256, range=None)
    centers = (edges[:-1] + edges[1:]) / 2.0
    cdf = hist.cumsum()
    cdf = (cdf / float(cdf[-1])).astype(np.float32)
    return np.interp(img.flat, centers, cdf).reshape(img.shape).astype(np.float32)


def to_u8(img):
    x = np.multiply(np.asarray(img, np.float32), 255, dtype=np.float32)
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def to_f32(u8):
    return np.multiply(u8, 1.0 / 255, dtype=np.float32)


def equalize_local(img, radius=55):
    u8 = to_u8(img)
    H, W = u8.shape
    out = np.zeros((H, W), np.uint8)
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    fp = (X ** 2 + Y ** 2) <= radius ** 2
    for y in range(H):
        y0, y1 = max(y - radius, 0), min(y + radius, H - 1)
        for x in range(W):
            x0, x1 = max(x - radius, 0), min(x + radius, W - 1)
            win = u8[y0:y1 + 1, x0:x1 + 1]
            m = fp[y0 - y + radius:y1 - y + radius + 1, x0 - x + radius:x1 - x + radius + 1]
            pop = int(m.sum())
            s = int((win[m] <= u8[y, x]).sum())
            out[y, x] = int((255 * s) / float(pop)) if pop else 0
    return to_f32(out)


def clahe(img, clip=2.0, tiles=8):
    u8 = to_u8(img)
    H, W = u8.shape
    pb = 0 if H % tiles == 0 else tiles - H % tiles
    pr = 0 if W % tiles == 0 else tiles - W % tiles
    ext = np.pad(u8, ((0, pb), (0, pr)), mode='reflect') if (pb or pr) else u8
    th, tw = ext.shape[0] // tiles, ext.shape[1] // tiles
    area = tw * th
    lut_scale = np.float32(255) / np.float32(area)
    limit = max(int(clip * area / 256), 1)
    luts = np.zeros((tiles, tiles, 256), np.uint8)
    for ty in range(tiles):
        for tx in range(tiles):
            h = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            clipped = int(np.maximum(h - limit, 0).sum())
            h = np.minimum(h, limit)
            batch = clipped // 256
            residual = clipped - batch * 256
            h += batch
            if residual:
                step = max(256 // residual, 1)
                i = 0
                while i < 256 and residual > 0:
                    h[i] += 1
                    i += step; residual -= 1
            c = np.cumsum(h).astype(np.float32) * lut_scale
            luts[ty, tx] = np.clip(np.rint(c), 0, 255).astype(np.uint8)
    ys, xs = np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32)
    tyf = ys * (np.float32(1) / np.float32(th)) - np.float32(0.5)
    txf = xs * (np.float32(1) / np.float32(tw)) - np.float32(0.5)
    ty1 = np.floor(tyf).astype(int); tx1 = np.floor(txf).astype(int)
    ya = (tyf - ty1).astype(np.float32); xa = (txf - tx1).astype(np.float32)
    ty2 = np.minimum(ty1 + 1, tiles - 1); tx2 = np.minimum(tx1 + 1, tiles - 1)
    ty1 = np.maximum(ty1, 0); tx1 = np.maximum(tx1, 0)
    v = u8.astype(int)
    f = lambda ty, tx: luts[ty[:, None], tx[None, :], v].astype(np.float32)
    ya_, xa_ = ya[:, None], xa[None, :]
    res = (f(ty1, tx1) * (1 - xa_) + f(ty1, tx2) * xa_) * (1 - ya_) + (f(ty2, tx1) * (1 - xa_) + f(ty2, tx2) * xa_) * ya_
    return to_f32(np.clip(np.rint(res), 0, 255).astype(np.uint8))


def histogram_equalization(img, mode):
    if mode == 'none':
        return img
    return {'global': equalize_global, 'local': equalize_local, 'clahe': clahe}[mode](img)

"""Synthetic event streams in the reference's on-disk sequence format.

The datasets the reference evaluates on (ECD, MVSEC, HQF, BS-ERGB ...) are not available
offline; SURVEY.md section 8d defines the stand-in used for parity and benchmarking:
a homogeneous Poisson stream, t = sort(U(0,T)) float64 seconds, x~U{0..W-1}, y~U{0..H-1},
p~Bernoulli(1/2), numpy default_rng(seed).  Files follow tools/bag_to_npy.py:53-94 /
dataset.py:230-281 of the reference (events_ts/xy/p.npy, images.npy, images_ts.npy,
image_event_indices.npy, metadata.json).
"""
import json
import os

import numpy as np


def poisson_events(seed, n_events, rate_hz, width, height):
    rng = np.random.default_rng(seed)
    T = n_events / float(rate_hz)
    t = np.sort(rng.uniform(0.0, T, n_events))
    x = rng.integers(0, width, n_events, dtype=np.int16)
    y = rng.integers(0, height, n_events, dtype=np.int16)
    p = rng.integers(0, 2, n_events, dtype=np.uint8)
    return t, x, y, p


def smooth_frames(seed, n_frames, width, height):
    """uint8 [F,H,W,1] smooth moving blobs (well-conditioned for SSIM)."""
    rng = np.random.default_rng(seed + 7919)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    frames = np.empty((n_frames, height, width, 1), dtype=np.uint8)
    ph = rng.uniform(0, 2 * np.pi, 4)
    for f in range(n_frames):
        a = 0.5 + 0.25 * np.sin(xx / 23.0 + ph[0] + 0.11 * f) * np.cos(yy / 17.0 + ph[1] - 0.07 * f) \
            + 0.2 * np.sin((xx + yy) / 41.0 + ph[2] + 0.05 * f)
        frames[f, :, :, 0] = np.clip(np.round(a * 255), 0, 255).astype(np.uint8)
    return frames


def write_sequence(path, seed, n_events, rate_hz, width, height, fps=25.0, with_images=True):
    """Write one synthetic sequence directory; returns dict of the arrays written."""
    os.makedirs(path, exist_ok=True)
    t, x, y, p = poisson_events(seed, n_events, rate_hz, width, height)
    xy = np.stack([x, y], axis=1)
    np.save(os.path.join(path, 'events_ts.npy'), t)
    np.save(os.path.join(path, 'events_xy.npy'), xy)
    np.save(os.path.join(path, 'events_p.npy'), p)
    out = dict(t=t, xy=xy, p=p)
    if with_images:
        T = t[-1]
        n_frames = max(int(T * fps), 2)
        img_ts = ((np.arange(n_frames) + 1) / fps).reshape(-1, 1)
        img_ts = img_ts[img_ts[:, 0] <= T]
        n_frames = len(img_ts)
        images = smooth_frames(seed, n_frames, width, height)
        idx = (np.searchsorted(t, img_ts[:, 0], side='right') - 1).reshape(-1, 1).astype(np.int64)
        np.save(os.path.join(path, 'images.npy'), images)
        np.save(os.path.join(path, 'images_ts.npy'), img_ts)
        np.save(os.path.join(path, 'image_event_indices.npy'), idx)
        out.update(images=images, images_ts=img_ts, image_event_indices=idx)
    with open(os.path.join(path, 'metadata.json'), 'w') as f:
        json.dump({"sensor_resolution": [height, width]}, f)
    return out


def window_events_f32(t, xy, p, idx0, idx1):
    """The four fp32 event arrays MemMapDataset hands to the tensorizer for [idx0, idx1)
    (dataset.py:48-57,222-228): x,y -> f32; p*2-1; ts -> float32(ts - ts[idx0]) where the
    subtraction is done in float64."""
    xs = xy[idx0:idx1, 0].astype(np.float32)
    ys = xy[idx0:idx1, 1].astype(np.float32)
    ts = t[idx0:idx1]
    ts = (ts - ts[0]).astype(np.float32) if idx1 > idx0 else ts.astype(np.float32)
    ps = (p[idx0:idx1] * 2.0 - 1.0).astype(np.float32)
    return xs, ys, ts, ps


def sparse_voxels(seed, n_frames, num_bins, height, width, density=0.07):
    """[F,B,H,W] fp32 voxel-like tensors (mostly zeros, ~N(0,1) elsewhere): stand-ins for
    normalized event tensors when a test wants network inputs that do not depend on the
    tensorizer."""
    rng = np.random.default_rng([seed, n_frames, num_bins, height, width])
    v = rng.standard_normal((n_frames, num_bins, height, width)).astype(np.float32)
    m = rng.random((n_frames, num_bins, height, width)) < density
    return (v * m).astype(np.float32)

"""Oracle: events -> voxel grid (numpy restatement).  TEST INFRASTRUCTURE ONLY.

Follows utils/event_utils.py:4-59 of the reference:
  events_to_image_torch  (:4-24)   img[ys.long(), xs.long()] += ps, sequentially in
                                   event order (CPU index_put_(accumulate=True)).
  events_to_voxel_torch  (:27-59)  dt = ts[-1]-ts[0]; dt < 1e-9 -> linspace branch
                                   (:48-49) else t_norm=(ts-ts[0])/dt*(B-1) (:51);
                                   per bin w = ps*max(0, 1-|t_norm-b|) (:54-55).
All arithmetic is fp32, one IEEE rounding per operation, no fused multiply-add.
"""
import numpy as np

F32 = np.float32


def linspace_f32(start, end, steps):
    """torch.linspace(start, end, steps) in fp32 -- ATen's scalar formula
    (aten/src/ATen/native/cpu/RangeFactoriesKernel.cpp, linspace_kernel):
    step=(end-start)/(steps-1); idx<steps//2: start+step*idx else end-step*(steps-idx-1);
    steps==1 -> [start].  (ATen's SIMD path evaluates base+i*step per vector and is
    host-ISA dependent for steps >= the vector width; see DESIGN.md "linspace".)"""
    if steps == 1:
        return np.array([start], dtype=F32)
    start, end = F32(start), F32(end)
    step = F32((end - start) / F32(steps - 1))
    idx = np.arange(steps, dtype=np.int64)
    lo = (start + step * idx.astype(F32)).astype(F32)
    hi = (end - step * (steps - idx - 1).astype(F32)).astype(F32)
    return np.where(idx < steps // 2, lo, hi).astype(F32)


def t_norm_f32(ts, num_bins):
    """event_utils.py:46-51."""
    ts = np.asarray(ts, dtype=F32)
    dt = F32(ts[-1] - ts[0])
    if float(dt) < 1e-9:
        return linspace_f32(0, num_bins - 1, len(ts))
    return (((ts - ts[0]).astype(F32) / dt).astype(F32) * F32(num_bins - 1)).astype(F32)


def events_to_image(xs, ys, ps, sensor_size):
    """event_utils.py:4-24 (sequential accumulate in event order)."""
    H, W = sensor_size
    img = np.zeros((H, W), dtype=F32)
    xi = np.asarray(xs).astype(np.int64)          # .long(): truncation toward zero
    yi = np.asarray(ys).astype(np.int64)
    np.add.at(img, (yi, xi), np.asarray(ps, dtype=F32))   # unbuffered, in index order
    return img


def events_to_voxel(xs, ys, ts, ps, num_bins, sensor_size):
    """event_utils.py:27-59.  xs, ys, ts, ps: fp32 [n], n >= 1.  Returns [B,H,W] fp32."""
    xs = np.asarray(xs, dtype=F32); ys = np.asarray(ys, dtype=F32)
    ts = np.asarray(ts, dtype=F32); ps = np.asarray(ps, dtype=F32)
    assert len(xs) == len(ys) == len(ts) == len(ps)
    t_norm = t_norm_f32(ts, num_bins)
    bins = []
    for bi in range(num_bins):
        w = np.maximum(F32(0), (F32(1.0) - np.abs((t_norm - F32(bi)).astype(F32))).astype(F32))
        weights = (ps * w).astype(F32)
        bins.append(events_to_image(xs, ys, weights, sensor_size))
    return np.stack(bins)


def voxelize_windows(x, y, t, p, win_offsets, num_bins, sensor_size):
    """Batch form used by the C ABI (evr_voxelize): window w = events
    [win_offsets[w], win_offsets[w+1]); an empty window -> zeros (dataset.py:200-203)."""
    H, W = sensor_size
    nw = len(win_offsets) - 1
    out = np.zeros((nw, num_bins, H, W), dtype=F32)
    for w in range(nw):
        a, b = int(win_offsets[w]), int(win_offsets[w + 1])
        if b > a:
            out[w] = events_to_voxel(x[a:b], y[a:b], t[a:b], p[a:b], num_bins, sensor_size)
    return out

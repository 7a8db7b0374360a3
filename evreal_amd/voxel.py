"""Host side of the tensorizer: same call shape as the reference's
utils/event_utils.py:27-59 (events_to_voxel_torch) plus the batched form the C ABI offers."""
import torch

from . import lib as _lib


class Voxelizer:
    """Owns the workspace for evr_voxelize and keeps it across calls (no per-call allocation)."""

    def __init__(self, device='cuda:0'):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.ws = None

    def _workspace(self, n_events, n_windows, B, H, W, stream=None):
        need = self.lib.evr_voxelize_workspace_bytes(n_events, n_windows, B, H, W)
        if self.ws is None or self.ws.numel() < need:
            old = self.ws
            new = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=self.device)
            # the 256-B header (drop counters) starts at zero and SURVIVES a regrow; both on the stream the kernels are
            # launched on, so neither can race with them.  The rest of the workspace is scratch.
            with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream(self.device)):
                if old is None:
                    new[:256].zero_()
                else:
                    new[:256].copy_(old[:256])
                    old.record_stream(torch.cuda.current_stream(self.device))
            self.ws = new
        return self.ws

    def voxelize(self, x, y, t, p, win_offsets, num_bins, sensor_size, out=None, stats=None, stream=None):
        """x,y,t,p: fp32 cuda tensors [n]; win_offsets: int64 cuda tensor [n_windows+1].
        Returns out [n_windows, B, H, W] fp32 (every cell written)."""
        H, W = sensor_size
        n = int(x.numel()); nw = int(win_offsets.numel()) - 1
        for a in (x, y, t, p):
            assert a.is_cuda and a.dtype == torch.float32 and a.is_contiguous() and a.numel() == n
        assert win_offsets.is_cuda and win_offsets.dtype == torch.int64
        if out is None:
            out = torch.empty((nw, num_bins, H, W), dtype=torch.float32, device=x.device)
        ws = self._workspace(n, nw, num_bins, H, W, stream)
        rc = self.lib.evr_voxelize(_lib.ptr(x), _lib.ptr(y), _lib.ptr(t), _lib.ptr(p), _lib.ptr(win_offsets),
                                   nw, n, num_bins, H, W, _lib.ptr(out), _lib.ptr(stats), _lib.ptr(ws),
                                   ws.numel(), _lib.stream_ptr(stream))
        _lib.check(rc, 'evr_voxelize')
        return out

    def voxelize_raw(self, xy, ts, pol, win_offsets, num_bins, sensor_size, out=None, stats=None, stream=None, n_window_events=None):
        """Raw memmap form (dataset.py:222-228 fused): xy int16 [n,2], ts float64 [n], pol uint8 [n].
        n_window_events: the number of events the windows cover, when the caller knows it without a device read-back (consecutive
        windows of a LARGER resident stream).  The library sizes the record workspace and the split work-groups per window from the
        event count it is given: the whole stream's length made both several times too large (bench.py with 40 resident steps: 32
        split work-groups per 15k-event window instead of 8, 614 MB of records instead of 123 MB, the launch 1.5x slower)."""
        if n_window_events is not None:
            begin, end = win_offsets[:-1], win_offsets[1:]
            return self.voxelize_raw_windows(xy, ts, pol, begin, end, begin - win_offsets[0], int(n_window_events), num_bins,
                                             sensor_size, out=out, stats=stats, stream=stream)
        H, W = sensor_size
        n = int(ts.numel()); nw = int(win_offsets.numel()) - 1
        assert xy.dtype == torch.int16 and ts.dtype == torch.float64 and pol.dtype == torch.uint8
        assert xy.is_contiguous() and ts.is_contiguous() and pol.is_contiguous()
        if out is None:
            out = torch.empty((nw, num_bins, H, W), dtype=torch.float32, device=ts.device)
        ws = self._workspace(n, nw, num_bins, H, W, stream)
        rc = self.lib.evr_voxelize_raw(_lib.ptr(xy), _lib.ptr(ts), _lib.ptr(pol), _lib.ptr(win_offsets), nw, n,
                                       num_bins, H, W, _lib.ptr(out), _lib.ptr(stats), _lib.ptr(ws), ws.numel(),
                                       _lib.stream_ptr(stream))
        _lib.check(rc, 'evr_voxelize_raw')
        return out

    def voxelize_raw_windows(self, xy, ts, pol, win_begin, win_end, rec_base, n_window_events, num_bins,
                             sensor_size, out=None, stats=None, stream=None):
        """Arbitrary (overlapping / empty) windows of one resident raw stream; all index tensors int64 cuda."""
        H, W = sensor_size
        nw = int(win_begin.numel())
        if out is None:
            out = torch.empty((nw, num_bins, H, W), dtype=torch.float32, device=ts.device)
        ws = self._workspace(int(n_window_events), nw, num_bins, H, W, stream)
        rc = self.lib.evr_voxelize_raw_windows(_lib.ptr(xy), _lib.ptr(ts), _lib.ptr(pol), _lib.ptr(win_begin),
                                               _lib.ptr(win_end), _lib.ptr(rec_base), nw, int(n_window_events),
                                               num_bins, H, W, _lib.ptr(out), _lib.ptr(stats), _lib.ptr(ws),
                                               ws.numel(), _lib.stream_ptr(stream))
        _lib.check(rc, 'evr_voxelize_raw_windows')
        return out

    def dropped(self):
        """Out-of-sensor events of the LAST call (synchronises)."""
        import ctypes
        v = ctypes.c_int64(0)
        _lib.check(self.lib.evr_voxelize_dropped(_lib.ptr(self.ws), ctypes.byref(v), _lib.stream_ptr()),
                   'evr_voxelize_dropped')
        return v.value

    def dropped_total(self):
        """Out-of-sensor events of EVERY call of this Voxelizer so far (synchronises): what a frame loop polls once per
        sequence or batch."""
        import ctypes
        if self.ws is None:
            return 0
        v = ctypes.c_int64(0)
        _lib.check(self.lib.evr_voxelize_dropped_total(_lib.ptr(self.ws), ctypes.byref(v), _lib.stream_ptr()),
                   'evr_voxelize_dropped_total')
        return v.value

    def raise_if_dropped(self, what='events'):
        """The reference raises from index_put_ when a pixel lies outside the sensor; here such events
        are dropped by the kernel, so the host loop turns a non-zero count into the same failure."""
        n = self.dropped_total()
        if n:
            raise IndexError(f"{n} {what} fall outside the sensor (wrong sensor_resolution / metadata.json?): "
                             "index out of range in the voxel grid, as events_to_image_torch's index_put_ would report")


_default = None


def events_to_voxel_torch(xs, ys, ts, ps, num_bins, device=None, sensor_size=(180, 240)):
    """Drop-in for utils/event_utils.py:27-59: one window, tensors in, [B,H,W] tensor out.
    Inputs are moved to the GPU; the result stays there."""
    global _default
    if _default is None:
        _default = Voxelizer()
    dev = _default.device
    assert len(xs) == len(ys) == len(ts) == len(ps)
    f = lambda a: a.to(device=dev, dtype=torch.float32).contiguous()
    offs = torch.tensor([0, len(xs)], dtype=torch.int64, device=dev)
    return _default.voxelize(f(xs), f(ys), f(ts), f(ps), offs, num_bins, tuple(sensor_size))[0]

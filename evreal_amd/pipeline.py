"""The per-frame hot loop of eval.py:203-238 for a batch of independent sequences, fully on the GPU:

    events window -> voxel grid (+stats) -> [normalize] -> pad -> network -> crop -> [robust norm]
                  -> clip -> MSE / SSIM / LPIPS

All launches go through the C ABI; nothing synchronises with the host inside a step (the reference forces a
cuda.synchronize() and a D2H copy per frame, eval.py:227-233).  With `overlap=True` the evaluation half of a frame
(robust normalisation, MSE/SSIM, LPIPS) runs on a second HIP stream while the first already reconstructs the next
frame: those kernels are small and leave most of the chip idle on their own; HIP events order the two streams and
the image buffer is double-buffered.  Scores then land in `scores_out` asynchronously: call flush() and synchronise
before reading them or the (post-normalised) image -- the evaluation of the newest frame is held back until the next
frame has been enqueued, so that it can start behind an event recorded INSIDE that frame (EVR_EVAL_GATE).
Contract of the deferred evaluation: `ref` and `scores_out` passed to step t are READ / WRITTEN during step t + 1 (or
flush()), so the caller must not overwrite `ref` in place between the two calls (pass a fresh tensor per frame, as
evreal_amd.eval does, or use overlap=False); a caller that only does `step(); synchronize()` sees the newest frame's
scores one step late.
"""
import torch

from . import lib as _lib
from .prepost import Metrics, post_process_normalization
from .voxel import Voxelizer


def default_eval_gate(n_seq, environ=None):
    """Layer of frame t+1 behind which the evaluation of frame t may start: EVR_EVAL_GATE if set ('none' = ungated), else after the
    first residual block, from 48 sequences per step on after the second one (measured: profiles/r03_gate_ab_workloads.txt)."""
    import os
    g = (os.environ if environ is None else environ).get('EVR_EVAL_GATE')
    return g or ('res1.conv2' if n_seq >= 48 else 'res0.conv2')


class HotPath:
    def __init__(self, model, num_bins, sensor_size, n_seq, event_tensor_normalization=True,
                 post_process_norm='robust', metrics=('mse', 'ssim'), device='cuda:0', lpips=None, overlap=False):
        _lib.require_gpu()
        self.model, self.B, (self.H, self.W), self.n = model, num_bins, sensor_size, n_seq
        self.norm_in, self.post = event_tensor_normalization, post_process_norm
        self.want_mse, self.want_ssim = 'mse' in metrics, 'ssim' in metrics
        self.lpips = lpips if 'lpips' in metrics else None      # evreal_amd.lpips.LPIPS instance
        self.ncol = 3 if self.lpips is not None else 2
        self.dev = torch.device(device)
        self.vox = Voxelizer(device)
        self.met = Metrics()
        self.grid = torch.empty((n_seq, num_bins, self.H, self.W), dtype=torch.float32, device=self.dev)
        self.stats = torch.zeros((n_seq, 3), dtype=torch.float64, device=self.dev)
        self.img = torch.empty((n_seq, 1, self.H, self.W), dtype=torch.float32, device=self.dev)
        self.overlap = bool(overlap)
        if self.overlap:
            # EVR_IMG_BUFFERS image buffers in rotation (default 3): frame t + 2 must not wait for the evaluation of frame t, which starts
            # late inside frame t + 1 (the gate below) and ends after it -- with two buffers the reconstruction stream stalled at
            # every frame start until that evaluation had released its buffer
            import os as _os
            nbuf = max(2, int(_os.environ.get('EVR_IMG_BUFFERS', '3') or 3))
            self.imgs = [self.img] + [torch.empty_like(self.img) for _ in range(nbuf - 1)]
            self.side = self._side_stream()
            self.ev_model = [torch.cuda.Event() for _ in range(nbuf)]    # image k is complete (main stream)
            self.ev_done = [None] * nbuf                                 # evaluation of image k has finished (side stream)
            self.k = 0
            # The evaluation of frame t starts only when frame t+1 has passed the layer EVR_EVAL_GATE names (the library records an
            # event there; 'none': at once, as in rounds 1-2): the evaluation kernels then share the chip with the residual blocks /
            # decoders of the next frame instead of its head and ConvLSTM layers (2.2 ms -> 1.5 ms for enc0.rec in the step; +1 % end
            # to end).  Default: after res0.conv2; from 48 sequences per step on after res1.conv2 -- there the evaluation half fits
            # beside the three decoders (+1 % at 64 sequences), while a small batch is a latency chain and loses 5-6 % to the later
            # start (tools/gate_sweep.sh, tools/gate_ab_workloads.sh; profiles/r03_gate_sweep.txt).  The scores of frame t land one
            # step later; flush() ends a run.  Models without that layer (FireNet) are not gated.
            import ctypes, os
            self._gate = None
            self._pending = None
            g = default_eval_gate(n_seq)
            if g and g != 'none' and hasattr(model, 'set_gate'):
                h = ctypes.c_void_p()
                _lib.check(_lib.load().evr_event_create(ctypes.byref(h)), 'evr_event_create')
                # (a layout without the chosen layer -- one residual block, no residual blocks -- falls back to the earlier gate
                # positions before giving the gate up; FireNet has none of them and runs ungated)
                for layer in dict.fromkeys([g, 'res0.conv2', 'enc2.rec']):
                    try:
                        model.set_gate(layer, h)
                        self._gate, self._gate_layer = h, layer
                        model._gate_owner = self
                        break
                    except _lib.EvrError:
                        continue
                if self._gate is None:
                    _lib.load().evr_event_destroy(h)
        self._vox_events = None
        self._ring, self._ring_stats, self._ring_pos, self._ring_len = None, None, 0, 0
        model.reset_states()

    def close(self):
        """Release the HIP objects this HotPath created through the C ABI (gate event, CU-masked stream)."""
        lib = _lib.load()
        gate = getattr(self, '_gate', None)
        if gate is not None:
            try:
                torch.cuda.synchronize(self.dev)
                if getattr(self.model, '_gate_owner', None) is self and getattr(self.model, 'handle', None) is not None:
                    self.model.set_gate(None, None)          # the model must not record into a destroyed event
                    self.model._gate_owner = None
                lib.evr_event_destroy(gate)
            finally:
                self._gate = None
        side = getattr(self, '_side_handle', None)
        if side is not None:
            torch.cuda.synchronize(self.dev)
            lib.evr_stream_destroy(side)
            self._side_handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _side_stream(self):
        """The evaluation stream.  EVR_SIDE_CUS=n restricts it to n compute units spread over the XCDs
        (hipExtStreamCreateWithCUMask through the C ABI); unset / 0: an ordinary stream."""
        import ctypes, os
        n = int(os.environ.get('EVR_SIDE_CUS', '0') or 0)
        if n <= 0:
            return torch.cuda.Stream(device=self.dev)
        h = ctypes.c_void_p()
        lib = _lib.load()
        _lib.check(lib.evr_stream_create_cu_masked(self.dev.index or 0, n, int(os.environ.get('EVR_SIDE_TOP', '1')), ctypes.byref(h)),
                   'evr_stream_create_cu_masked')
        self._side_handle = h
        return torch.cuda.ExternalStream(h.value, device=self.dev)

    def time_voxelizer(self, on):
        """on=True: bracket the tensorizer launches of every following step with HIP events on the stream they are
        launched on.  on=False: stop and return their average duration in ms (None if nothing was recorded)."""
        if on:
            self._vox_events = []
            return None
        ev, self._vox_events = self._vox_events, None
        if not ev:
            return None
        ev[-1][1].synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    def check_dropped(self):
        """Raise if any event of any step so far fell outside the sensor (the reference raises from index_put_; the
        kernel drops and counts).  Synchronises: call it once per sequence / batch, not per step."""
        self.vox.raise_if_dropped()

    def step_raw(self, xy, ts, pol, win_offsets, ref=None, scores_out=None, n_window_events=None):
        """One frame for every sequence.  xy/ts/pol: resident raw event arrays; win_offsets: int64
        [n_seq+1] device tensor delimiting this step's n_seq windows.  ref: [n_seq,H,W] reference
        frames (already /255) or None.  n_window_events: events inside these windows if known on the host (Voxelizer.voxelize_raw).
        Returns (img [n_seq,1,H,W], scores [n_seq,2|3] = mse, ssim[, lpips] or None)."""
        if self._vox_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.vox.voxelize_raw(xy, ts, pol, win_offsets, self.B, (self.H, self.W), out=self.grid, stats=self.stats,
                              n_window_events=n_window_events)
        if self._vox_events is not None:
            e1.record()
            self._vox_events.append((e0, e1))
        return self._rest(ref, scores_out)

    # -- tensorizer look-ahead ---------------------------------------------------------------------------------------
    # Windows do not depend on the recurrence, so the tensorizer may run for several steps at once: ONE launch over
    # A x n_seq windows (its kernels reach a higher fraction of the HBM roof on 512 windows than on 64) fills a
    # ring of voxel grids that the following A steps consume.
    def prefetch_raw(self, xy, ts, pol, win_offsets, n_steps, n_window_events=None, capacity=0):
        """win_offsets: int64 [n_steps * n_seq + 1], the windows of the next n_steps steps in step-major order.
        capacity: the most steps any later call will ask for -- the ring is allocated once for that many (a shorter first call followed by a
        longer one would otherwise re-allocate ~1 GB in the middle of a run)."""
        if self._ring is None or self._ring.shape[0] < n_steps:
            cap = max(n_steps, int(capacity))
            self._ring = torch.empty((cap, self.n, self.B, self.H, self.W), dtype=torch.float32, device=self.dev)
            self._ring_stats = torch.zeros((cap, self.n, 3), dtype=torch.float64, device=self.dev)
        assert int(win_offsets.numel()) == n_steps * self.n + 1
        if self._vox_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.vox.voxelize_raw(xy, ts, pol, win_offsets, self.B, (self.H, self.W),
                              out=self._ring[:n_steps].view(n_steps * self.n, self.B, self.H, self.W),
                              stats=self._ring_stats[:n_steps].view(n_steps * self.n, 3), n_window_events=n_window_events)
        if self._vox_events is not None:
            e1.record()
            self._vox_events.append((e0, e1))
        self._ring_pos, self._ring_len = 0, n_steps

    def step_ahead(self, ref=None, scores_out=None):
        """One frame for every sequence from the next prefetched voxel grid."""
        assert self._ring is not None and self._ring_pos < self._ring_len, "step_ahead without a prefetched window"
        i = self._ring_pos; self._ring_pos += 1
        return self._rest(ref, scores_out, self._ring[i], self._ring_stats[i])

    def step(self, x, y, t, p, win_offsets, ref=None, scores_out=None):
        self.vox.voxelize(x, y, t, p, win_offsets, self.B, (self.H, self.W), out=self.grid, stats=self.stats)
        return self._rest(ref, scores_out)

    def _enqueue_eval(self, k, ref, scores_out, gated):
        img = self.imgs[k]
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_model[k])
            if gated:
                _lib.check(_lib.load().evr_stream_wait_event(_lib.stream_ptr(self.side), self._gate), 'evr_stream_wait_event')
            im = img.view(self.n, self.H, self.W)
            if self.post != 'none':
                post_process_normalization(im, self.post)
            if ref is not None and scores_out is not None:
                scores = self.met(im, ref, mse=self.want_mse, ssim=self.want_ssim, clip=True)
                scores_out[:, :2].copy_(scores)
                if self.lpips is not None:
                    lp = self.lpips(im, ref, clip=True)
                    scores_out[:, 2].copy_(lp)
            ev = torch.cuda.Event(); ev.record(self.side); self.ev_done[k] = ev

    def flush(self):
        """Enqueue the evaluation half still held back by EVR_EVAL_GATE (the last frame has no successor to wait for)."""
        if self.overlap and self._pending is not None:
            self._enqueue_eval(*self._pending, gated=False)
            self._pending = None

    def _rest_overlapped(self, ref, scores_out, grid, stats):
        k = self.k; self.k = (k + 1) % len(self.imgs)
        main = torch.cuda.current_stream(self.dev)
        img = self.imgs[k]
        if self.ev_done[k] is not None:
            main.wait_event(self.ev_done[k])           # the side stream is done with this buffer (len(imgs) frames ago)
        if self._gate is not None and getattr(self.model, '_gate_owner', None) is not self:
            self.model.set_gate(self._gate_layer, self._gate)      # (another HotPath over the same model took the gate)
            self.model._gate_owner = self
        self.model(grid, stats=stats if self.norm_in else None, out=img)      # (records the gate event inside, if set)
        self.ev_model[k].record(main)
        if self._gate is None:
            self._enqueue_eval(k, ref, scores_out, gated=False)
        else:
            if self._pending is not None:              # the previous frame's evaluation, now that THIS frame's gate is enqueued
                self._enqueue_eval(*self._pending, gated=True)
            self._pending = (k, ref, scores_out)
        return img, scores_out

    def _rest(self, ref, scores_out, grid=None, stats=None):
        grid = self.grid if grid is None else grid
        stats = self.stats if stats is None else stats
        if self.overlap:
            return self._rest_overlapped(ref, scores_out, grid, stats)
        self.model(grid, stats=stats if self.norm_in else None, out=self.img)
        im = self.img.view(self.n, self.H, self.W)
        if self.post != 'none':
            post_process_normalization(im, self.post)
        scores = None
        if ref is not None and (self.want_mse or self.want_ssim or self.lpips is not None):
            scores = self.met(im, ref, mse=self.want_mse, ssim=self.want_ssim, clip=True)
            if scores_out is not None:
                scores_out[:, :2].copy_(scores)
            if self.lpips is not None:
                lp = self.lpips(im, ref, clip=True, out=scores_out[:, 2] if scores_out is not None and scores_out[:, 2].is_contiguous() else None)
                if scores_out is not None and not scores_out[:, 2].is_contiguous():
                    scores_out[:, 2].copy_(lp)
                scores = torch.cat([scores, lp.unsqueeze(1)], dim=1) if scores_out is None else scores_out
        return self.img, scores

"""Host side of the pre/post-processing and metric kernels (same names as the reference)."""
import torch

from . import lib as _lib


def normalize_event_tensor(event_tensor, stats=None):
    """eval.py:398-410, per window, IN PLACE on a cuda tensor [N,B,H,W]."""
    lib = _lib.load()
    assert event_tensor.is_cuda and event_tensor.dtype == torch.float32 and event_tensor.is_contiguous()
    n, B, H, W = event_tensor.shape
    ws = None
    if stats is None:
        ws = torch.empty(n * 32 * 3, dtype=torch.float64, device=event_tensor.device)
    _lib.check(lib.evr_event_tensor_normalize(_lib.ptr(event_tensor), n, B, H, W, _lib.ptr(stats), _lib.ptr(ws),
                                              0 if ws is None else ws.numel() * 8, _lib.stream_ptr()),
               'evr_event_tensor_normalize')
    return event_tensor


def post_process_normalization(img, norm):
    """eval.py:380-395 on cuda tensors [N,H,W] (or [H,W]), in place."""
    if norm == 'none':
        return img
    if norm not in ('robust', 'standard', 'exprobust'):
        raise ValueError(f"Unrecognized normalization argument: {norm}")
    lib = _lib.load()
    assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()
    v = img if img.dim() == 3 else img.unsqueeze(0)
    n, H, W = v.shape
    q = (0.0, 100.0) if norm == 'standard' else (1.0, 99.0)
    ws = torch.empty(max(int(lib.evr_percentile_normalize_workspace_bytes(n, H, W)), 16) // 4, dtype=torch.float32, device=img.device)
    _lib.check(lib.evr_percentile_normalize(_lib.ptr(v), n, H, W, q[0], q[1], 1 if norm == 'exprobust' else 0,
                                            _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()), 'evr_percentile_normalize')
    return img


class Metrics:
    """MSE + SSIM of utils/eval_metrics.py:77-97 (with the [0,1] clip of :253-255) for a batch of frames."""

    def __init__(self):
        self.lib = _lib.load()
        self.ws = None

    def __call__(self, img, ref, mse=True, ssim=True, clip=True):
        assert img.is_cuda and ref.is_cuda and img.shape == ref.shape
        img = img.contiguous(); ref = ref.contiguous()
        v = img if img.dim() == 3 else img.reshape(-1, img.shape[-2], img.shape[-1])
        n, H, W = v.shape
        need = self.lib.evr_metrics_workspace_bytes(n, H, W)
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(need, dtype=torch.uint8, device=img.device)
        out = torch.empty((n, 2), dtype=torch.float64, device=img.device)
        which = (1 if mse else 0) | (2 if ssim else 0)
        _lib.check(self.lib.evr_metrics(_lib.ptr(img), _lib.ptr(ref), n, H, W, which, 1 if clip else 0,
                                        _lib.ptr(out), _lib.ptr(self.ws), self.ws.numel(), _lib.stream_ptr()),
                   'evr_metrics')
        return out


HISTEQ_MODES = {'none': 0, 'global': 1, 'local': 2, 'clahe': 3}
_histeq_ws = {}


def histogram_equalization(img, mode):
    """EvalMetricsTracker.histogram_equalization (utils/eval_metrics.py:326-350) on cuda tensors [N,H,W] already clipped
    to [0,1], in place.  'none' returns the input; an unknown name raises like the reference."""
    if mode not in HISTEQ_MODES:
        raise ValueError(f"Unrecognized histogram equalization argument: {mode}")
    if mode == 'none':
        return img
    lib = _lib.load()
    assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()
    v = img if img.dim() == 3 else img.unsqueeze(0)
    n, H, W = v.shape
    code = HISTEQ_MODES[mode]
    need = lib.evr_hist_equalize_workspace_bytes(n, H, W, code)
    ws = _histeq_ws.get(img.device)
    if need and (ws is None or ws.numel() < need):
        ws = _histeq_ws[img.device] = torch.empty(int(need), dtype=torch.uint8, device=img.device)
    _lib.check(lib.evr_hist_equalize(_lib.ptr(v), n, H, W, code, _lib.ptr(ws) if need else None, int(need),
                                     _lib.stream_ptr()), 'evr_hist_equalize')
    return img

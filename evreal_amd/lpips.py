"""Host side of the LPIPS metric (evr_lpips_*): pyiqa.create_metric('lpips') stand-in on the GPU."""
import ctypes

import numpy as np
import torch

from . import lib as _lib


class LPIPS:
    def __init__(self, state_dict):
        _lib.require_gpu()
        self.lib = _lib.load()
        sd = {k: np.ascontiguousarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float32)
              for k, v in state_dict.items()}
        tensors = (_lib.Tensor * len(sd))()
        self._keep = []
        for i, (k, v) in enumerate(sd.items()):
            name = k.encode(); self._keep.append((name, v))
            tensors[i].name = name
            tensors[i].data_host = v.ctypes.data_as(ctypes.c_void_p)
            tensors[i].ndim = v.ndim
            for d in range(v.ndim):
                tensors[i].shape[d] = v.shape[d]
        h = ctypes.c_void_p()
        _lib.check(self.lib.evr_lpips_create(tensors, len(sd), ctypes.byref(h)), 'evr_lpips_create')
        self.handle = h

    def __call__(self, img, ref, clip=True, out=None):
        """img, ref: cuda fp32 [n,H,W] in [0,1] -> double [n] on device."""
        assert img.is_cuda and ref.is_cuda and img.shape == ref.shape and img.dim() == 3
        img = img.contiguous(); ref = ref.contiguous()
        n, H, W = img.shape
        if out is None:
            out = torch.empty(n, dtype=torch.float64, device=img.device)
        _lib.check(self.lib.evr_lpips_forward(self.handle, _lib.ptr(img), _lib.ptr(ref), n, H, W, 1 if clip else 0,
                                              _lib.ptr(out), _lib.stream_ptr()), 'evr_lpips_forward')
        return out

    def flops(self):
        return float(self.lib.evr_lpips_flops(self.handle))

    def __del__(self):
        try:
            if self.handle is not None:
                self.lib.evr_lpips_destroy(self.handle); self.handle = None
        except Exception:
            pass

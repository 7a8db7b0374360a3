"""Method plugins: the reference's model classes (model/__init__.py:1-4) backed by evr_model_*.

Same contract as the reference (eval.py:163,197,228,230): an object with `.num_encoders`,
`.reset_states()` and `__call__(voxel[N,B,H,W] fp32 on the GPU) -> {'image': [N,1,H,W]}`.
N is the number of independent sequences advanced together (the reference always uses 1).
The zero padding / centre crop of utils/util.py:30-59 happens inside the library, so the caller
may pass either the raw HxW voxel grid or an already padded one (then both are no-ops).
"""
import ctypes

import numpy as np
import torch

from . import lib as _lib

ARCH_UNET, ARCH_FIRENET_LEGACY, ARCH_FIRENET, ARCH_SPADE_E2VID, ARCH_ETNET = 0, 1, 2, 3, 4


def _np_state_dict(state_dict):
    out = {}
    for k, v in state_dict.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if v.dtype.kind != 'f':
            continue            # num_batches_tracked
        out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


class _HipModel:
    arch = None

    def __init__(self):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.handle = None
        self.debug_taps = False      # keep every intermediate readable (read_tensor) -- parity tests only
        self.device = torch.device('cuda', torch.cuda.current_device())
        self._shape = None
        self._needs_reset = True
        self.arith_override = None   # None: the process-wide EVR_ARITH / EVR_FP32; 'fp32' | 'mx' | 'h3' | 'mx6': this model only
        self._exact = None           # exact_twin()

    # -- construction ----------------------------------------------------------------------
    def _desc(self):
        raise NotImplementedError

    def load_state_dict(self, state_dict, strict=True):
        sd = _np_state_dict(state_dict)
        self._sd = sd                 # kept so ColorNet can build its second (half-resolution) executor
        tensors = (_lib.Tensor * len(sd))()
        keep = []
        for i, (k, v) in enumerate(sd.items()):
            name = k.encode()
            keep.append((name, v))
            tensors[i].name = name
            tensors[i].data_host = v.ctypes.data_as(ctypes.c_void_p)
            tensors[i].ndim = v.ndim
            for d in range(v.ndim):
                tensors[i].shape[d] = v.shape[d]
        self.destroy()
        h = ctypes.c_void_p()
        desc = self._desc()
        desc.reserved[0] = 1 if self.debug_taps else 0
        desc.reserved[1] = 1 if getattr(self, 'kwargs', {}).get('use_dynamic_decoder', False) else 0
        desc.reserved[2] = 0 if self.arith_override is None else {'fp32': 0, 'mx': 2, 'h3': 3, 'mx6': 4}[self.arith_override] + 1
        _lib.check(self.lib.evr_model_create(ctypes.byref(desc), tensors, len(sd), ctypes.byref(h)),
                   'evr_model_create')
        self.handle = h
        self._shape = None
        self._needs_reset = True
        return self

    def to(self, device):
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(())

    def destroy(self):
        if self.handle is not None:
            self.lib.evr_model_destroy(self.handle)
            self.handle = None
        twin, self._exact = getattr(self, '_exact', None), None
        if twin is not None:
            twin.destroy()

    def release_shape(self):
        """Return the activations / recurrent state of the last shape to the allocator (weights stay resident; the next call plans
        again).  The caller has synchronised the streams that ran the model."""
        if self.handle is not None:
            _lib.check(self.lib.evr_model_release_shape(self.handle), 'evr_model_release_shape')
        self._shape, self._needs_reset = None, True

    def exact_twin(self):
        """The same network on the library's exact-fp32 HIP kernels (fp32 MFMA, PLAIN tensors: the reference's arithmetic,
        model/submodules.py:227-245, and no range limit) -- what a sequence is re-run on when its activations left the split
        format's range (saturation()).  Built once from the weights this model was loaded with; `self` if it already is exact."""
        if self.arith == 'fp32':
            return self
        if self._exact is None:
            if getattr(self, '_sd', None) is None:
                raise _lib.EvrError("exact_twin: the model has no weights")
            import copy
            twin = copy.copy(self)
            twin.handle, twin._shape, twin._needs_reset, twin._exact = None, None, True, None
            twin.arith_override = 'fp32'
            twin.load_state_dict(self._sd)
            self._exact = twin
        return self._exact

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # -- plugin contract -------------------------------------------------------------------
    def reset_states(self):
        """eval.py:197 -- recurrent state is per sequence; buffers are (re)built lazily."""
        self._needs_reset = True

    def _ensure(self, n, H, W):
        if self.handle is None:
            raise _lib.EvrError("model has no weights: call load_state_dict first")
        if self._needs_reset or self._shape != (n, H, W):
            _lib.check(self.lib.evr_model_reset_states(self.handle, n, H, W, _lib.stream_ptr()),
                       'evr_model_reset_states')
            self._shape = (n, H, W)
            self._needs_reset = False

    def forward(self, voxel, stats=None, out=None):
        assert voxel.is_cuda and voxel.dtype == torch.float32 and voxel.dim() == 4
        voxel = voxel.contiguous()
        n, B, H, W = voxel.shape
        self._ensure(n, H, W)
        if out is None:
            out = torch.empty((n, 1, H, W), dtype=torch.float32, device=voxel.device)
        flags = 1 if stats is not None else 0
        _lib.check(self.lib.evr_model_step(self.handle, _lib.ptr(voxel), _lib.ptr(stats), _lib.ptr(out), flags,
                                           _lib.stream_ptr()), 'evr_model_step')
        return {'image': out}

    __call__ = forward

    # -- parity/debug helpers --------------------------------------------------------------
    def read_tensor(self, name):
        n = ctypes.c_int64(0)
        _lib.check(self.lib.evr_model_read_tensor(self.handle, name.encode(), None, 0, ctypes.byref(n),
                                                  _lib.stream_ptr()), 'evr_model_read_tensor')
        buf = torch.empty(n.value, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.evr_model_read_tensor(self.handle, name.encode(), _lib.ptr(buf), n.value,
                                                  ctypes.byref(n), _lib.stream_ptr()), 'evr_model_read_tensor')
        return buf

    def profile(self, filter=''):
        """Bracket conv launches whose layer name contains `filter` with HIP events (None: off)."""
        _lib.check(self.lib.evr_model_profile_enable(self.handle, None if filter is None else filter.encode()),
                   'evr_model_profile_enable')

    def profile_read(self):
        """-> list of dict(name, ms, flops_per_launch, launches) since profile() was enabled."""
        mx = 128
        names = ctypes.create_string_buffer(mx * 64)
        ms = (ctypes.c_double * mx)(); fl = (ctypes.c_double * mx)()
        ln = (ctypes.c_int64 * mx)(); n = ctypes.c_int(0)
        _lib.check(self.lib.evr_model_profile_read(self.handle, mx, names, ms, fl, ln, ctypes.byref(n),
                                                   _lib.stream_ptr()), 'evr_model_profile_read')
        out = []
        for i in range(n.value):
            nm = names.raw[i * 64:(i + 1) * 64].split(b'\0', 1)[0].decode()
            out.append(dict(name=nm, ms=ms[i], flops_per_launch=fl[i], launches=ln[i]))
        return out

    def set_gate(self, layer, event_handle):
        """Record the HIP event (a handle from evr_event_create) after `layer` in every following step; layer=None: off."""
        _lib.check(self.lib.evr_model_set_gate(self.handle, None if layer is None else layer.encode(), event_handle),
                   'evr_model_set_gate')

    def saturation(self, clear=False):
        """(runs, layer): output runs of the matrix-core layers that left the packed activation format's exact range since
        the counters were cleared, and the layer with most of them ('' when zero).  Synchronises."""
        n = ctypes.c_int64(0)
        name = ctypes.create_string_buffer(64)
        _lib.check(self.lib.evr_model_saturation(self.handle, ctypes.byref(n), name, 64, 1 if clear else 0, _lib.stream_ptr()),
                   'evr_model_saturation')
        return n.value, name.value.decode()

    def saturation_counters(self):
        """Number of range-guard counters evr_model_saturation_async copies (matrix-core layers + the head)."""
        n = ctypes.c_int(0)
        _lib.check(self.lib.evr_model_saturation_async(self.handle, None, 0, ctypes.byref(n), _lib.stream_ptr()), 'evr_model_saturation_async')
        return n.value

    def saturation_async(self, host_pinned):
        """Copy the range-guard counters (cumulative since the last clear) into `host_pinned` (pinned int32/uint32 CPU tensor)
        asynchronously on the current stream: no host synchronisation; read it after an event recorded behind this call."""
        n = ctypes.c_int(0)
        _lib.check(self.lib.evr_model_saturation_async(self.handle, ctypes.c_void_p(host_pinned.data_ptr()), int(host_pinned.numel()),
                                                       ctypes.byref(n), _lib.stream_ptr()), 'evr_model_saturation_async')

    def warn_if_saturated(self, what=''):
        """Once per sequence (eval loops): report activations beyond the split format's range instead of degrading silently."""
        n, layer = self.saturation(clear=True)
        if n:
            mode = self.arith
            how = {'h3': " (clamped at +-4094)", 'mx6': " (beyond +-65504: clamped)"}.get(mode, " (f16 only, 2^-12 relative)")
            print(f"WARNING: {n} activation runs{(' of ' + what) if what else ''} left the exact range of the '{mode}' packed "
                  f"format (most in layer '{layer}'): those values kept reduced precision" + how
                  + "; evreal_amd.eval re-runs such sequences on exact_twin() itself -- a caller stepping the model directly should do "
                  "the same (or set EVR_FP32=1) for data of this magnitude")
        return n

    def flops_per_step(self):
        return float(self.lib.evr_model_flops_per_step(self.handle))

    @property
    def arith(self):
        """'mx6' | 'mx' | 'h3' | 'fp32': the arithmetic this model's convolutions run (EVR_ARITH narrowed to what the layout supports)."""
        return {0: 'fp32', 2: 'mx', 3: 'h3', 4: 'mx6'}.get(int(self.lib.evr_model_arith(self.handle)), '?')


class E2VIDRecurrent(_HipModel):
    """model/model.py:108-144 over model/unet.py:85-143 (E2VID, E2VID+, SSL-E2VID layouts)."""

    def __init__(self, unet_kwargs):
        super().__init__()
        kw = dict(unet_kwargs)
        self.kwargs = kw
        self.num_bins = kw['num_bins']
        self.num_encoders = kw['num_encoders']
        if kw.get('skip_type', 'sum') != 'sum':
            raise _lib.EvrError("only skip_type='sum' exists in the reference (model/unet.py:4,31)")
        if kw.get('norm') not in (None, 'none', 'BN', 'IN'):
            raise _lib.EvrError(f"norm={kw.get('norm')!r} is not supported (BN, IN or none)")
        if kw.get('num_output_channels', 1) != 1:
            raise _lib.EvrError("only num_output_channels=1 (image) is supported")

    def _desc(self):
        kw = self.kwargs
        d = _lib.ModelDesc()
        d.arch = ARCH_UNET
        d.num_bins = kw['num_bins']
        d.base_num_channels = kw.get('base_num_channels', 32)
        d.num_encoders = kw['num_encoders']
        d.num_residual_blocks = kw.get('num_residual_blocks', 2)
        d.kernel_size = kw.get('kernel_size', 5)
        d.norm = {'BN': 1, 'IN': 2}.get(kw.get('norm'), 0)
        d.use_upsample_conv = 1 if kw.get('use_upsample_conv', True) else 0
        d.recurrent_block = 0 if kw.get('recurrent_block_type', 'convlstm') == 'convlstm' else 1
        # getattr(torch, name, None) in model/unet.py:95-96: only 'sigmoid' is used by the reference
        fa = kw.get('final_activation', 'none')
        if fa not in ('none', '', None, 'sigmoid'):
            raise _lib.EvrError(f"final_activation={fa!r} is not supported")
        d.final_activation = 1 if fa == 'sigmoid' else 0
        d.pad_multiple_log2 = kw['num_encoders']
        return d


class FireNet_legacy(_HipModel):
    """model/legacy.py:155-187 (the "FireNet" method): cropper pads to a multiple of 16 because
    num_encoders silently defaults to 4 (legacy.py:127-130)."""

    def __init__(self, config=None, unet_kwargs=None):
        super().__init__()
        cfg = dict(unet_kwargs or config or {})
        self.config = cfg
        self.num_bins = int(cfg['num_bins'])
        self.num_encoders = int(cfg.get('num_encoders', 4))
        if str(cfg.get('recurrent_block_type', 'convgru')) != 'convgru':
            raise _lib.EvrError("FireNet_legacy: only convgru is supported")
        if cfg.get('recurrent_blocks', {'resblock': [0]}) != {'resblock': [0]}:
            raise _lib.EvrError("FireNet_legacy: only recurrent_blocks={'resblock':[0]} is supported")
        if str(cfg.get('norm', 'none')) not in ('none', 'None'):
            raise _lib.EvrError("FireNet_legacy: norm must be 'none'")

    def _desc(self):
        d = _lib.ModelDesc()
        d.arch = ARCH_FIRENET_LEGACY
        d.num_bins = self.num_bins
        d.base_num_channels = int(self.config.get('base_num_channels', 32))
        d.num_residual_blocks = int(self.config.get('num_residual_blocks', 2))
        d.kernel_size = int(self.config.get('kernel_size', 5))
        d.pad_multiple_log2 = self.num_encoders
        return d


class FireNet(_HipModel):
    """model/model.py:147-190 (the "FireNet+" method; eval.py:154-155 forces num_encoders = 0)."""

    def __init__(self, num_bins=5, base_num_channels=16, kernel_size=3):
        super().__init__()
        self.num_bins, self.base_num_channels, self.kernel_size = num_bins, base_num_channels, kernel_size
        self.num_encoders = 0

    def _desc(self):
        d = _lib.ModelDesc()
        d.arch = ARCH_FIRENET
        d.num_bins = self.num_bins
        d.base_num_channels = self.base_num_channels
        d.num_residual_blocks = 2
        d.kernel_size = self.kernel_size
        d.pad_multiple_log2 = self.num_encoders
        return d


class SpadeE2vid(_HipModel):
    """model/spade_e2v.py:113-179 (Unet6, exported as SpadeE2vid in model/__init__.py:4), the 'SPADE-E2VID' method:
    full-resolution ConvLSTM encoder, pixel-shuffle decoders with SPADE normalisation conditioned on the previous
    3-channel reconstruction (first frame: the min/max-normalised first three input channels, rewritten in place),
    image = mean of the three sigmoid outputs.  eval.py:130-133 sets num_encoders = 3 for the cropper."""

    def __init__(self):
        super().__init__()
        self.num_bins = 5
        self.num_encoders = 3

    def _desc(self):
        d = _lib.ModelDesc()
        d.arch = ARCH_SPADE_E2VID
        d.num_bins = 5
        d.base_num_channels = 32
        d.num_encoders = 3
        d.num_residual_blocks = 2
        d.kernel_size = 5
        d.norm = 1
        d.final_activation = 1
        d.pad_multiple_log2 = self.num_encoders
        return d


class EITR(_HipModel):
    """model/eitr/eitr.py:4-16 over model/eitr/u_trans.py:13-123 (the 'ET-Net' method): ConvLSTM encoder, three token
    scales (1/8 resolution; the 1/4 and 1/2 maps through 2x2 / 4x4 patch embeddings) each through a 3-layer pre-norm
    transformer encoder and a 2-layer decoder (8 heads, d = 256, sine position table), the mean of the six token sets,
    three bilinear-upsample decoders with skip sums, sigmoid.  eval.py:152-153 sets num_encoders = 3 for the cropper."""

    def __init__(self, eitr_kwargs):
        super().__init__()
        kw = dict(eitr_kwargs)
        self.kwargs = kw
        self.num_bins = int(kw['num_bins'])
        self.num_encoders = 3
        if kw.get('norm') not in (None, 'none', 'BN', 'IN'):
            raise _lib.EvrError(f"ET-Net with norm={kw.get('norm')!r} is not supported (BN, IN or none)")

    def _desc(self):
        d = _lib.ModelDesc()
        d.arch = ARCH_ETNET
        d.norm = {'BN': 1, 'IN': 2}.get(self.kwargs.get('norm'), 0)
        d.num_bins = self.num_bins
        d.base_num_channels = 32
        d.num_encoders = 3
        d.kernel_size = 5
        d.final_activation = 1
        d.pad_multiple_log2 = self.num_encoders
        return d


class ColorNet:
    """model/model.py:46-105: the event tensor is split into the R,G,B,W Bayer sub-lattices plus the full-resolution
    grayscale stream; every stream is reconstructed by the SAME recurrent network with its own state; the five uint8
    reconstructions are merged into one BGR frame (utils/color_utils.py:53-88).

    Here the four half-resolution colour streams of all sequences advance as extra sequences of one batched executor
    (n_seq = 4N) and the grayscale streams in a second executor that shares the weights; nothing leaves the GPU.
    The reference converts every stream to uint8 on the CPU (5 D2H copies per frame, model.py:100-101)."""

    def __init__(self, model):
        self.model = model                       # full-resolution (grayscale) executor
        if getattr(model, '_sd', None) is None:
            raise _lib.EvrError("ColorNet needs a model with loaded weights")
        import copy
        self.half = copy.copy(model)             # shallow: same kwargs/desc (and arithmetic), own handle
        self.half.handle = None
        self.half._shape = None
        self.half._exact = None
        self.half.load_state_dict(model._sd)
        self._exact = None
        self.lib = _lib.load()
        import os
        self._side = torch.cuda.Stream(device=model.device) if os.environ.get('EVR_COLOR_STREAMS', '2') != '1' else None
        self.reset_states()

    @property
    def num_encoders(self):
        return self.model.num_encoders

    @property
    def arith(self):
        return self.model.arith

    def saturation(self, clear=False):
        """Range-guard counters of both executors (see _HipModel.saturation)."""
        n0, l0 = self.model.saturation(clear)
        n1, l1 = self.half.saturation(clear)
        return n0 + n1, (l0 if n0 >= n1 else l1)

    def exact_twin(self):
        """ColorNet over the exact-fp32 twin of the base network (both executors)."""
        if self.arith == 'fp32':
            return self
        if getattr(self, '_exact', None) is None:
            self._exact = ColorNet(self.model.exact_twin())
        return self._exact

    def reset_states(self):
        self.model.reset_states()
        self.half.reset_states()

    def forward(self, event_tensor, stats=None):
        """event_tensor [N,B,H,W] (H, W even) -> {'image': uint8 BGR [N,H,W,3] on the GPU, 'planes': [N,4,H/2,W/2],
        'gray': [N,1,H,W]} (planes/gray are the float reconstructions before the uint8 truncation)."""
        assert event_tensor.is_cuda and event_tensor.dim() == 4
        ev = event_tensor.contiguous()
        n, B, H, W = ev.shape
        split = torch.empty((4 * n, B, H // 2, W // 2), dtype=torch.float32, device=ev.device)
        _lib.check(self.lib.evr_bayer_split(_lib.ptr(ev), n, B, H, W, _lib.ptr(split), _lib.stream_ptr()), 'evr_bayer_split')
        # the four half-resolution streams and the full-resolution stream are independent: two HIP streams, so that the deep
        # (small) layers of one executor run beside the other's (EVR_COLOR_STREAMS=1: one after the other)
        main = torch.cuda.current_stream(ev.device)
        if self._side is None:
            planes = self.half(split)['image'].view(n, 4, H // 2, W // 2)
            gray = self.model(ev)['image']
        else:
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                planes = self.half(split)['image'].view(n, 4, H // 2, W // 2)
            split.record_stream(self._side)
            gray = self.model(ev)['image']
            main.wait_stream(self._side)
            planes.record_stream(main)
        bgr = torch.empty((n, H, W, 3), dtype=torch.uint8, device=ev.device)
        _lib.check(self.lib.evr_color_merge(_lib.ptr(planes), _lib.ptr(gray), n, H, W, _lib.ptr(bgr), _lib.stream_ptr()),
                   'evr_color_merge')
        return {'image': bgr, 'planes': planes, 'gray': gray}

    __call__ = forward

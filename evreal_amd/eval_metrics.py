"""EvalMetricsTracker of the reference (utils/eval_metrics.py:162-350) with the metric arithmetic on the GPU.

Output files are byte-compatible with the reference (utils/eval_utils.py:57-84): `timestamps.txt`
("{idx} {ts:.15f}"), `<metric>.txt` / `event_rate.txt` ("{idx} {score:.5f}"), `frame_%010d.png`
(round(img*255) uint8) -- the formats analyze_robustness.py:55-65,109-110 and downstream_tasks/ parse.
Frames arrive in batches; gating (start/end time, |ref_ts-img_ts| <= ts_tol_ms, not color) is per frame.

Metric plug-ins keep the reference's contract (utils/eval_metrics.py:18-75): a `BaseMetric` subclass with `name`,
`no_ref`, `calculate(img, ref) -> float | list`, `finish_queue()`, `reset()`.  Three kinds live side by side:
  * 'mse', 'ssim' and -- when a weights file is available -- 'lpips' run batched on the GPU (evr_metrics / evr_lpips_*);
  * anything registered with `register_metric(name, factory)` runs per frame on host arrays, exactly like the reference's
    MseMetric / SsimMetric (clipped float32 [H,W] images in, a float or a list of floats out);
  * any other name is looked up in pyiqa.list_models() when pyiqa is importable (queued in batches of 4 as
    PyIqaMetricFactory does, :100-156); without pyiqa it is reported as "Unknown metric", as the reference does for a
    name it does not know (:203).
"""
import math
import os
import traceback
import warnings
from os.path import join

import numpy as np
import torch

from .prepost import Metrics, histogram_equalization

GPU_METRICS = ('mse', 'ssim')
LPIPS_WEIGHTS_ENV = 'EVREAL_LPIPS_WEIGHTS'      # path to a pyiqa/lpips AlexNet-v0.1 state_dict (torch.save'd)


def _load_lpips():
    """The reference lets pyiqa download the LPIPS weights; offline they must be supplied as a file."""
    path = os.environ.get(LPIPS_WEIGHTS_ENV, os.path.join('pretrained', 'lpips_alex.pth'))
    if not os.path.exists(path):
        return None
    from .lpips import LPIPS
    return LPIPS(torch.load(path, map_location='cpu', weights_only=False))


class BaseMetric:
    """Base class for quantitative evaluation metrics -- the plug-in contract of utils/eval_metrics.py:18-75."""

    def __init__(self, name, no_ref=False):
        self.scores = []
        self.name = name
        self.no_ref = no_ref
        self.updated = 0
        self.image_queue = []
        self.ref_queue = []
        self.batch_size = 4

    def reset(self):
        self.scores, self.image_queue, self.ref_queue, self.updated = [], [], [], 0

    def finish_queue(self):
        self.updated = 0

    def get_num_updated(self):
        return self.updated

    def calculate(self, img, ref):
        raise NotImplementedError

    def update(self, img, ref=None):
        self.updated = 0
        score = self.calculate(img, ref)
        self.add(score if isinstance(score, list) else [score])

    def add(self, values):
        """Keep the finite scores (utils/eval_metrics.py:49-53)."""
        self.updated = 0
        for s in values:
            s = float(s)
            if math.isfinite(s) and not math.isnan(s):
                self.updated += 1
                self.scores.append(s)

    def get_num_scores(self):
        return len(self.scores)

    def get_all_scores(self):
        return self.scores

    def get_last_score(self):
        return self.scores[-1]

    def get_last_scores(self, n):
        return self.scores[-n:]

    def get_mean_score(self):
        return -1 if not self.scores else sum(self.scores) / len(self.scores)

    def get_name(self):
        return self.name


class GpuMetric(BaseMetric):
    """'mse' / 'ssim' / 'lpips': the tracker computes a whole batch with one launch and hands the scores to add()."""
    on_gpu = True

    def calculate(self, img, ref):
        raise RuntimeError(f"{self.name} is computed in batches by EvalMetricsTracker.update_batch")


_REGISTRY = {}


def register_metric(name, factory):
    """Make `name` available to -qm / quan_eval_metric_names: factory() -> BaseMetric (per-frame, host arrays)."""
    _REGISTRY[name] = factory


class PyIqaMetricFactory:
    """Any pyiqa metric by name (utils/eval_metrics.py:100-156), when pyiqa can be imported: gray frames are replicated
    to 3 channels (cv2torch(num_ch=3), eval_utils.py:46-54), queued, and scored four at a time; the queue's tail is
    flushed by finish_queue()."""

    def __init__(self):
        try:
            import pyiqa
        except Exception:
            pyiqa = None
        self.pyiqa = pyiqa
        self.list_of_metrics = list(pyiqa.list_models()) if pyiqa is not None else []
        self.created_metrics = {}

    def get_metric(self, name):
        if name in self.created_metrics:
            return self.created_metrics[name]
        with warnings.catch_warnings():
            warnings.filterwarnings("ignore", category=UserWarning)
            iqa_metric = self.pyiqa.create_metric(name)
        metric = _QueuedIqaMetric(name.lower(), iqa_metric.metric_mode == 'NR', iqa_metric)
        self.created_metrics[name] = metric
        return metric


def _as_batch(image, num_ch=3):
    t = torch.as_tensor(np.asarray(image))
    if t.dim() == 2:
        t = t.unsqueeze(0)
        if num_ch > 1:
            t = t.repeat(num_ch, 1, 1)
    return t.unsqueeze(0) if t.dim() == 3 else t


class _QueuedIqaMetric(BaseMetric):
    def __init__(self, name, no_ref, fn):
        super().__init__(name, no_ref)
        self.fn = fn

    def inference(self):
        if not self.image_queue:
            return []
        imgs = torch.cat(self.image_queue[-self.batch_size:])
        if self.ref_queue[0] is not None:
            scores = self.fn(imgs, torch.cat(self.ref_queue[-self.batch_size:]))
        else:
            scores = self.fn(imgs)
        self.image_queue, self.ref_queue = [], []
        out = scores.squeeze().tolist()
        return out if isinstance(out, list) else [out]

    def finish_queue(self):
        self.updated = 0
        score = self.inference()
        self.updated += len(score)
        self.scores.extend(score)

    def calculate(self, img, ref=None):
        self.image_queue.append(_as_batch(img))
        self.ref_queue.append(None if ref is None else _as_batch(ref))
        return [] if len(self.image_queue) < self.batch_size else self.inference()


_pyiqa_factory = None


def pyiqa_metric_factory():
    global _pyiqa_factory
    if _pyiqa_factory is None:
        _pyiqa_factory = PyIqaMetricFactory()
    return _pyiqa_factory


class EvalMetricsTracker:
    def __init__(self, save_images=False, save_processed_images=False, output_dir=None, hist_eq='none',
                 quan_eval_metric_names=None, quan_eval_start_time=0, quan_eval_end_time=float('inf'),
                 quan_eval_ts_tol_ms=float('inf'), has_reference_frames=False, color=False):
        if quan_eval_metric_names is None:
            quan_eval_metric_names = ['mse', 'ssim', 'lpips']
        if hist_eq not in ('none', 'global', 'local', 'clahe'):
            raise ValueError(f"Unrecognized histogram equalization argument: {hist_eq}")
        self.save_images, self.output_dir, self.hist_eq = save_images, output_dir, hist_eq
        self.save_processed_images = save_processed_images
        if self.hist_eq == 'none' and self.save_processed_images:
            print("Can not save processed images when hist_eq is none")
            self.save_processed_images = False
        self._pending = []
        self._files = {}
        self.start, self.end, self.tol_ms = quan_eval_start_time, quan_eval_end_time, quan_eval_ts_tol_ms
        self.has_reference_frames, self.color = has_reference_frames, color
        self.quan_eval_indices = []
        self.metrics = []
        for name in quan_eval_metric_names:
            if name in GPU_METRICS:
                self.metrics.append(GpuMetric(name))
            elif name == 'lpips' and self._lpips_model() is not None:
                self.metrics.append(GpuMetric(name))
            elif name in _REGISTRY:
                self.metrics.append(_REGISTRY[name]())
            elif name in pyiqa_metric_factory().list_of_metrics:
                self.metrics.append(pyiqa_metric_factory().get_metric(name))
            else:
                print("Unknown metric " + name)     # utils/eval_metrics.py:203
        if not self.has_reference_frames:
            self.metrics = [m for m in self.metrics if m.no_ref]
        self.only_no_ref = all(m.no_ref for m in self.metrics)
        self._gpu = Metrics()
        self.reset()

    _lpips_cache = [False, None]

    @classmethod
    def _lpips_model(cls):
        if not cls._lpips_cache[0]:
            cls._lpips_cache = [True, _load_lpips()]
            if cls._lpips_cache[1] is None:
                print(f"lpips: no weights at ${LPIPS_WEIGHTS_ENV} or pretrained/lpips_alex.pth (pyiqa downloads them; "
                      "offline they must be provided) -> falling back to pyiqa if it is installed")
        return cls._lpips_cache[1]

    # -- files --------------------------------------------------------------------------------
    def reset(self):
        os.makedirs(self.output_dir, exist_ok=True)
        if self.save_processed_images:
            self.processed_output_dir = self.output_dir + "_processed"
            os.makedirs(self.processed_output_dir, exist_ok=True)
        self._close_files()
        open(join(self.output_dir, 'timestamps.txt'), 'w', encoding="utf-8").close()
        for m in self.metrics:
            open(join(self.output_dir, m.name + '.txt'), 'w', encoding="utf-8").close()
            m.reset()

    # Text outputs: the reference re-opens a file for every line (eval_utils.py:57-69).  Here a tracker keeps its files
    # open in append mode and writes a batch of lines at a time; finalize() closes them.  Same bytes, no per-frame
    # open/close on the critical path.
    def _file(self, path):
        f = self._files.get(path)
        if f is None:
            f = self._files[path] = open(path, 'a', encoding="utf-8")
        return f

    def _close_files(self):
        for f in self._files.values():
            f.close()
        self._files = {}

    def _append(self, path, pairs, fmt='{} {:.5f}\n'):
        f = self._file(path)
        f.write(''.join(fmt.format(a, b) for a, b in pairs))

    def save_custom_metric(self, idx, metric_name, metric_value, is_int=False):
        path = join(self.output_dir, metric_name + '.txt')
        if idx == 0:      # the reference only truncates on idx 0 (eval_metrics.py:277-278)
            old = self._files.pop(path, None)
            if old is not None:
                old.close()
            open(path, 'w', encoding="utf-8").close()
        self._append(path, [(idx, metric_value)], '{} {}\n' if is_int else '{} {:.5f}\n')

    def save_new_scores(self, metric):
        """eval_metrics.py:217-223: the scores a metric produced since its last update, against the LAST evaluated indices
        (a queued metric returns nothing until its batch is full, then `batch_size` scores at once)."""
        n = metric.get_num_updated()
        if n > 0:
            self._append(join(self.output_dir, metric.name + '.txt'),
                         zip(self.quan_eval_indices[-n:], metric.get_last_scores(n)))

    # -- per batch ----------------------------------------------------------------------------
    def wants_precomputed(self):
        """Names of the GPU metrics a frame loop may compute for a whole chunk itself and hand to update_batch(scores=...):
        only when no histogram equalisation stands between the frames and the scores."""
        if self.hist_eq != 'none' or self.color or not self.has_reference_frames:
            return []
        return [m.name for m in self.metrics if getattr(m, 'on_gpu', False)]

    def update_batch(self, indices, imgs, refs, img_ts, ref_ts, scores=None, u8=None):
        """indices: dataset indices; imgs [n,H,W] cuda (unclipped); refs [n,H,W] cuda or None;
        img_ts / ref_ts: python floats per frame (ref_ts None -> img_ts).
        scores: optional {metric name: numpy [n]} already computed for EVERY frame of this call (see wants_precomputed);
        u8: optional numpy uint8 [n,H,W] = round(clip(imgs)*255), already on the host, for the PNG writers."""
        try:
            self._update_batch(indices, imgs, refs, img_ts, ref_ts, scores, u8)
        finally:
            for f in self._files.values():      # one write per file per batch reaches the OS even if a later frame raises
                f.flush()

    def _update_batch(self, indices, imgs, refs, img_ts, ref_ts, pre=None, u8=None):
        n = len(indices)
        if ref_ts is None:
            ref_ts = img_ts
        self._append(join(self.output_dir, 'timestamps.txt'), zip(indices, img_ts), '{} {:.15f}\n')
        if self.save_images:
            self._save_pngs(self.output_dir, indices, imgs, u8)    # clipped inside; BEFORE hist-eq (eval_metrics.py:257-258)
        sel = []
        for j in range(n):
            inside = self.start <= img_ts[j] <= self.end
            tol_ok = abs(ref_ts[j] - img_ts[j]) * 1000 <= self.tol_ms or self.only_no_ref
            if inside and tol_ok and not self.color:
                sel.append(j)
        need_proc = self.hist_eq != 'none' and (self.save_processed_images or (sel and self.metrics))
        if need_proc:
            # histogram equalisation of the clipped frames (eval_metrics.py:260-265); `none` leaves the clip to the kernels
            imgs = histogram_equalization(torch.clamp(imgs, 0.0, 1.0).contiguous(), self.hist_eq)
            if refs is not None and self.has_reference_frames:
                refs = histogram_equalization(torch.clamp(refs, 0.0, 1.0).contiguous(), self.hist_eq)
            if self.save_processed_images:
                self._save_pngs(self.processed_output_dir, indices, imgs)
        if not sel or not self.metrics:
            self.quan_eval_indices.extend(indices[j] for j in sel)
            return
        idxs = [indices[j] for j in sel]
        gpu = [m for m in self.metrics if getattr(m, 'on_gpu', False)]
        host = [m for m in self.metrics if not getattr(m, 'on_gpu', False)]
        have_pre = pre is not None and not need_proc and all(m.name in pre for m in gpu)
        isel = rsel = None
        if host or (gpu and not have_pre):
            js = torch.tensor(sel, device=imgs.device)
            isel = imgs[js].contiguous()
            rsel = refs[js].contiguous() if refs is not None else None
        if gpu:
            want = {m.name for m in gpu}
            scores = lp = None
            if not have_pre:
                if want & set(GPU_METRICS):
                    scores = self._gpu(isel, rsel, mse='mse' in want, ssim='ssim' in want, clip=True).cpu().numpy()
                lp = self._lpips_model()(isel, rsel, clip=True).cpu().numpy() if 'lpips' in want else None
            for m in gpu:
                if have_pre:
                    col = np.asarray(pre[m.name])[sel]
                else:
                    col = scores[:, 0] if m.name == 'mse' else scores[:, 1] if m.name == 'ssim' else lp
                m.add(col)
                self._append(join(self.output_dir, m.name + '.txt'),
                             [(i, float(s)) for i, s in zip(idxs, col) if math.isfinite(s)])
        if not host:
            self.quan_eval_indices.extend(idxs)
            return
        # plug-in metrics: clipped host arrays, frame by frame, exactly as update_quantitative_metrics (:230-242)
        himg = torch.clamp(isel, 0.0, 1.0).cpu().numpy()
        href = torch.clamp(rsel, 0.0, 1.0).cpu().numpy() if (rsel is not None and self.has_reference_frames) else None
        for k, idx in enumerate(idxs):
            self.quan_eval_indices.append(idx)
            for m in host:
                try:
                    if not self.has_reference_frames or m.no_ref:
                        m.update(himg[k])
                    else:
                        m.update(himg[k], href[k])
                    self.save_new_scores(m)
                except Exception as e:
                    print("Exception in metric " + m.get_name() + ": " + str(e))
                    print(traceback.format_exc())
                    m.reset()

    def update_batch_color(self, indices, bgr_u8, img_ts):
        """Colour frames (uint8 BGR [n,H,W,3] on the GPU): timestamps + PNGs only -- the reference skips every
        quantitative metric in colour mode (utils/eval_metrics.py:272)."""
        self._append(join(self.output_dir, 'timestamps.txt'), zip(indices, img_ts), '{} {:.15f}\n')
        if self.save_images:
            rgb = bgr_u8.flip(-1).contiguous().cpu()     # cv2.imwrite stores BGR arrays as RGB files
            if self._use_pil():
                for i, a in zip(indices, rgb.numpy()):
                    self._submit_png(join(self.output_dir, 'frame_{:010d}.png'.format(i)), a, 'RGB')
            else:
                self._submit_native(self.output_dir, indices, rgb, 3)

    # PNG encoding leaves the frame loop.  Round 6: a pool of NATIVE writer threads inside the library
    # (csrc/hostcodec.cpp: filter 0 + one zlib stream of level EVREAL_PNG_LEVEL, default 1; no GIL, ~0.3 ms per 346x260 frame and
    # core) takes whole chunks of frames per call; the files decode to exactly round(clip(img) * 255) (eval_utils.py:80-84).
    # EVREAL_PNG_THREADS=<n> sizes the pool (0: every call waits for its files -- synchronous, as the reference);
    # EVREAL_PNG_WRITER=pil keeps round 5's pool of PIL writers (~3 ms per frame and core, zlib level 6: smaller files).
    # finalize() waits for the files (and raises a writer's failure there).
    _pool = None
    _native = None

    @classmethod
    def _png_threads(cls):
        return int(os.environ.get('EVREAL_PNG_THREADS', '') or min(16, max(4, (os.cpu_count() or 4) // 2)))

    @classmethod
    def _writer_pool(cls):
        if cls._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            cls._pool = ThreadPoolExecutor(max_workers=max(1, cls._png_threads()))
        return cls._pool

    @classmethod
    def _native_pool(cls):
        if cls._native is None:
            import ctypes
            from . import lib as _lib
            h = ctypes.c_void_p()
            _lib.check(_lib.load().evr_png_pool_create(max(1, cls._png_threads()), int(os.environ.get('EVREAL_PNG_LEVEL', '1')), ctypes.byref(h)),
                       'evr_png_pool_create')
            cls._native = h
        return cls._native

    @staticmethod
    def _use_pil():
        return os.environ.get('EVREAL_PNG_WRITER', 'native') == 'pil'

    def _submit_png(self, path, array, mode):
        def job():
            from PIL import Image
            Image.fromarray(array, mode=mode).save(path)
        if os.environ.get('EVREAL_PNG_THREADS', '') == '0':
            job()                                            # synchronous, as the reference
        else:
            self._pending.append(self._writer_pool().submit(job))

    def _submit_native(self, folder, indices, frames, channels):
        """frames: uint8 host array/tensor [n, H, W] or [n, H, W, 3]; frame 0 contiguous, frames a constant stride apart."""
        import ctypes
        from . import lib as _lib
        t = frames if isinstance(frames, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frames))
        n, H, W = int(t.shape[0]), int(t.shape[1]), int(t.shape[2])
        if n == 0:
            return
        if not t[0].is_contiguous() or (n > 1 and t.stride(0) < H * W * channels):
            t = t.contiguous()
        idx = (ctypes.c_int64 * n)(*[int(i) for i in indices])
        _lib.check(_lib.load().evr_png_pool_submit(self._native_pool(), os.fsencode(folder), idx, n, ctypes.c_void_p(t.data_ptr()), H, W, channels,
                                                  int(t.stride(0)) if n > 1 else 0), 'evr_png_pool_submit')
        self._native_pending = True
        if os.environ.get('EVREAL_PNG_THREADS', '') == '0':
            self._wait_native()

    def _wait_native(self, ignore_errors=False):
        if getattr(self, '_native_pending', False) and type(self)._native is not None:
            from . import lib as _lib
            self._native_pending = False
            rc = _lib.load().evr_png_pool_wait(type(self)._native, None)
            if rc and not ignore_errors:
                _lib.check(rc, 'evr_png_pool_wait')

    def _save_pngs(self, folder, indices, imgs, u8=None):
        if u8 is None:
            u8 = torch.round(torch.clamp(imgs, 0.0, 1.0) * 255).to(torch.uint8).cpu()               # eval_utils.py:83
        if self._use_pil():
            for i, a in zip(indices, u8.numpy() if isinstance(u8, torch.Tensor) else u8):
                self._submit_png(join(folder, 'frame_{:010d}.png'.format(i)), np.ascontiguousarray(a), 'L')
        else:
            self._submit_native(folder, indices, u8, 1)

    def finalize(self, idx, wait_png=True):
        """eval_metrics.py:225-228: flush the queued metrics; then wait for this tracker's files.  wait_png=False (the drop-in's
        sequence loops): the native writers keep working on this tracker's last frames while the next sequence starts; the caller
        owes a wait_all_pngs() before it reports the dataset (evreal_amd.eval does it per dataset) -- 3 ms per 160-frame sequence."""
        for m in self.metrics:
            if getattr(m, 'on_gpu', False):
                m.updated = 0
                continue
            try:
                m.finish_queue()
                self.save_new_scores(m)
            except Exception as e:
                print("Exception in metric " + m.get_name() + ": " + str(e))
                m.reset()
        for f in self._pending:
            f.result()                                       # re-raises a writer's exception here
        self._pending = []
        if wait_png:
            self._wait_native()                              # ... and a native writer's failure here
        self._close_files()

    @classmethod
    def wait_all_pngs(cls):
        """Every frame handed to the native writers so far is on disk when this returns; raises the first write failure."""
        if cls._native is not None:
            from . import lib as _lib
            _lib.check(_lib.load().evr_png_pool_wait(cls._native, None), 'evr_png_pool_wait')

    def discard(self):
        """Abandon this tracker (its sequence is re-run from the start with a fresh one, which truncates the same files): wait for
        the PNG writers still holding its frames -- they must not race the re-run's writes to the same paths -- and close the files."""
        for f in self._pending:
            try:
                f.result()
            except Exception:
                pass
        self._pending = []
        self._wait_native(ignore_errors=True)
        self._close_files()

    def get_num_quan_evaluations(self):
        return len(self.quan_eval_indices)

    def get_mean_scores(self):
        return {m.name: m.get_mean_score() for m in self.metrics}


class MetricTracker:
    """eval.py:249-276."""

    def __init__(self):
        self.data_dict = {}

    def init_key(self, key):
        self.data_dict[key] = {'total': 0.0, 'count': 0, 'average': 0.0}

    def update(self, key, value, count=1):
        if count == 0:
            return
        if key not in self.data_dict:
            self.init_key(key)
        d = self.data_dict[key]
        d['total'] += value * count
        d['count'] += count
        d['average'] = d['total'] / d['count']

    def get_average(self, key):
        if key not in self.data_dict:
            self.init_key(key)
        return self.data_dict[key]['average']

    def get_count(self, key):
        if key not in self.data_dict:
            self.init_key(key)
        return self.data_dict[key]['count']

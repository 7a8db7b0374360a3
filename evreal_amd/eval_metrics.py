"""EvalMetricsTracker of the reference (utils/eval_metrics.py:162-350) with the metric arithmetic on the GPU.

Output files are byte-compatible with the reference (utils/eval_utils.py:57-84): `timestamps.txt`
("{idx} {ts:.15f}"), `<metric>.txt` / `event_rate.txt` ("{idx} {score:.5f}"), `frame_%010d.png`
(round(img*255) uint8) -- the formats analyze_robustness.py:55-65,109-110 and downstream_tasks/ parse.
Frames arrive in batches; gating (start/end time, |ref_ts-img_ts| <= ts_tol_ms, not color) is per frame.
"""
import math
import os
from os.path import join

import numpy as np
import torch

from .prepost import Metrics

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
    """Score bookkeeping of utils/eval_metrics.py:18-75 (finite scores only; mean or -1)."""

    def __init__(self, name):
        self.name, self.scores, self.updated, self.no_ref = name, [], 0, False

    def reset(self):
        self.scores, self.updated = [], 0

    def add(self, values):
        self.updated = 0
        for s in values:
            s = float(s)
            if math.isfinite(s) and not math.isnan(s):
                self.updated += 1
                self.scores.append(s)

    def get_mean_score(self):
        return -1 if not self.scores else sum(self.scores) / len(self.scores)


class EvalMetricsTracker:
    def __init__(self, save_images=False, save_processed_images=False, output_dir=None, hist_eq='none',
                 quan_eval_metric_names=None, quan_eval_start_time=0, quan_eval_end_time=float('inf'),
                 quan_eval_ts_tol_ms=float('inf'), has_reference_frames=False, color=False):
        if quan_eval_metric_names is None:
            quan_eval_metric_names = ['mse', 'ssim', 'lpips']
        if hist_eq != 'none':
            raise NotImplementedError(f"histeq={hist_eq!r}: every shipped eval config uses 'none' (SURVEY 8f-4)")
        self.save_images, self.output_dir, self.hist_eq = save_images, output_dir, hist_eq
        self._pending = []
        self.save_processed_images = False
        self.start, self.end, self.tol_ms = quan_eval_start_time, quan_eval_end_time, quan_eval_ts_tol_ms
        self.has_reference_frames, self.color = has_reference_frames, color
        self.quan_eval_indices = []
        self.metrics = []
        for name in quan_eval_metric_names:
            if name in GPU_METRICS:
                self.metrics.append(BaseMetric(name))
            elif name == 'lpips' and (self._lpips_model() is not None):
                self.metrics.append(BaseMetric(name))
            else:
                print("Unknown metric " + name)     # utils/eval_metrics.py:203 (LPIPS/pyiqa: not built yet)
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
                      "offline they must be provided) -> metric skipped")
        return cls._lpips_cache[1]

    # -- files --------------------------------------------------------------------------------
    def reset(self):
        os.makedirs(self.output_dir, exist_ok=True)
        open(join(self.output_dir, 'timestamps.txt'), 'w', encoding="utf-8").close()
        for m in self.metrics:
            open(join(self.output_dir, m.name + '.txt'), 'w', encoding="utf-8").close()
            m.reset()

    @staticmethod
    def _append(path, pairs, fmt='{} {:.5f}\n'):
        with open(path, 'a', encoding="utf-8") as f:
            for a, b in pairs:
                f.write(fmt.format(a, b))

    def save_custom_metric(self, idx, metric_name, metric_value):
        path = join(self.output_dir, metric_name + '.txt')
        if idx == 0:
            open(path, 'w', encoding="utf-8").close()   # the reference only truncates on idx 0 (eval_metrics.py:277-278)
        self._append(path, [(idx, metric_value)])

    # -- per batch ----------------------------------------------------------------------------
    def update_batch(self, indices, imgs, refs, img_ts, ref_ts):
        """indices: dataset indices; imgs [n,H,W] cuda (unclipped); refs [n,H,W] cuda or None;
        img_ts / ref_ts: python floats per frame (ref_ts None -> img_ts)."""
        n = len(indices)
        if ref_ts is None:
            ref_ts = img_ts
        self._append(join(self.output_dir, 'timestamps.txt'), zip(indices, img_ts), '{} {:.15f}\n')
        if self.save_images:
            self._save_pngs(indices, imgs)
        sel = []
        for j in range(n):
            inside = self.start <= img_ts[j] <= self.end
            tol_ok = abs(ref_ts[j] - img_ts[j]) * 1000 <= self.tol_ms or self.only_no_ref
            if inside and tol_ok and not self.color:
                sel.append(j)
        if not sel or not self.metrics:
            self.quan_eval_indices.extend(indices[j] for j in sel)
            return
        js = torch.tensor(sel, device=imgs.device)
        want = {m.name for m in self.metrics}
        scores = self._gpu(imgs[js].contiguous(), refs[js].contiguous(), mse='mse' in want, ssim='ssim' in want,
                           clip=True).cpu().numpy()
        idxs = [indices[j] for j in sel]
        self.quan_eval_indices.extend(idxs)
        lp = None
        if 'lpips' in want:
            lp = self._lpips_model()(imgs[js].contiguous(), refs[js].contiguous(), clip=True).cpu().numpy()
        for m in self.metrics:
            col = scores[:, 0] if m.name == 'mse' else scores[:, 1] if m.name == 'ssim' else lp
            finite = [(i, float(s)) for i, s in zip(idxs, col) if math.isfinite(s)]
            m.add(col)
            self._append(join(self.output_dir, m.name + '.txt'), finite)

    def update_batch_color(self, indices, bgr_u8, img_ts):
        """Colour frames (uint8 BGR [n,H,W,3] on the GPU): timestamps + PNGs only -- the reference skips every
        quantitative metric in colour mode (utils/eval_metrics.py:272)."""
        self._append(join(self.output_dir, 'timestamps.txt'), zip(indices, img_ts), '{} {:.15f}\n')
        if self.save_images:
            rgb = bgr_u8.flip(-1).cpu().numpy()          # cv2.imwrite stores BGR arrays as RGB files
            for i, a in zip(indices, rgb):
                self._submit_png(join(self.output_dir, 'frame_{:010d}.png'.format(i)), a, 'RGB')

    # PNG encoding (zlib, ~3 ms per 346x260 frame on one core) leaves the frame loop: a small shared thread pool
    # encodes and writes while the GPU goes on (PIL releases the GIL in the compressor); same PIL calls, so the
    # files are byte-identical to the synchronous ones.  finalize() waits for this tracker's files.
    _pool = None

    @classmethod
    def _writer_pool(cls):
        if cls._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            cls._pool = ThreadPoolExecutor(max_workers=int(os.environ.get('EVREAL_PNG_THREADS', '4')))
        return cls._pool

    def _submit_png(self, path, array, mode):
        def job():
            from PIL import Image
            Image.fromarray(array, mode=mode).save(path)
        if os.environ.get('EVREAL_PNG_THREADS', '') == '0':
            job()                                            # synchronous, as the reference
        else:
            self._pending.append(self._writer_pool().submit(job))

    def _save_pngs(self, indices, imgs):
        u8 = torch.round(torch.clamp(imgs, 0.0, 1.0) * 255).to(torch.uint8).cpu().numpy()
        for i, a in zip(indices, u8):
            self._submit_png(join(self.output_dir, 'frame_{:010d}.png'.format(i)), a, 'L')

    def finalize(self, idx):
        for f in self._pending:
            f.result()                                       # re-raises a writer's exception here
        self._pending = []
        for m in self.metrics:
            m.updated = 0

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

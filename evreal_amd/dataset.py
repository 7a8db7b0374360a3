"""Sequence reader + window planner: MemMapDataset of the reference (dataset.py:14-294) with the
events resident in HBM and the tensorizer on the GPU.

Same constructor, same length/indices/timestamps/`dt`/frame choice per item (including the reference's
quirks, SURVEY 8a list), same item dict -- but `item['events']` / `item['frame']` are CUDA tensors, and
`voxel_batch` turns many windows into voxel grids with one launch.  Per-item bookkeeping is computed
for the whole sequence up front with vectorised numpy (the reference does it item by item in Python).
"""
import os
from bisect import bisect_left

import numpy as np
import torch

from . import lib as _lib
from .config import read_json
from .voxel import Voxelizer


class SequenceFiles:
    """The memory-mapped columns of one sequence directory (tools/bag_to_npy.py:53-94 writes them):
    events_ts [N] f64, events_xy [N,2] int, events_p [N] {0,1}; optionally images [F,H,W,1] u8, images_ts [F,1] f64,
    image_event_indices [F,1] int64 (all three or none)."""
    EVENT_FILES = ('events_ts.npy', 'events_xy.npy', 'events_p.npy')
    FRAME_FILES = ('images.npy', 'images_ts.npy', 'image_event_indices.npy')

    def __init__(self, path, t, xy, p, images=None, frame_stamps=None, image_event_indices=None):
        self.path, self.t, self.xy, self.p = path, t, xy, p
        self.images, self.frame_stamps, self.image_event_indices = images, frame_stamps, image_event_indices
        if not (len(t) == len(xy) == len(p)):
            raise AssertionError("Number of events, timestamps and coordinates do not match")
        self.num_events = len(t)

    @classmethod
    def open(cls, path):
        at = lambda n: os.path.join(path, n)
        t, xy, p = (np.load(at(n), mmap_mode='r').squeeze() for n in cls.EVENT_FILES)
        if all(os.path.exists(at(n)) for n in cls.FRAME_FILES):
            return cls(path, t, xy, p, np.load(at('images.npy'), mmap_mode='r'), np.load(at('images_ts.npy')),
                       np.load(at('image_event_indices.npy')))
        return cls(path, t, xy, p)

    def metadata_resolution(self):
        meta = os.path.join(self.path, 'metadata.json')
        return read_json(meta)["sensor_resolution"] if os.path.exists(meta) else None


class MemMapDataset:
    def __init__(self, data_path, sensor_resolution=None, num_bins=5, voxel_method=None, max_length=None,
                 keep_ratio=1, device=None):
        # (no GPU is touched until upload(): the window tables and the rank-assignment weights are host work)
        self.num_bins = num_bins
        self.data_path = data_path
        self.keep_ratio = keep_ratio
        self.sensor_resolution = sensor_resolution
        self.has_images = True
        self.channels = num_bins
        self._device = torch.device(device) if device is not None else None
        self.load_data(data_path)
        if voxel_method is None:
            voxel_method = {'method': 'between_frames'}
        self.voxel_method = voxel_method
        self.set_voxel_method()
        if max_length is not None:
            self.length = min(self.length, max_length + 1)
        self._vox = None
        self._dev = None
        self._host_cols = None
        self._img = None
        self._255 = None
        self._table = None

    @property
    def device(self):
        if self._device is None:
            _lib.require_gpu()
            self._device = torch.device('cuda', torch.cuda.current_device())
        return self._device

    # -- loading: what dataset.py:230-281 establishes, organised around the GPU path ------------
    def load_data(self, data_path):
        """Open one sequence directory.  The event columns stay memory-mapped until `upload()` moves them
        into HBM; only the small per-frame tables are read eagerly.  Establishes the same facts as the reference's
        loader: `t0`/`tk`, `num_events`, `num_frames`, `frame_ts`, `has_images`, and the sensor resolution taken from,
        in this order, the constructor argument, metadata.json, the first reference frame, or max(xy)+1."""
        if not os.path.isdir(data_path):
            raise AssertionError(f'{data_path} is not a valid data_path')
        seq = SequenceFiles.open(data_path)
        self.seq = seq
        self.has_images = seq.images is not None
        self.num_events = seq.num_events
        self.t0, self.tk = seq.t[0], seq.t[-1]
        self.num_frames = 0 if seq.images is None else len(seq.images)
        # dataset.py:262: [ts.item() for ts in frame_stamps] -- one stamp per row; an images.npy with zero frames yields []
        # there and falls through to the max(xy)+1 resolution
        stamps = [] if seq.images is None else np.asarray(seq.frame_stamps)
        self.frame_ts = [float(np.asarray(r).reshape(-1)[0]) for r in stamps]
        if self.num_frames == 0:
            self.has_images = False      # (the reference would go on to index frame_ts[0] in get_min_max_t and raise)
        if len(self.frame_ts) != self.num_frames:
            raise AssertionError("Number of frames and timestamps do not match")
        res = self.sensor_resolution
        if res is None:
            res = seq.metadata_resolution()
        if res is None and self.num_frames > 0:
            res = seq.images[0].shape[:2]
        if res is None:
            res = [np.max(seq.xy[:, 1]) + 1, np.max(seq.xy[:, 0]) + 1]
        self.sensor_resolution = [int(res[0]), int(res[1])]

    @property
    def filehandle(self):
        """The reference's dict of handles (dataset.py:232-251), for code written against it."""
        d = {'t': self.seq.t, 'xy': self.seq.xy, 'p': self.seq.p}
        if self.seq.images is not None:
            d.update(images=self.seq.images, frame_stamps=self.seq.frame_stamps,
                     image_event_indices=self.seq.image_event_indices)
        return d

    # -- window tables (dataset.py:104-130,168-186,287-294) --------------------------------------
    def set_voxel_method(self):
        vm = self.voxel_method
        if vm['method'] == 'k_events':
            step = vm['k'] - vm['sliding_window_w']
            self.length = max(int(self.num_events / step), 0)
            i0 = np.arange(self.length, dtype=np.int64) * step
            self.event_indices = np.stack([i0, i0 + vm['k']], axis=1)
        elif vm['method'] == 't_seconds':
            step = vm['t'] - vm['sliding_window_t']
            self.length = max(int((self.tk - self.t0) / step), 0)
            end_times = (step * np.arange(self.length)) + self.t0 + vm['t']
            ends = np.searchsorted(self.filehandle["t"], end_times).astype(np.int64)
            starts = np.concatenate([[0], ends[:-1]]).astype(np.int64) if self.length else ends
            self.event_indices = np.stack([starts, ends], axis=1)
        elif vm['method'] == 'between_frames':
            assert self.has_images, "Cannot use between_frames voxel method without images"
            self.length = self.num_frames - 1
            ends = np.asarray(self.filehandle["image_event_indices"]).reshape(len(self.filehandle["image_event_indices"]), -1)[:, 0].astype(np.int64)
            starts = np.concatenate([[0], ends[:-1]]).astype(np.int64)
            self.event_indices = np.stack([starts, ends], axis=1)
            self.choose_frames_to_use()
        else:
            raise ValueError("Invalid voxel forming method chosen ({})".format(vm))

    def choose_frames_to_use(self):
        self.frames_to_use = list(range(0, self.num_frames))
        if self.keep_ratio != 1:
            assert self.voxel_method['method'] == 'between_frames', \
                "keep_ratio can only specified for between_frames voxel method"
            assert self.keep_ratio < 1, "keep_ratio cannot be greater than 1"
            n_use = int(self.num_frames * self.keep_ratio)
            # unseeded, as in the reference (dataset.py:139): the kr* configs are not reproducible there either
            self.frames_to_use = sorted(np.random.choice(self.frames_to_use, size=n_use, replace=False))
            self.length = n_use - 1

    def __len__(self):
        return self.length

    def get_min_max_t(self):
        if self.has_images:
            return min(self.frame_ts[0], self.t0), max(self.frame_ts[-1], self.tk)
        return self.t0, self.tk

    def get_closest_frame_index(self, ts):
        pos = bisect_left(self.frame_ts, ts)
        if pos == 0:
            return 0
        if pos == len(self.frame_ts):
            return pos - 1
        return pos if self.frame_ts[pos] - ts < ts - self.frame_ts[pos - 1] else pos - 1

    # -- per-item bookkeeping for the whole sequence (dataset.py:33-102) --------------------------
    def table(self):
        """dict of numpy arrays over items: idx0, idx1, event_count, ts_0, ts_k, dt, frame_index,
        frame_timestamp, voxel_timestamp, valid (False where the reference raises, dataset.py:196-197)."""
        if self._table is not None:
            return self._table
        L, vm, t = self.length, self.voxel_method, self.filehandle["t"]
        method = vm['method']
        if method == 'between_frames':
            ftu = np.asarray(self.frames_to_use, dtype=np.int64)
            cur = ftu[:L]
            prev = np.concatenate([[0], ftu[:max(L - 1, 0)]])[:L]
            idx0 = self.event_indices[prev, 1]; idx1 = self.event_indices[cur, 1]
            fidx = cur.copy()
        else:
            idx0 = self.event_indices[:L, 0].copy(); idx1 = self.event_indices[:L, 1].copy()
            fidx = np.arange(L, dtype=np.int64)
        valid = (idx0 >= 0) & (idx1 <= self.num_events)
        cnt = np.where(valid, np.maximum(idx1 - idx0, 0), 0)
        ts_0 = np.zeros(L); ts_k = np.zeros(L)
        has = valid & (cnt > 0)
        ts_0[has] = t[idx0[has]]; ts_k[has] = t[idx1[has] - 1]
        empty = valid & (cnt == 0) & (idx0 > 0)
        if empty.any():
            last = np.asarray(t[idx0[empty] - 1])
            ts_0[empty] = last
            if method == 't_seconds':
                ts_k[empty] = last + vm['t']
            else:
                fts = np.asarray(self.frame_ts)
                # dataset.py:68: frame_ts[index] with the ITEM index for k_events (may raise there too)
                idx = fidx[empty]
                if (idx >= len(fts)).any():
                    valid[np.flatnonzero(empty)[idx >= len(fts)]] = False
                    idx = np.minimum(idx, len(fts) - 1)
                ts_k[empty] = fts[idx]
        dt = ts_k - ts_0
        if method == 't_seconds':
            dt = np.full(L, float(vm['t']))
        if self.has_images and method != 'between_frames':
            fts = np.asarray(self.frame_ts)
            pos = np.searchsorted(fts, ts_k, side='left')          # bisect_left
            pos_c = np.clip(pos, 1, len(fts) - 1)
            choose_after = (fts[pos_c] - ts_k) < (ts_k - fts[pos_c - 1])
            fidx = np.where(pos == 0, 0, np.where(pos == len(fts), len(fts) - 1, np.where(choose_after, pos_c, pos_c - 1)))
        if self.has_images:
            frame_ts = np.asarray(self.frame_ts)[fidx]
        else:
            frame_ts = np.zeros(L)
        vts = frame_ts if method == 'between_frames' else ts_k
        self._table = dict(idx0=idx0.astype(np.int64), idx1=idx1.astype(np.int64), event_count=cnt.astype(np.int64),
                           ts_0=ts_0, ts_k=ts_k, dt=dt, frame_index=np.asarray(fidx, dtype=np.int64),
                           frame_timestamp=frame_ts, voxel_timestamp=np.asarray(vts, dtype=np.float64), valid=valid)
        return self._table

    # -- device residency ----------------------------------------------------------------------
    def host_events(self, keep=False):
        """The sequence's event columns as the arrays that go to HBM (int16 xy, float64 t, uint8 p), validated.  The result is
        cached until it has been handed out once with keep=False (eval's grouping validates a sequence before the batch that
        uploads it is formed: one pass over the files, and the host copy is released as soon as it has been concatenated)."""
        if self._host_cols is not None:
            cols = self._host_cols
            if not keep:
                self._host_cols = None
            return cols
        fh = self.filehandle
        xy = np.ascontiguousarray(fh["xy"])
        # The reference trusts the coordinates: index_put_ raises beyond the sensor and WRAPS negative
        # ones.  Here coordinates beyond the sensor are dropped by the kernel and counted (evr_voxelize_dropped); negative
        # ones cannot be represented in the resident int16 form and are refused outright.
        if xy.size and (xy.min() < 0 or xy.max() >= 32768):
            raise ValueError(f"{self.data_path}: pixel coordinates outside [0, 32767] (min {xy.min()}, max {xy.max()})")
        # ... and a coordinate beyond sensor_resolution (a wrong constructor argument / metadata.json) would give plausible
        # but wrong voxel grids: fail here, once per sequence, the way the reference fails on the first such window -- for the
        # events some window of this dataset object can touch only (the reference never indexes the others: events before the
        # first / after the last window, or beyond max_length)
        H, W = self.sensor_resolution
        tb = self.table()
        ok = tb['valid'] & (tb['idx1'] > tb['idx0'])
        if xy.size and ok.any():
            lo, hi = int(tb['idx0'][ok].min()), int(tb['idx1'][ok].max())
            used = xy[max(lo, 0):max(hi, 0)]
            if used.size and (int(used[:, 0].max()) >= W or int(used[:, 1].max()) >= H):
                raise IndexError(f"{self.data_path}: event coordinates up to x={int(used[:, 0].max())}, y={int(used[:, 1].max())} "
                                 f"lie outside the {W}x{H} sensor (sensor_resolution is [H, W]): index out of range in the "
                                 "voxel grid")
        pol = np.ascontiguousarray(fh["p"])
        # dataset.py:227 computes p*2-1 from {0,1}; a file that stores -1/+1 (or anything else) would silently become
        # 255 -> weight 509 after a uint8 cast
        # (one max / min pass: np.isin over millions of events was a third of a short call's set-up)
        # (a float file holding exactly 0.0 / 1.0 is accepted, as the reference's p.astype(float32)*2-1 accepts it)
        if pol.size:
            if pol.dtype.kind in 'ub':
                bad_pol = pol.max() > 1
            elif pol.dtype.kind == 'i':
                bad_pol = pol.max() > 1 or pol.min() < 0
            elif pol.dtype.kind == 'f':
                bad_pol = not bool(((pol == 0) | (pol == 1)).all())
            else:
                bad_pol = True
        if pol.size and bad_pol:
            raise ValueError(f"{self.data_path}: events_p.npy must hold 0/1 (or bool) polarities, found values "
                             f"{np.unique(pol)[:6].tolist()}")
        cols = (xy.astype(np.int16), np.array(fh["t"], dtype=np.float64), pol.astype(np.uint8))
        if keep:
            self._host_cols = cols
        return cols

    def upload_images(self):
        """Reference frames (uint8) into HBM once."""
        if self._img is None and self.has_images:
            _lib.require_gpu()
            self._img = torch.from_numpy(np.array(self.filehandle["images"][..., 0])).to(self.device)   # [F,H,W] u8
        return self._img

    def upload(self):
        """Move the sequence's events (13 B each) and reference frames (uint8) into HBM once."""
        if self._dev is not None:
            return self._dev
        xy, t, pol = self.host_events()
        _lib.require_gpu()
        d = {'xy': torch.from_numpy(xy).to(self.device), 'ts': torch.from_numpy(t).to(self.device),
             'p': torch.from_numpy(pol).to(self.device)}
        if self.has_images:
            d["images"] = self.upload_images()
        self._dev = d
        self._vox = Voxelizer(self.device)
        return d

    def voxel_batch(self, items, out=None, stats=None):
        """Voxel grids [len(items), B, H, W] (+ stats [len,3]) for the given item indices, one launch."""
        tb, d = self.table(), self.upload()
        items = np.asarray(items, dtype=np.int64)
        if not tb['valid'][items].all():
            bad = int(items[~tb['valid'][items]][0])
            raise ValueError("WARNING: Event indices {},{} out of bounds 0,{}".format(
                int(tb['idx0'][bad]), int(tb['idx1'][bad]), self.num_events))
        b = tb['idx0'][items]; e = tb['idx1'][items]
        lens = np.maximum(e - b, 0)
        base = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(self.device)
        H, W = self.sensor_resolution
        if stats is None:
            stats = torch.zeros((len(items), 3), dtype=torch.float64, device=self.device)
        grid = self._vox.voxelize_raw_windows(d['xy'], d['ts'], d['p'], dev(b), dev(e), dev(base), int(lens.sum()),
                                              self.num_bins, (H, W), out=out, stats=stats)
        return grid, stats

    def raise_if_dropped(self):
        """Poll the tensorizer's cumulative out-of-sensor counter (one synchronisation; call once per sequence)."""
        if self._vox is not None:
            self._vox.raise_if_dropped(f"events of {self.data_path}")

    def window_cost(self):
        """Work of this sequence for the rank assignment: windows x padded pixels of the largest crop the
        methods use (a multiple of 16 covers every method's num_encoders)."""
        H, W = self.sensor_resolution
        return int(max(self.length, 1)) * (-(-H // 16) * 16) * (-(-W // 16) * 16)

    def frames(self, frame_indices, out=None):
        """Reference frames [n,1,H,W] fp32 in [0,1] (dataset.py:80-85: images[i][:,:,0] / 255)."""
        images = self.upload_images()
        # (a DEVICE index tensor is used as it is: the frame loop uploads a sequence's frame indices once -- a host list here is a
        # synchronous copy in stream order, i.e. a host <-> GPU rendezvous per call: behind the evaluation stream's wait for a
        # chunk's network steps it cost the drop-in its whole run-ahead, 5 ms per chunk of 8 frames)
        if isinstance(frame_indices, torch.Tensor) and frame_indices.is_cuda:
            idx = frame_indices
        else:
            idx = torch.from_numpy(np.asarray(frame_indices, dtype=np.int64)).to(self.device)
        # tensor / tensor: a true IEEE division (torch's CUDA kernel turns `/ python_scalar` into `* (1/255)`,
        # which differs from the reference's CPU result in the last bit)
        if self._255 is None:
            self._255 = torch.tensor(255.0, dtype=torch.float32, device=self.device)
        if out is not None:
            return torch.div(images[idx].to(torch.float32), self._255, out=out)
        return torch.div(images[idx].to(torch.float32), self._255).unsqueeze(1)

    def __getitem__(self, index):
        assert 0 <= index < len(self), f"index {index} out of bounds (0 <= x < {len(self)})"
        tb = self.table()
        grid, _ = self.voxel_batch([index])
        if self.has_images:
            frame = self.frames([tb['frame_index'][index]])[0]
            fts = torch.tensor(tb['frame_timestamp'][index], dtype=torch.float64)
        else:
            frame = torch.zeros((1, *self.sensor_resolution), dtype=torch.float32, device=self.device)
            fts = torch.tensor(0.0, dtype=torch.float64)
        return {'frame': frame, 'events': grid[0], 'frame_timestamp': fts,
                'voxel_timestamp': torch.tensor(tb['voxel_timestamp'][index], dtype=torch.float64),
                'dt': torch.tensor(tb['dt'][index], dtype=torch.float64), 'event_count': int(tb['event_count'][index])}


class SequenceBatch:
    """S sequences of one sensor size whose events sit in ONE resident array, so that a single tensorizer launch covers
    windows of different sequences (step-major: window (step, slot)) -- the input side of eval.eval_method_on_sequences."""

    def __init__(self, datasets):
        assert len(datasets) >= 1
        self.dss = list(datasets)
        H, W = self.dss[0].sensor_resolution
        assert all(tuple(d.sensor_resolution) == (H, W) for d in self.dss), "batched sequences must share the sensor size"
        self.H, self.W, self.num_bins = H, W, self.dss[0].num_bins
        self.device = self.dss[0].device
        cols = [d.host_events() for d in self.dss]      # (validated; a cached copy from eval's grouping is handed over and dropped)
        self.base = np.concatenate([[0], np.cumsum([len(c[1]) for c in cols])]).astype(np.int64)    # slot j = events [base[j], base[j+1])
        _lib.require_gpu()
        n_ev = int(self.base[-1])
        # one resident array per column, filled slot by slot: the host never holds a concatenated second copy of the batch
        self.xy = torch.empty((n_ev, 2), dtype=torch.int16, device=self.device)
        self.ts = torch.empty((n_ev,), dtype=torch.float64, device=self.device)
        self.p = torch.empty((n_ev,), dtype=torch.uint8, device=self.device)
        for j in range(len(cols)):
            b, e = int(self.base[j]), int(self.base[j + 1])
            if e > b:
                self.xy[b:e].copy_(torch.from_numpy(cols[j][0].reshape(-1, 2)))
                self.ts[b:e].copy_(torch.from_numpy(cols[j][1]))
                self.p[b:e].copy_(torch.from_numpy(cols[j][2]))
            cols[j] = None
        self.vox = Voxelizer(self.device)

    def voxel_steps(self, items, out, stats, stream=None):
        """items[j] = the item indices of slot j for this chunk (ragged: an exhausted slot gets empty windows);
        out [n_steps, S, B, H, W], stats [n_steps, S, 3] are filled for n_steps = max(len(items[j]))."""
        S = len(self.dss)
        n = max(len(it) for it in items)
        b = np.zeros((n, S), np.int64); e = np.zeros((n, S), np.int64)
        for j, (ds, it) in enumerate(zip(self.dss, items)):
            tb = ds.table()
            it = np.asarray(it, dtype=np.int64)
            if len(it) and not tb['valid'][it].all():
                bad = int(it[~tb['valid'][it]][0])
                raise ValueError("WARNING: Event indices {},{} out of bounds 0,{}".format(
                    int(tb['idx0'][bad]), int(tb['idx1'][bad]), ds.num_events))
            b[:len(it), j] = tb['idx0'][it] + self.base[j]
            e[:len(it), j] = tb['idx1'][it] + self.base[j]
        b = b.reshape(-1); e = np.maximum(e.reshape(-1), b)
        lens = e - b
        rec = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(self.device, non_blocking=True)
        self.vox.voxelize_raw_windows(self.xy, self.ts, self.p, dev(b), dev(e), dev(rec), int(lens.sum()), self.num_bins,
                                      (self.H, self.W), out=out[:n].view(n * S, self.num_bins, self.H, self.W),
                                      stats=stats[:n].view(n * S, 3), stream=stream)
        return n

    def raise_if_dropped(self):
        self.vox.raise_if_dropped("events of " + ", ".join(d.data_path for d in self.dss))

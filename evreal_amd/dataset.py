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


class MemMapDataset:
    def __init__(self, data_path, sensor_resolution=None, num_bins=5, voxel_method=None, max_length=None,
                 keep_ratio=1, device=None):
        _lib.require_gpu()
        self.num_bins = num_bins
        self.data_path = data_path
        self.keep_ratio = keep_ratio
        self.sensor_resolution = sensor_resolution
        self.has_images = True
        self.channels = num_bins
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.load_data(data_path)
        if voxel_method is None:
            voxel_method = {'method': 'between_frames'}
        self.voxel_method = voxel_method
        self.set_voxel_method()
        if max_length is not None:
            self.length = min(self.length, max_length + 1)
        self._vox = None
        self._dev = None
        self._table = None

    # -- loading (dataset.py:230-281) -----------------------------------------------------------
    def load_data(self, data_path):
        assert os.path.isdir(data_path), f'{data_path} is not a valid data_path'
        p = lambda n: os.path.join(data_path, n)
        data = {}
        if os.path.exists(p('images_ts.npy')) and os.path.exists(p('images.npy')) and \
                os.path.exists(p('image_event_indices.npy')):
            data["frame_stamps"] = np.load(p('images_ts.npy'))
            data["images"] = np.load(p('images.npy'), mmap_mode='r')
            data["image_event_indices"] = np.load(p('image_event_indices.npy'))
            self.has_images = True
        else:
            self.has_images = False
        data["t"] = np.load(p('events_ts.npy'), mmap_mode='r').squeeze()
        data["xy"] = np.load(p('events_xy.npy'), mmap_mode='r').squeeze()
        data["p"] = np.load(p('events_p.npy'), mmap_mode='r').squeeze()
        assert len(data['p']) == len(data['xy']) == len(data['t']), \
            "Number of events, timestamps and coordinates do not match"
        self.t0, self.tk = data['t'][0], data['t'][-1]
        self.num_events = len(data['p'])
        self.frame_ts = []
        if self.has_images:
            self.num_frames = len(data['images'])
            self.frame_ts = [ts.item() for ts in data["frame_stamps"]]
        else:
            self.num_frames = 0
        assert len(self.frame_ts) == self.num_frames, "Number of frames and timestamps do not match"
        self.filehandle = data
        if self.sensor_resolution is None:
            meta = p("metadata.json")
            if os.path.exists(meta):
                self.sensor_resolution = read_json(meta)["sensor_resolution"]
            elif self.has_images and self.num_frames > 0:
                self.sensor_resolution = data["images"][0].shape[:2]
            else:
                self.sensor_resolution = [np.max(data["xy"][:, 1]) + 1, np.max(data["xy"][:, 0]) + 1]
        self.sensor_resolution = [int(self.sensor_resolution[0]), int(self.sensor_resolution[1])]

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
    def upload(self):
        """Move the sequence's events (13 B each) and reference frames (uint8) into HBM once."""
        if self._dev is not None:
            return self._dev
        fh = self.filehandle
        xy = np.ascontiguousarray(fh["xy"])
        assert xy.min() >= 0 and xy.max() < 32768, "pixel coordinates do not fit int16"
        d = {'xy': torch.from_numpy(xy.astype(np.int16)).to(self.device),
             'ts': torch.from_numpy(np.array(fh["t"], dtype=np.float64)).to(self.device),
             'p': torch.from_numpy(np.ascontiguousarray(fh["p"]).astype(np.uint8)).to(self.device)}
        if self.has_images:
            d["images"] = torch.from_numpy(np.array(fh["images"][..., 0])).to(self.device)   # [F,H,W] u8
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

    def frames(self, frame_indices):
        """Reference frames [n,1,H,W] fp32 in [0,1] (dataset.py:80-85: images[i][:,:,0] / 255)."""
        d = self.upload()
        idx = torch.from_numpy(np.asarray(frame_indices, dtype=np.int64)).to(self.device)
        # tensor / tensor: a true IEEE division (torch's CUDA kernel turns `/ python_scalar` into `* (1/255)`,
        # which differs from the reference's CPU result in the last bit)
        if '_255' not in d:
            d['_255'] = torch.tensor(255.0, dtype=torch.float32, device=self.device)
        return torch.div(d['images'][idx].to(torch.float32), d['_255']).unsqueeze(1)

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

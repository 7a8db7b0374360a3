"""Oracle: window tables and per-item bookkeeping of MemMapDataset.  TEST INFRASTRUCTURE ONLY.

Restates dataset.py of the reference on in-memory numpy arrays:
  k_indices            dataset.py:119-130,171-173  [i*(k-w), i*(k-w)+k), length int(N/(k-w))
  timeblock_indices    dataset.py:104-117,174-177  searchsorted chain, length int(dur/(t-w))
  frame_indices        dataset.py:287-294          [prev_end, image_event_indices[f])
  closest_frame_index  dataset.py:151-166          bisect_left, ties -> earlier
  window_item          dataset.py:33-102           (idx0, idx1, ts_0, ts_k, dt, frame index, voxel ts)
"""
from bisect import bisect_left
import numpy as np


def k_indices(num_events, k, w):
    length = max(int(num_events / (k - w)), 0)
    return [[(k - w) * i, (k - w) * i + k] for i in range(length)]


def timeblock_indices(t, t_win, w):
    t0, tk = t[0], t[-1]
    length = max(int((tk - t0) / (t_win - w)), 0)
    out, start = [], 0
    for i in range(length):
        start_time = ((t_win - w) * i) + t0
        end = int(np.searchsorted(t, start_time + t_win))
        out.append([start, end]); start = end
    return out


def frame_indices(image_event_indices):
    out, start = [], 0
    for e in np.asarray(image_event_indices).reshape(-1):
        out.append([start, int(e)]); start = int(e)
    return out


def closest_frame_index(frame_ts, ts):
    pos = bisect_left(frame_ts, ts)
    if pos == 0:
        return 0
    if pos == len(frame_ts):
        return pos - 1
    return pos if frame_ts[pos] - ts < ts - frame_ts[pos - 1] else pos - 1


def window_item(method, index, event_indices, t, frame_ts, num_events, t_win=None):
    """Returns dict(idx0, idx1, event_count, ts_0, ts_k, dt, frame_index, voxel_timestamp)
    following dataset.py:33-102 for a dataset with reference frames."""
    if method == 'between_frames':
        prev = index - 1 if index > 0 else 0          # frames_to_use is identity at keep_ratio 1
        idx0 = event_indices[prev][1]; idx1 = event_indices[index][1]
        if index == 0:
            idx0 = event_indices[0][1]                # quirk 1: item 0 is always empty
    else:
        idx0, idx1 = event_indices[index]
    if not (idx0 >= 0 and idx1 <= num_events):
        raise ValueError("event indices out of bounds")
    n = max(idx1 - idx0, 0)
    fidx = index
    if n > 0:
        ts_0, ts_k = t[idx0], t[idx1 - 1]
    elif idx0 > 0:
        ts_0 = t[idx0 - 1]
        ts_k = ts_0 + t_win if method == 't_seconds' else frame_ts[index]
    else:
        ts_0, ts_k = 0, 0
    dt = ts_k - ts_0
    if method == 't_seconds':
        dt = t_win
    if method != 'between_frames':
        fidx = closest_frame_index(frame_ts, ts_k)
    vts = frame_ts[fidx] if method == 'between_frames' else ts_k
    return dict(idx0=int(idx0), idx1=int(idx1), event_count=int(n), ts_0=float(ts_0), ts_k=float(ts_k),
                dt=float(dt), frame_index=int(fidx), voxel_timestamp=float(vts))

#!/usr/bin/env python3
"""Drop-in for the reference's eval.py (same CLI, same config JSONs, same output files):

    python -m evreal_amd.eval -m E2VID FireNet -c std k15k -d ECD -qm mse ssim
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m evreal_amd.eval ...

Mirrors evaluate() / eval_method_with_config() / eval_method_on_sequence() (eval.py:189-246,333-445) with the
hot path on the GPU: a sequence's events live in HBM, windows are voxelised in chunks with one launch, the
recurrent network steps frame by frame without host synchronisation, post-normalisation and MSE/SSIM run per
chunk.  Under torch.distributed the sequences of every dataset are sharded across ranks and the per-dataset
totals are folded with ONE all-reduce (evreal_amd.dist).  `--batch-sequences S` (or eval-config key
`batch_sequences`; default 8) advances S sequences of a rank together, one batch slot each.
"""
import argparse
import glob
import os
import sys
import traceback
import types
from collections import OrderedDict

import numpy as np
import torch
from tabulate import tabulate

from . import model as model_arch
from .config import get_dataset_configs, get_eval_configs, get_method_config
from .dataset import MemMapDataset
from .dist import assign_sequences, force_collectives, reduce_metric_sums
from .eval_metrics import EvalMetricsTracker, MetricTracker
from .lib import EvrError
from .prepost import normalize_event_tensor, post_process_normalization

CHUNK = 16   # windows voxelised / frames scored per launch
# sequences of one dataset advanced together by default (same output files as one at a time -- tests/test_gpu_eval.py; a
# batch-1 step is bound by kernel latency: 1150 frames/s against 4300 at 8).  --batch-sequences 1 is the reference's loop.
DEFAULT_BATCH_SEQUENCES = '8'
TIMINGS = []  # host seconds of every eval_method_on_sequences call {setup, enqueue, book, finalize, frames, sequences} (bench.py reads it)


# ------------------------------------------------------------------------------------------------
# datasets (eval.py:38-105)
def get_sequences(dataset_config, dataset_kwargs):
    dataset_root = dataset_config['root_path']
    get_all = dataset_config.get('get_all_sequences', False)
    has_subfolders = dataset_config.get('has_subfolders', False)
    dataset_kwargs.update(dataset_config.get('dataset_kwargs', {}))
    if get_all:
        pattern = os.path.join(dataset_root, '*', '*') if has_subfolders else os.path.join(dataset_root, '*')
        sequences_config = OrderedDict()
        for path in glob.glob(pattern):
            name = (os.path.basename(os.path.dirname(path)) + "_" + os.path.basename(path)) if has_subfolders \
                else os.path.basename(path)
            sequences_config[name] = {'sequence_path': path}
    else:
        sequences_config = dataset_config.get('sequences', {})
    sequences = []
    for name, sequence in sequences_config.items():
        sequence = dict(sequence)
        sequence['name'] = name
        sequence['sequence_path'] = sequence.get('sequence_path', os.path.join(dataset_root, name))
        sequence['dataset_kwargs'] = dict(dataset_kwargs)
        sequences.append(sequence)
    return sequences


def open_sequence(sequence):
    """Instantiate the reader lazily (only on the rank that owns the sequence)."""
    if 'dataset' not in sequence:
        ds = MemMapDataset(sequence['sequence_path'], **sequence['dataset_kwargs'])
        sequence['dataset'] = ds
        min_t, max_t = ds.get_min_max_t()
        sequence.setdefault('start_time_s', min_t)
        sequence.setdefault('end_time_s', max_t)
    return sequence['dataset']


def get_datasets(dataset_configs, dataset_kwargs):
    return [{'name': c['name'], 'sequences': get_sequences(c, dict(dataset_kwargs))} for c in dataset_configs]


# ------------------------------------------------------------------------------------------------
# models (eval.py:109-158)
class _ConfigParserShim:
    """Stand-in so torch.load can unpickle the `parse_config.ConfigParser` object stored inside the
    E2VID+/FireNet+/HyperE2VID checkpoints (parse_config.py:1-22 of the reference)."""

    @property
    def config(self):
        return self._config

    def __getitem__(self, name):
        return self._config[name]


def _load_checkpoint(path):
    shim = types.ModuleType('parse_config')
    shim.ConfigParser = _ConfigParserShim
    had = sys.modules.get('parse_config')
    sys.modules['parse_config'] = shim
    try:
        return torch.load(path, map_location='cpu', weights_only=False)
    finally:
        if had is not None:
            sys.modules['parse_config'] = had
        else:
            del sys.modules['parse_config']


_MODEL_CACHE = OrderedDict()      # (method, checkpoint identity, arithmetic switches) -> model with packed weights resident in HBM


def _model_cache_key(model_name, checkpoint_path):
    from . import lib as _lib
    _lib.require_gpu()          # (EvrError on a machine without a GPU, before torch is asked for its current device)
    st = os.stat(checkpoint_path)
    # EVR_ARITH / EVR_FP32 / EVR_FIRENET_* are read when a model is created, so they belong in the key.  Most other EVR_* switches
    # (EVR_KSPLIT, EVR_C16_*, EVR_WIDE*, EVR_LPIPS_*) are function-local statics of the library, read ONCE PER PROCESS: changing
    # them inside a process gives a new cache entry that still runs the old setting -- set them before the first call.
    switches = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith('EVR_')))
    return (model_name, os.path.abspath(checkpoint_path), st.st_mtime_ns, st.st_size, torch.cuda.current_device(), switches)


def get_model_from_checkpoint_path(model_name, checkpoint_path):
    """eval.py:124-158.  The reference re-reads the checkpoint for every (eval config, method) pair of a run; unpickling + re-laying
    and splitting the weights for the matrix cores (evr_model_create: 0.1 s for E2VID) is most of a short call's set-up, so a model is
    kept per (method, file identity, EVR_* switches) for the life of the process -- EVREAL_MODEL_CACHE=0 disables it.  Recurrent
    state is reset per sequence by the callers, as before."""
    if os.environ.get('EVREAL_MODEL_CACHE', '1') == '0':
        return _build_model(model_name, checkpoint_path)
    key = _model_cache_key(model_name, checkpoint_path)
    model = _MODEL_CACHE.get(key)
    if model is None:
        model = _build_model(model_name, checkpoint_path)
        _MODEL_CACHE[key] = model
        while len(_MODEL_CACHE) > 4:
            _, old = _MODEL_CACHE.popitem(last=False)
            # Its HBM (weights, activations and state of the largest batch it ran) goes back NOW when the cache held the only
            # reference; a caller that still holds the model (the reference's get_model_from_checkpoint_path hands out models
            # the caller owns: eight methods loaded up front outlive a four-entry cache) keeps a working object -- it is freed by
            # __del__ when the caller lets go.  refcount 2 = `old` + the argument of getrefcount.
            if sys.getrefcount(old) <= 2 and hasattr(old, 'destroy'):
                old.destroy()
    else:
        _MODEL_CACHE.move_to_end(key)
    return model


def _build_model(model_name, checkpoint_path):
    checkpoint = _load_checkpoint(checkpoint_path)
    if model_name == "SPADE-E2VID":    # eval.py:130-133
        model, state_dict = model_arch.SpadeE2vid(), checkpoint
    elif model_name == "SSL-E2VID":    # eval.py:134-139
        kw = {"base_num_channels": 32, "kernel_size": 5, "num_bins": 5, "num_encoders": 3,
              "recurrent_block_type": "convlstm", "num_residual_blocks": 2, "skip_type": "sum", "norm": None,
              "use_upsample_conv": True}
        model, state_dict = model_arch.E2VIDRecurrent(kw), checkpoint
    elif model_name == "E2VID":        # eval.py:141-144
        kw = dict(checkpoint['model']); kw['final_activation'] = 'sigmoid'
        model, state_dict = model_arch.E2VIDRecurrent(kw), checkpoint['state_dict']
    elif model_name == "FireNet":      # eval.py:145-148
        kw = dict(checkpoint['config']['model']); kw['final_activation'] = ''
        model, state_dict = model_arch.FireNet_legacy(unet_kwargs=kw), checkpoint['state_dict']
    else:                              # eval.py:149-155: config.init_obj('arch', model_arch)
        arch = checkpoint['config']['arch']
        cls = getattr(model_arch, arch['type'], None)
        if cls is None:
            raise EvrError(f"architecture {arch['type']!r} is not available in evreal_amd.model")
        model = cls(**dict(arch['args']))
        if model_name == "ET-Net":          # eval.py:152-153
            model.num_encoders = 3
        elif model_name == "FireNet+":
            model.num_encoders = 0
        state_dict = checkpoint['state_dict']
    model.load_state_dict(state_dict)
    return model


# ------------------------------------------------------------------------------------------------
def get_eval_metrics_tracker(dataset_name, eval_config, method_name, sequence, metrics):
    output_path = os.path.join("outputs", eval_config['name'], dataset_name, sequence['name'], method_name)
    save_images = eval_config.get('save_images', True)
    return EvalMetricsTracker(save_images=save_images,
                              save_processed_images=save_images and eval_config['histeq'] != 'none',
                              output_dir=output_path, hist_eq=eval_config['histeq'], quan_eval_metric_names=metrics,
                              quan_eval_start_time=sequence['start_time_s'], quan_eval_end_time=sequence['end_time_s'],
                              quan_eval_ts_tol_ms=eval_config['ts_tol_ms'],
                              has_reference_frames=sequence['dataset'].has_images,
                              color=eval_config.get('color', False))


def _plan_items(ds, tb, sequence, infer_all):
    """Which items run (eval.py:212-216); an item the reference's loader raises on stops the loop there.
    Returns (items, index of the raising item or None, the reference loop's last `idx`)."""
    todo, bad, idx = [], None, 0
    for idx in range(len(ds)):
        if not tb['valid'][idx]:
            bad = idx
            break
        ts = tb['voxel_timestamp'][idx]
        if ts < sequence['start_time_s'] - 10 and not infer_all:
            continue
        if ts > sequence['end_time_s'] and not infer_all:
            idx -= 1
            break
        todo.append(idx)
    return todo, bad, idx


def _range_guard_policy(eval_config):
    """What a non-zero range counter of a split arithmetic does (eval config key `range_guard`, env EVREAL_RANGE_GUARD):
    'rerun' (default) -- the pass is abandoned and the whole group of sequences re-run on the exact-fp32 twin, so results never
    depend on the activation range; that twin is ~2.3x slower (2.4k vs 5.7k frames/s at 64 sequences) and its activations are
    allocated after the first model's are released.  'warn' -- keep the first pass and print the counters (in the opt-in
    `mx` / `mx6` modes a non-zero counter means some values kept f16-only precision, 2^-12 relative, not that they were clamped:
    a user of those modes may prefer the fast result).  'off' -- no polling."""
    v = str(eval_config.get('range_guard', os.environ.get('EVREAL_RANGE_GUARD', 'rerun'))).lower()
    return v if v in ('rerun', 'warn', 'off') else 'rerun'


_DEFER_PNG = [False]      # True while eval_method_with_config's dataset loop runs: it waits for the native PNG writers once per dataset


class _Saturated(Exception):
    """Activations of the running chunk left the split arithmetic's exact range (evr_model_saturation): nothing of it is booked."""


def _rerun_exact(model, trackers, names):
    """Saturation must not change results (the reference computes in fp32 throughout, model/submodules.py:227-245, and has no range
    limit): drain the GPU, drop what the first pass produced -- its trackers are closed, and the re-run's trackers truncate every
    file they own when they are created -- say so, and hand back the model's exact-fp32 twin (the library's fp32-MFMA HIP kernels
    on PLAIN tensors; never the oracle, never a CPU path)."""
    torch.cuda.synchronize()
    n, layer = model.saturation(clear=True)
    for t in trackers:
        t.discard()
    print(f"[evreal_amd.eval] {n} activation runs of {names} left the exact range of the '{model.arith}' split format (most in layer "
          f"'{layer}'): re-running in exact fp32 on the GPU (the library's fp32 kernels); no score, image or line of the first pass is kept")
    # the first model's activations / states for these S sequences go back before the twin allocates its own (both
    # resident at once risked running out of HBM at large batch_sequences); its weights stay for the next group
    if hasattr(model, 'release_shape'):
        model.release_shape()
    return model.exact_twin()


def eval_method_on_sequence(dataset_name, eval_config, method_name, model, method_config, sequence, metrics):
    """eval.py:189-246.  Grayscale evaluation is the one-slot case of eval_method_on_sequences (same pipelined chunk loop);
    colour evaluation (ColorNet, eval.py:222-232) keeps its own frame loop: no outer pad/crop, no metrics."""
    if not eval_config.get('color', False):
        return eval_method_on_sequences(dataset_name, eval_config, method_name, model, method_config, [sequence], metrics)[0]
    ds = open_sequence(sequence)
    tracker = get_eval_metrics_tracker(dataset_name, eval_config, method_name, sequence, metrics)
    try:
        return _eval_color_sequence(ds, tracker, eval_config, model, method_config, sequence)
    except _Saturated:
        exact = _rerun_exact(model, [tracker], sequence['name'])
        tracker = get_eval_metrics_tracker(dataset_name, eval_config, method_name, sequence, metrics)
        return _eval_color_sequence(ds, tracker, eval_config, exact, method_config, sequence)


def _eval_color_sequence(ds, tracker, eval_config, model, method_config, sequence):
    policy = _range_guard_policy(eval_config)
    guard = policy != 'off' and hasattr(model, 'saturation') and getattr(model, 'arith', 'fp32') != 'fp32'
    model.reset_states()
    if guard:
        model.saturation(clear=True)        # counters are cumulative: this sequence starts from zero
    infer_all = eval_config.get('eval_infer_all', False)
    post_norm = method_config.get('post_process_norm', "none")
    norm_in = method_config.get('event_tensor_normalization', False)
    tb = ds.table()
    todo, bad, idx = _plan_items(ds, tb, sequence, infer_all)
    for c0 in range(0, len(todo), CHUNK):
        items = todo[c0:c0 + CHUNK]
        n = len(items)
        grid, stats = ds.voxel_batch(items)
        # eval.py:222-232 with color: normalise the full tensor, no outer pad/crop, ColorNet splits the streams
        if post_norm != 'none':
            raise NotImplementedError("colour evaluation supports post_process_norm='none' only")
        if norm_in:
            normalize_event_tensor(grid, stats)
        bgr = [model(grid[j:j + 1])['image'][0] for j in range(n)]
        if guard and model.saturation()[0]:        # (synchronises; this loop copies every frame to the host anyway)
            if policy == 'rerun':
                raise _Saturated()
            model.model.warn_if_saturated(sequence['name']) if hasattr(model, 'model') else model.warn_if_saturated(sequence['name'])
            guard = False
        tracker.update_batch_color(items, torch.stack(bgr), [float(v) for v in tb['voxel_timestamp'][items]])
        for i in items:
            cnt, dt = int(tb['event_count'][i]), float(tb['dt'][i])
            tracker.save_custom_metric(i, "event_rate", 0 if (cnt <= 1 or dt == 0) else cnt / dt)
    tracker.finalize(idx, wait_png=not _DEFER_PNG[0])
    ds.raise_if_dropped()       # once per sequence: out-of-sensor events the kernel dropped (the reference raises)
    if bad is not None:
        raise ValueError("WARNING: Event indices {},{} out of bounds 0,{}".format(
            int(tb['idx0'][bad]), int(tb['idx1'][bad]), ds.num_events))
    return tracker.get_num_quan_evaluations(), tracker.get_mean_scores()


class _ChunkBuffers:
    """Device + pinned host buffers of one in-flight chunk (the frame loop keeps two: while the host books chunk c, the
    GPU already works on chunk c + 1)."""

    def __init__(self, S, B, H, W, dev, want_scores, want_lpips, want_u8, with_refs):
        f32 = dict(dtype=torch.float32, device=dev)
        self.grid = torch.empty((CHUNK, S, B, H, W), **f32)
        self.stats = torch.zeros((CHUNK, S, 3), dtype=torch.float64, device=dev)
        self.imgs = torch.empty((CHUNK, S, 1, H, W), **f32)
        self.refs = torch.zeros((CHUNK, S, H, W), **f32) if with_refs else None
        self.scores = torch.zeros((CHUNK * S, 2), dtype=torch.float64, device=dev) if want_scores else None
        self.lp = torch.zeros((CHUNK * S,), dtype=torch.float64, device=dev) if want_lpips else None
        self.h_scores = torch.empty((CHUNK * S, 2), dtype=torch.float64).pin_memory() if want_scores else None
        self.h_lp = torch.empty((CHUNK * S,), dtype=torch.float64).pin_memory() if want_lpips else None
        self.u8 = torch.empty((CHUNK, S, H, W), dtype=torch.uint8, device=dev) if want_u8 else None
        self.h_u8 = torch.empty((CHUNK, S, H, W), dtype=torch.uint8).pin_memory() if want_u8 else None
        self.ev_model = torch.cuda.Event()
        self.ev_done = None
        self.items = None
        self.n = 0
        self.h_sat = None       # pinned copy of the model's range-guard counters as of this chunk (eval_method_on_sequences)


def eval_method_on_sequences(dataset_name, eval_config, method_name, model, method_config, sequences, metrics):
    """The per-frame driver of eval.py:189-246 for SEVERAL sequences of one sensor size in lock-step: every sequence
    owns one batch slot of the recurrent network (slots are independent: state, normalisation statistics, metrics), so
    the results equal one eval_method_on_sequence call per sequence, but a frame of batch S costs far less than S
    frames of batch 1 (the batch-1 step is bound by kernel latency, not by the chip).  Not for colour evaluation.

    The loop is a two-deep pipeline over chunks of CHUNK steps: ONE tensorizer launch voxelizes the chunk's windows of all
    slots (their events share one resident array, dataset.SequenceBatch) straight into the step-major batch tensor; the
    network steps run back to back on the main stream; post-normalisation, MSE/SSIM/LPIPS of every frame of the chunk and
    the uint8 conversion for the PNG writers run on a second HIP stream (as pipeline.HotPath does per step) and land in
    pinned host memory; the host books chunk c (text files, PNG pool) while the GPU is already inside chunk c + 1.
    Returns [(num_evaluated, mean_scores)] in the order of `sequences`.

    Range guard: the split arithmetic's activation formats have a finite exact range (h3: +-4094).  The counters of every chunk are
    copied to pinned memory behind its last network step (evr_model_saturation_async, no synchronisation) and checked BEFORE the
    chunk is booked; a non-zero counter abandons the pass and the whole group is re-run on the model's exact-fp32 twin."""
    try:
        return _eval_method_on_sequences(dataset_name, eval_config, method_name, model, method_config, sequences, metrics)
    except _Saturated as e:
        exact = _rerun_exact(model, e.args[0], ', '.join(q['name'] for q in sequences))
        return _eval_method_on_sequences(dataset_name, eval_config, method_name, exact, method_config, sequences, metrics)


def _eval_method_on_sequences(dataset_name, eval_config, method_name, model, method_config, sequences, metrics):
    from .dataset import SequenceBatch
    import time as _time
    _t = {'setup': -_time.perf_counter(), 'enqueue': 0.0, 'book': 0.0, 'finalize': 0.0}
    S = len(sequences)
    dss = [open_sequence(q) for q in sequences]
    trackers = [get_eval_metrics_tracker(dataset_name, eval_config, method_name, q, metrics) for q in sequences]
    infer_all = eval_config.get('eval_infer_all', False)
    post_norm = method_config.get('post_process_norm', "none")
    norm_in = method_config.get('event_tensor_normalization', False)
    tbs = [ds.table() for ds in dss]
    batch = SequenceBatch(dss)
    H, W, dev = batch.H, batch.W, batch.device
    plans = [_plan_items(ds, tb, q, infer_all) for ds, tb, q in zip(dss, tbs, sequences)]
    # the reference-frame index of every planned item, resident: the chunk loop slices it on the device (see MemMapDataset.frames)
    fidx = [torch.from_numpy(np.asarray(tb['frame_index'][np.asarray(p[0], dtype=np.int64)], dtype=np.int64)).to(dev) if (ds.has_images and len(p[0])) else None
            for ds, tb, p in zip(dss, tbs, plans)]
    model.reset_states()
    steps = max((len(p[0]) for p in plans), default=0)
    # what the frame loop computes for whole chunks (every tracker of a dataset is configured alike)
    pre = trackers[0].wants_precomputed() if all(ds.has_images for ds in dss) else []
    want_u8 = trackers[0].save_images
    lp_model = EvalMetricsTracker._lpips_model() if 'lpips' in pre else None
    gpu_metrics = trackers[0]._gpu
    bufs = [_ChunkBuffers(S, batch.num_bins, H, W, dev, bool(set(pre) & {'mse', 'ssim'}), lp_model is not None, want_u8,
                          all(ds.has_images for ds in dss)) for _ in range(2)]
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)
    policy = _range_guard_policy(eval_config)
    guard = policy != 'off' and hasattr(model, 'saturation_async') and getattr(model, 'arith', 'fp32') != 'fp32'
    warned = [False]
    if guard:
        model._ensure(S, H, W)              # (the counters exist from the first reset on)
        model.saturation(clear=True)        # counters are cumulative: start this group from zero
        for b in bufs:
            b.h_sat = torch.zeros((model.saturation_counters(),), dtype=torch.int32).pin_memory()

    def enqueue(c0, b):
        items = [p[0][c0:c0 + CHUNK] for p in plans]
        if b.ev_done is not None:
            main.wait_event(b.ev_done)              # the evaluation of the chunk that used these buffers has finished
        n = batch.voxel_steps(items, b.grid, b.stats)
        for i in range(n):
            model(b.grid[i], stats=b.stats[i] if norm_in else None, out=b.imgs[i])
        if guard:
            model.saturation_async(b.h_sat)     # lands before ev_model -> before ev_done, which book() waits for
        b.ev_model.record(main)
        b.items, b.n = items, n
        with torch.cuda.stream(side):
            side.wait_event(b.ev_model)
            im = b.imgs[:n].view(n * S, H, W)
            post_process_normalization(im, post_norm)
            if b.refs is not None:
                for j, (ds, it) in enumerate(zip(dss, items)):
                    if it:
                        b.refs[:len(it), j] = ds.frames(fidx[j][c0:c0 + len(it)])[:, 0]
                rf = b.refs[:n].view(n * S, H, W)
                if b.scores is not None:
                    b.scores[:n * S].copy_(gpu_metrics(im, rf, mse='mse' in pre, ssim='ssim' in pre, clip=True))
                    b.h_scores[:n * S].copy_(b.scores[:n * S], non_blocking=True)
                if b.lp is not None:
                    b.lp[:n * S].copy_(lp_model(im, rf, clip=True))
                    b.h_lp[:n * S].copy_(b.lp[:n * S], non_blocking=True)
            if b.u8 is not None:
                b.u8[:n].copy_(torch.round(torch.clamp(b.imgs[:n, :, 0], 0.0, 1.0) * 255))     # eval_utils.py:83
                b.h_u8[:n].copy_(b.u8[:n], non_blocking=True)
            b.ev_done = torch.cuda.Event(); b.ev_done.record(side)

    def book(b):
        b.ev_done.synchronize()
        if guard and bool(b.h_sat.any()):
            if policy == 'rerun':
                raise _Saturated(trackers)          # before anything of this chunk is written
            if not warned[0]:
                warned[0] = True
                model.warn_if_saturated(', '.join(q['name'] for q in sequences) + " (range_guard: 'warn' -- results of the first pass are kept)")
        n = b.n
        sc = b.h_scores[:n * S].numpy().reshape(n, S, 2) if b.h_scores is not None else None
        lp = b.h_lp[:n * S].numpy().reshape(n, S) if b.h_lp is not None else None
        for j in range(S):
            it = b.items[j]
            if not it:
                continue
            tb, ds, k = tbs[j], dss[j], len(it)
            scores = None
            if pre:
                scores = {}
                if 'mse' in pre: scores['mse'] = sc[:k, j, 0].copy()
                if 'ssim' in pre: scores['ssim'] = sc[:k, j, 1].copy()
                if 'lpips' in pre and lp is not None: scores['lpips'] = lp[:k, j].copy()
            u8 = b.h_u8[:k, j] if b.h_u8 is not None else None      # (a strided view of the pinned buffer: the native writers copy it inside the call)
            if ds.has_images:
                refs = b.refs[:k, j] if b.refs is not None else ds.frames(tb['frame_index'][it])[:, 0]
                ref_ts = [float(v) for v in tb['frame_timestamp'][it]]
            else:
                refs, ref_ts = None, None
            trackers[j].update_batch(it, b.imgs[:k, j, 0], refs, [float(v) for v in tb['voxel_timestamp'][it]], ref_ts,
                                     scores=scores, u8=u8)
            for i in it:
                cnt, dt = int(tb['event_count'][i]), float(tb['dt'][i])
                trackers[j].save_custom_metric(i, "event_rate", 0 if (cnt <= 1 or dt == 0) else cnt / dt)

    _t['setup'] += _time.perf_counter()
    prev = None
    for k, c0 in enumerate(range(0, steps, CHUNK)):
        b = bufs[k & 1]
        t0 = _time.perf_counter()
        enqueue(c0, b)
        t1 = _time.perf_counter()
        if prev is not None:
            book(prev)
        _t['enqueue'] += t1 - t0; _t['book'] += _time.perf_counter() - t1
        prev = b
    t1 = _time.perf_counter()
    if prev is not None:
        book(prev)
    _t['book'] += _time.perf_counter() - t1
    main.wait_stream(side)
    t1 = _time.perf_counter()
    out = []
    for j in range(S):
        trackers[j].finalize(plans[j][2], wait_png=not _DEFER_PNG[0])      # (inside evaluate() the dataset loop waits for the PNG writers once)
        out.append((trackers[j].get_num_quan_evaluations(), trackers[j].get_mean_scores()))
    _t['finalize'] = _time.perf_counter() - t1
    TIMINGS.append(dict(_t, frames=sum(len(p[0]) for p in plans), sequences=S))
    del TIMINGS[:-64]
    if os.environ.get('EVR_EVAL_TIMING'):
        print('[evreal_amd.eval] host seconds: ' + ', '.join(f'{k} {v:.3f}' for k, v in _t.items()) + f' ({S} sequences, {steps} steps)', file=sys.stderr)
    batch.raise_if_dropped()
    for j in range(S):      # the reference raises inside the sequence loop: same message, after the files are written
        bad = plans[j][1]
        if bad is not None:
            raise ValueError("WARNING: Event indices {},{} out of bounds 0,{}".format(
                int(tbs[j]['idx0'][bad]), int(tbs[j]['idx1'][bad]), dss[j].num_events))
    return out


def sequence_costs(seqs):
    """Longest-processing-time weights of SURVEY 8e: number of windows x padded pixels per sequence.  Needs only the
    .npy headers and the small per-frame tables (the event columns stay memory-mapped).  Rank 0 computes them and
    broadcasts the vector: every rank must partition by the SAME numbers (a rank-local read failure would otherwise
    give ranks different plans: sequences evaluated twice or not at all); a sequence whose reader fails weighs 1 and
    fails again, loudly, on the rank that owns it."""
    d = _dist()
    if d is None or (d.get_world_size() <= 1 and not force_collectives()) or len(seqs) <= 1:
        return [1] * len(seqs)          # one rank takes everything: nothing to balance
    box = [None]
    if d.get_rank() == 0:
        costs = []
        for s in seqs:
            try:
                costs.append(MemMapDataset(s['sequence_path'], **s['dataset_kwargs']).window_cost())
            except Exception:
                costs.append(1)
        box[0] = costs
    d.broadcast_object_list(box, src=0)
    return list(box[0])


class _SequencePrefetcher:
    """One helper thread that prepares the sequences the main loop will need next (eval_method_with_config): memmap open, window
    tables, the validated host copy of the events.  (Uploading the next sequence's device copy from this thread as well -- a copy
    stream beside the running sequence -- measured SLOWER in round 6: one sequence at a time 2096 -> 1.96-1.98 k frames/s, with PNGs
    1980 -> 1.1-1.7 k; allocations and pageable copies issued beside a full kernel queue wait for it.  The same upload from the MAIN
    thread on a copy stream, under the current sequence's last chunk: no difference, 2124 vs 2114 frames/s -- it is ~1 ms of a 73-ms
    sequence.  The upload stays between the sequences.)"""

    def __init__(self):
        self._thread = None

    @staticmethod
    def _work(seqs):
        for q in seqs:
            try:
                ds = open_sequence(q)
                ds.table()
                ds.host_events(keep=True)
            except Exception:
                return          # the main flow meets the same failure at the same sequence and reports it

    def start(self, seqs):
        import threading
        seqs = [q for q in seqs if 'dataset' not in q]
        if seqs:
            self._thread = threading.Thread(target=self._work, args=(seqs,), daemon=True)
            self._thread.start()

    def wait(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None


def _open_group_hosts(seqs):
    """The first sequences of a dataset: their host-side set-up (31 MB of events each, read and validated) on a few threads at once
    -- numpy releases the GIL in the copies and reductions; failures are left for the main flow to meet in order."""
    import threading
    def one(q):
        try:
            ds = open_sequence(q); ds.table(); ds.host_events(keep=True)
        except Exception:
            pass
    ts = [threading.Thread(target=one, args=(q,), daemon=True) for q in seqs[1:] if 'dataset' not in q]
    for t in ts: t.start()
    if seqs and 'dataset' not in seqs[0]:
        one(seqs[0])
    for t in ts: t.join()


def fold_dataset_metrics(dataset_metrics, metric_names, dist, device=None):
    """Fold one dataset's MetricTracker over the ranks: ONE all-reduce(SUM) of [total, count] per requested metric
    (what MetricTracker.update accumulates, eval.py:259-266).  The column set is the REQUESTED metric list -- identical
    on every rank, including ranks that own no sequence of this dataset -- so every tracked metric (mse, ssim, lpips,
    plug-ins) survives the fold; metrics nobody scored (count 0 everywhere) stay absent, as in a single-rank run."""
    # the trackers key their scores by metric.get_name(), which for pyiqa plug-ins is the requested name lower-cased
    # (eval_metrics.PyIqaMetricFactory.get_metric): fold under the same normalised names on every rank
    names = list(dict.fromkeys(str(nm).lower() for nm in metric_names))
    by_lower = {str(k).lower(): v for k, v in dataset_metrics.data_dict.items()}
    if device is None:
        device = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    sums = torch.zeros((len(names), 2), dtype=torch.float64, device=device)
    for i, nm in enumerate(names):
        d = by_lower.get(nm)
        if d is not None:
            sums[i, 0], sums[i, 1] = d['total'], d['count']
    tot = reduce_metric_sums(sums, dist)
    out = MetricTracker()
    for i, nm in enumerate(names):
        if tot[i, 1] > 0:
            out.data_dict[nm] = {'total': float(tot[i, 0]), 'count': int(round(tot[i, 1])),
                                 'average': float(tot[i, 0] / tot[i, 1])}
    return out


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def eval_method_with_config(eval_config, method_name, datasets, metrics):
    """eval.py:333-377, with the sequences of each dataset sharded across ranks."""
    method_config = get_method_config(method_name)
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
    method_metrics = []
    try:
        model = get_model_from_checkpoint_path(method_config['model_name'], method_config['model_path'])
        if eval_config.get('color', False):
            model = model_arch.ColorNet(model)          # eval.py:346-347
    except Exception as e:
        print(f"Exception while getting method {method_name} from checkpoint path {method_config['model_path']}")
        print(e); print(traceback.format_exc())
        model = None
    collectives = dist is not None and (world > 1 or force_collectives())      # (EVR_FORCE_DIST=1: also on one rank -- the RCCL path on a one-GPU box)
    if collectives:
        # the dataset loop below holds collectives (sequence_costs' broadcast, the per-dataset all-reduce): a rank whose checkpoint
        # load failed locally must not leave the others waiting in them -- agree first, then skip the method everywhere
        ok = torch.tensor([0.0 if model is None else 1.0], device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0 and model is not None:
            print(f"[rank {rank}] method {method_name}: another rank failed to load the checkpoint -- skipped on every rank")
            model = None
    if model is None:
        return method_metrics           # eval.py:348-352: the method is skipped
    for dataset in datasets:
        dataset_metrics = MetricTracker()
        seqs = dataset['sequences']
        try:
            mine = [seqs[i] for i in assign_sequences(sequence_costs(seqs), world)[rank]]
            S = int(eval_config.get('batch_sequences', os.environ.get('EVREAL_BATCH_SEQUENCES', DEFAULT_BATCH_SEQUENCES)))
            k = 0
            # the NEXT group's host-side set-up (memmap open, window tables, the validated host copy of the events) runs on a helper
            # thread while the GPU is inside the current group: at one sequence per group -- the reference's own loop -- that set-up
            # was 8 % of a 160-frame sequence.  Exceptions are swallowed there and raised again, in order, by the main flow below.
            prefetcher = _SequencePrefetcher() if os.environ.get('EVREAL_PREFETCH', '1') != '0' else None
            _DEFER_PNG[0] = True
            if prefetcher is not None and S > 1 and not eval_config.get('color', False):
                _open_group_hosts(mine[:S])
            while k < len(mine):
                if prefetcher is not None:
                    prefetcher.wait()
                # a batch = up to S consecutive sequences of one sensor size; it ends at a sequence whose loader would
                # raise (the reference stops the dataset there).  S = 1 (default) is the reference's own loop.
                group = [mine[k]]
                if S > 1 and not eval_config.get('color', False):
                    res0 = tuple(open_sequence(mine[k]).sensor_resolution)
                    ok0 = bool(open_sequence(mine[k]).table()['valid'].all())
                    try:      # (a first sequence that does not validate runs -- and raises -- alone: no later sequence is opened)
                        open_sequence(mine[k]).host_events(keep=True)
                    except Exception:
                        ok0 = False
                    while ok0 and len(group) < S and k + len(group) < len(mine):
                        nxt = mine[k + len(group)]
                        try:      # a sequence whose files do not validate (coordinates beyond the sensor, polarities other than 0/1)
                            # must not take the healthy sequences before it down: it starts its own group and fails there,
                            # after they have been evaluated and counted, as in the reference's one-at-a-time loop
                            if tuple(open_sequence(nxt).sensor_resolution) != res0:
                                break
                            open_sequence(nxt).host_events(keep=True)
                        except Exception:
                            break           # (evaluated on its own next: raises there with its own message)
                        group.append(nxt)
                        if not bool(open_sequence(nxt).table()['valid'].all()):
                            break
                k += len(group)
                if prefetcher is not None and not eval_config.get('color', False):
                    prefetcher.start(mine[k:k + max(S, 1)])
                for sequence in group:
                    open_sequence(sequence)
                    print(f"[rank {rank}] Evaluating {method_name} with {eval_config['name']} config on "
                          f"{sequence['name']} from {dataset['name']}")
                if len(group) == 1:
                    results = [eval_method_on_sequence(dataset['name'], eval_config, method_name, model, method_config,
                                                       group[0], metrics)]
                else:
                    results = eval_method_on_sequences(dataset['name'], eval_config, method_name, model, method_config,
                                                       group, metrics)
                for num_evaluated, mean_scores in results:
                    for metric_name, score in mean_scores.items():
                        dataset_metrics.update(metric_name, score, num_evaluated)
        except Exception as e:
            print(f"Exception while evaluating method {method_name} on {dataset['name']} dataset:")
            print(e); print(traceback.format_exc())
        finally:
            _DEFER_PNG[0] = False
            try:
                _t_png = __import__('time').perf_counter()
                EvalMetricsTracker.wait_all_pngs()      # the frames the sequence loops left with the native writers: on disk before the dataset is reported
                if os.environ.get('EVR_EVAL_TIMING'):
                    print(f'[evreal_amd.eval] waited {__import__("time").perf_counter() - _t_png:.3f} s for the PNG writers at the end of {dataset["name"]}', file=sys.stderr)
            except Exception as e:
                print(f"Exception while writing the images of method {method_name} on {dataset['name']} dataset:")
                print(e)
            if collectives:
                dataset_metrics = fold_dataset_metrics(dataset_metrics, metrics, dist)
            method_metrics.append(dataset_metrics)
    return method_metrics


def print_scores(all_metrics, method_names, dataset_names, config_name):
    """eval.py:279-303."""
    scores_table, headers = [], ["\nMethod"]
    for method_name, method_metrics in zip(method_names, all_metrics):
        row = []
        for dataset_name, dm in zip(dataset_names, method_metrics):
            for i, metric in enumerate(dm.data_dict):
                if len(scores_table) == 0:
                    headers.append((dataset_name + f' ({dm.get_count(metric)})' + "\n" if i == 0 else "\n") + metric.upper())
                row.append(dm.get_average(metric))
        scores_table.append([method_name] + row)
    print('')
    print(f'Image Quality Scores (for {config_name} config)')
    print(tabulate(scores_table, headers=headers, floatfmt=".3f"))
    print('')


def evaluate(method_names, eval_config_names=None, dataset_names=None, metrics=None):
    """eval.py:413-445."""
    if method_names is None:
        method_names = ['E2VID', 'E2VID+', 'FireNet', 'FireNet+', 'SPADE-E2VID', 'SSL-E2VID', 'ET-Net', 'HyperE2VID']
    eval_config_names = eval_config_names or ['std']
    dataset_names = dataset_names or ['ECD', 'MVSEC', 'HQF']
    metrics = metrics or ['mse', 'ssim', 'lpips']
    results = {}
    dataset_configs = get_dataset_configs(dataset_names)
    for eval_config in get_eval_configs(eval_config_names):
        datasets = get_datasets(dataset_configs, eval_config.get('dataset_kwargs', {}))
        all_metrics = [eval_method_with_config(eval_config, m, datasets, metrics) for m in method_names]
        dist = _dist()
        if dist is None or dist.get_rank() == 0:
            print_scores(all_metrics, method_names, [d['name'] for d in datasets], eval_config['name'])
        results[eval_config['name']] = all_metrics
    return results


def main():
    parser = argparse.ArgumentParser(description='event2im evaluation script (MI355X hot path)')
    parser.add_argument('-c', '--config', nargs='+', type=str, help='evaluation configs')
    parser.add_argument('-m', '--method', nargs='+', type=str, help='methods')
    parser.add_argument('-d', '--dataset', nargs='+', type=str, help='datasets')
    parser.add_argument('-qm', '--metrics', nargs='+', type=str,
                        help='quantitative evaluation metrics that will be used calculate scores')
    parser.add_argument('--batch-sequences', type=int, default=None,
                        help='(extension) advance this many sequences of a dataset together, one batch slot each '
                             '(default 8; 1 = one sequence at a time as the reference; the output files are the same)')
    args = parser.parse_args()
    if args.batch_sequences:
        os.environ['EVREAL_BATCH_SEQUENCES'] = str(args.batch_sequences)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl')
    evaluate(args.method, args.config, args.dataset, args.metrics)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == '__main__':
    main()

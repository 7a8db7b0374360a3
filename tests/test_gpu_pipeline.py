"""The batched per-frame hot loop bench.py times (evreal_amd.pipeline.HotPath): raw events -> voxel grid -> network ->
robust normalisation -> MSE/SSIM/LPIPS, single-stream and with the evaluation half on a second HIP stream."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

W_, H_, BINS, K_EV = 96, 64, 5, 3000


def _inputs(n_seq, n_steps):
    from evreal_amd import synth
    xy = np.empty((n_steps, n_seq, K_EV, 2), np.int16); ts = np.empty((n_steps, n_seq, K_EV), np.float64)
    pol = np.empty((n_steps, n_seq, K_EV), np.uint8); refs = np.empty((n_seq, H_, W_), np.float32)
    for s in range(n_seq):
        t, x, y, p = synth.poisson_events(100 + s, n_steps * K_EV, 2.0e5, W_, H_)
        xy[:, s, :, 0] = x.reshape(n_steps, K_EV); xy[:, s, :, 1] = y.reshape(n_steps, K_EV)
        ts[:, s] = t.reshape(n_steps, K_EV); pol[:, s] = p.reshape(n_steps, K_EV)
        refs[s] = synth.smooth_frames(100 + s, 1, W_, H_)[0, :, :, 0].astype(np.float32) / 255
    offs = (np.arange(n_steps)[:, None] * (n_seq * K_EV) + np.arange(n_seq + 1)[None, :] * K_EV).astype(np.int64)
    d = lambda a: torch.from_numpy(a).cuda()
    return d(xy.reshape(-1, 2)), d(ts.reshape(-1)), d(pol.reshape(-1)), d(offs), d(refs), refs


def _run(overlap, n_seq=3, n_steps=5, read_every_frame=True):
    from evreal_amd import model, weights
    from evreal_amd.lpips import LPIPS
    from evreal_amd.pipeline import HotPath
    kw = dict(weights.E2VID_KWARGS)
    net = model.E2VIDRecurrent(kw)
    net.load_state_dict(weights.synth_state_dict(weights.unet_recurrent_schema(**kw), seed=3))
    lp = LPIPS(weights.synth_lpips_state_dict(seed=0))
    xy, ts, pol, offs, refs, refs_h = _inputs(n_seq, n_steps)
    hp = HotPath(net, BINS, (H_, W_), n_seq, event_tensor_normalization=True, post_process_norm='robust',
                 metrics=('mse', 'ssim', 'lpips'), lpips=lp, overlap=overlap)
    scores = torch.zeros((n_steps, n_seq, 3), dtype=torch.float64, device='cuda')
    imgs = []
    for s in range(n_steps):
        img, _ = hp.step_raw(xy, ts, pol, offs[s], refs, scores[s])
        if read_every_frame:
            hp.flush()                                # the evaluation half of THIS frame (held back for the next frame's gate otherwise)
            torch.cuda.synchronize()                  # (the test reads every frame; bench.py never synchronises)
            imgs.append(img.clone().cpu().numpy())
    hp.flush()
    torch.cuda.synchronize()
    return (np.stack(imgs) if imgs else None), scores.cpu().numpy(), refs_h


def test_gated_evaluation_stream_gives_the_same_scores():
    """Default two-stream flow: the evaluation of frame t is enqueued one step later, behind an event the library records inside
    frame t+1 (after res0.conv2 at this batch size); scores land in the rows they were given, identical to the single-stream run."""
    _, sc1, _ = _run(False, n_steps=7)
    _, sc2, _ = _run(True, n_steps=7, read_every_frame=False)
    np.testing.assert_array_equal(sc1, sc2)


def test_two_stream_step_equals_single_stream_and_the_oracle_metrics():
    from oracle import metrics as omet
    img1, sc1, refs = _run(False)
    img2, sc2, _ = _run(True)
    np.testing.assert_array_equal(img1, img2)
    np.testing.assert_array_equal(sc1, sc2)
    assert np.isfinite(sc1).all() and (sc1[..., 0] > 0).all()
    # MSE / SSIM of the produced (robust-normalised, clipped) frames against the oracle's scikit-image restatement
    for f in (0, img1.shape[0] - 1):
        for s in range(img1.shape[1]):
            a, b = omet.clip01(img1[f, s, 0]), omet.clip01(refs[s])
            assert abs(sc1[f, s, 0] - omet.mse(a, b)) < 1e-6
            assert abs(sc1[f, s, 1] - omet.ssim(a, b)) < 1e-5

"""The model parity suite again under the kernel-selection switches and arithmetic modes the default run does not reach.

The default arithmetic since round 4 is `h3` (three f16 products per term on H2 tensors, fp32-grade; image gate 1e-5, csrc/conv.h).
* EVR_BAND_MIN=1 (+ EVR_BAND_PROG_ALL=0: the implicit GEMM instead of the space-to-depth form for the 128/256-column encoders) -- the band
  kernels (3x3 / 5x5 stride-1 convolutions with the input rows resident in LDS), once with the 128 x 128-tile kernel for every layer
  (EVR_WIDE=0), once with the 256 x 256-tile ConvLSTM kernel and once with its twin form (EVR_WIDE_MIN=1: the golden sequences are
  small, so the fill threshold is lowered).
* EVR_NO_BAND=1 -- the split implicit-GEMM kernels for every layer.
* EVR_FP32=1      -- exact fp32-MFMA arithmetic and PLAIN activations (Winograd F(2x2, 3x3) for the 3x3 stride-1 layers, the transposed
                     decoders -- four phases, the last with the prediction fused -- and the k5 s2 encoders in space-to-depth form; again
                     with EVR_WINO_TCONV=0 EVR_WINO_S2D=0 (decoders / encoders on the direct implicit GEMM) and with EVR_WINO=0).
* EVR_ARITH=mx6   -- the opt-in fast mode: f16 + MX-fp6 cross terms on P6 tensors for the E2VID-type layouts (the others narrow to mx),
                     image gate 1e-4; again on the wide / twin band kernels.
* EVR_ARITH=mx    -- f16 + MX-fp8 cross terms on PACKED tensors for every layout; again on the band kernels and the implicit GEMM.
* EVR_FIRENET_PAD32=1 -- FireNet's trained checkpoints on the 32-channel kernels (default: the unpadded 16-channel h3 kernel), in h3 and mx.
* EVR_C16_ROWS=2 -- FireNet's 16-channel layers on the row-walking kernel (round 6; by default only launches big enough to fill the
                     chip take it), 1 and 2 image rows per step; EVR_C16_ROWS=0 -- the 128-pixel tile kernel everywhere.
The switches are read when the library plans its launches, hence one fresh interpreter per mode.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', 'tests/test_gpu_model.py', 'tests/test_gpu_color.py',
           'tests/test_gpu_eval.py', '-p', 'no:cacheprovider']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_parity_with_band_kernel_on_small_shapes():
    _run({'EVR_BAND_MIN': '1', 'EVR_BAND_PROG_ALL': '0', 'EVR_WIDE': '0'})


def test_parity_on_the_implicit_gemm():
    # since round 3 the band kernels take every eligible launch (no fill threshold): the split implicit-GEMM kernels -- still the
    # path of 1x1 layers, strided / grouped shapes the band forms do not cover, and of EVR_NO_BAND=1 -- get their own pass
    _run({'EVR_NO_BAND': '1'})


def test_parity_with_wide_band_kernel_on_small_shapes():
    # the 256 x 256-tile ConvLSTM kernel the 64-sequence bench runs (conv3x3_wide_kernel), on the golden sequences
    # (EVR_PROG_WIDE_MIN=1: the k5 s2 encoders on the 256-pixel form of the programmed band kernel too, round 5)
    _run({'EVR_BAND_MIN': '1', 'EVR_WIDE_MIN': '1', 'EVR_WIDE': '3', 'EVR_PROG_WIDE_MIN': '1'})


def test_parity_with_twin_band_kernel_on_small_shapes():
    # its two-blocks-per-CU form (256 x 128 tiles, one band buffer), the default for ConvLSTM layers of up to 256 input channels
    _run({'EVR_BAND_MIN': '1', 'EVR_WIDE_MIN': '1', 'EVR_WIDE': '2', 'EVR_PROG_WIDE_MIN': '1'})


def test_parity_with_firenet_row_kernel_on_small_shapes():
    _run({'EVR_C16_ROWS': '2'})


def test_parity_with_firenet_row_kernel_other_step_heights():
    _run({'EVR_C16_ROWS': '2', 'EVR_C16_ROWS_1': '2', 'EVR_C16_ROWS_2': '1'})


def test_parity_with_firenet_tile_kernel_everywhere():
    _run({'EVR_C16_ROWS': '0'})


def test_parity_in_exact_fp32_mode():
    # (since round 6 the 3x3 stride-1 layers of this mode run Winograd F(2x2, 3x3): csrc/wino.hip)
    _run({'EVR_FP32': '1'})


def test_parity_in_exact_fp32_mode_winograd_for_the_plain_3x3_layers_only():
    _run({'EVR_FP32': '1', 'EVR_WINO_TCONV': '0', 'EVR_WINO_S2D': '0'})


def test_parity_in_exact_fp32_mode_direct_form():
    # ... and EVR_WINO=0 keeps them on the direct fp32 implicit GEMM
    _run({'EVR_FP32': '1', 'EVR_WINO': '0'})


def test_parity_in_f16_fp6_mode():
    # the opt-in fast arithmetic (rounds 2-3's default): f16 + MX-fp6 on P6 tensors for the E2VID-type layouts (ConvLSTM, transposed /
    # upsample-conv decoders, 5-bin k5 head), f16 + MX-fp8 on PACKED tensors for every other layout (model.cpp evr_model_create)
    _run({'EVR_ARITH': 'mx6'})


def test_parity_in_f16_fp6_mode_on_the_wide_band_kernel():
    _run({'EVR_ARITH': 'mx6', 'EVR_BAND_MIN': '1', 'EVR_WIDE_MIN': '1', 'EVR_WIDE': '3'})


def test_parity_in_f16_fp6_mode_on_the_twin_band_kernel():
    _run({'EVR_ARITH': 'mx6', 'EVR_BAND_MIN': '1', 'EVR_WIDE_MIN': '1', 'EVR_WIDE': '2'})


def test_parity_in_f16_fp8_mode():
    # EVR_ARITH=mx puts every layout on the fp8 form
    _run({'EVR_ARITH': 'mx'})


def test_parity_in_f16_fp8_mode_on_the_band_kernels():
    _run({'EVR_ARITH': 'mx', 'EVR_BAND_MIN': '1', 'EVR_WIDE_MIN': '1', 'EVR_WIDE': '3'})


def test_parity_in_f16_fp8_mode_on_the_implicit_gemm():
    _run({'EVR_ARITH': 'mx', 'EVR_NO_BAND': '1'})


def _run_firenet(env_extra):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', 'tests/test_gpu_model.py', 'tests/test_gpu_fullsize.py', 'tests/test_gpu_eval.py',
           '-k', 'firenet or evaluate or eval_loop', '-p', 'no:cacheprovider']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]


def test_fp32_grade_arithmetic_on_trained_weights_padded_to_32_channels():
    """The only TRAINED checkpoints available offline are the shipped FireNet / FireNet+ models (16 channels: the unpadded
    three-f16-product kernel by default).  EVR_FIRENET_PAD32=1 zero-pads their tensors to one 32-channel chunk, which puts every layer on
    the 32-channel-chunk kernels the E2VID-type networks run: the real-weight goldens (images from the reference classes), the 40-frame
    run, the 240x180 k_events sequence and the eval-loop goldens must still pass -- in the default h3 arithmetic at 1e-5 ..."""
    _run_firenet({'EVR_FIRENET_PAD32': '1'})


def test_fast_arithmetic_on_trained_weights():
    """... and in the f16 + MX-fp8 arithmetic at 1e-4 (mx6 narrows to mx for this layout: ConvGRU's epilogue writes 4-channel runs)."""
    _run_firenet({'EVR_FIRENET_PAD32': '1', 'EVR_ARITH': 'mx6'})


def test_drift_100_frames_in_the_fast_mode():
    """100 frames x 8 sequences at 346x260 in the f16 + MX-fp6 mode on the kernels the 64-sequence bench runs, gate 1e-4 per pixel
    (measured 4e-6).  The GPU advances all 8 sequences; sequences 0 and 7 are replayed through the CPU oracle (the default-mode run in
    test_gpu_fullsize.py -- h3, gate 1e-5 -- replays all 8)."""
    env = dict(os.environ, EVR_ARITH='mx6', EVR_WIDE_MIN='1', EVR_PROG_WIDE_MIN='1', EVR_TEST_DRIFT_ORACLE_SEQS='0,7')
    cmd = [sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', 'tests/test_gpu_fullsize.py', '-k', 'drift_100', '-p', 'no:cacheprovider']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and '1 passed' in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]


def test_drift_100_frames_with_wide_band_kernel():
    """The 100-frame, 8-sequence, 346x260 drift gate again in the default arithmetic with the ConvLSTM gates on the kernel the 64-sequence
    bench times (8 sequences alone stay below its 1024-block threshold).  Sequences 0 and 7 against the oracle, as above."""
    env = dict(os.environ, EVR_WIDE_MIN='1', EVR_WIDE='1', EVR_PROG_WIDE_MIN='1', EVR_TEST_DRIFT_ORACLE_SEQS='0,7')      # (EVR_WIDE=1, the default: twin AND 256 x 256 forms)
    cmd = [sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', 'tests/test_gpu_fullsize.py', '-k', 'drift_100', '-p', 'no:cacheprovider']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and '1 passed' in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]

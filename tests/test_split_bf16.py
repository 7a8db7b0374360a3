"""The split-bf16 (PACKED) storage codec of the default arithmetic mode, checked on the CPU against numpy.

include/evreal_hip.h: every 8 values -> 8 bf16 'hi' halves (RNE) then 8 bf16 'lo' halves (RNE of value - hi).
"""
import ctypes

import numpy as np

from evreal_amd import lib as _lib


def _bf16_rne_bits(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def _bf16_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def _ref_pack(x):
    x = x.astype(np.float32).reshape(-1, 8)
    hi = _bf16_rne_bits(x)
    lo = _bf16_rne_bits(x - _bf16_to_f32(hi))
    return np.concatenate([hi, lo], axis=1).reshape(-1).view(np.float32)


def _call(name, src):
    L = _lib.load()          # host-only entry points: no GPU needed
    dst = np.empty_like(src)
    rc = getattr(L, name)(src.ctypes.data_as(ctypes.c_void_p), dst.ctypes.data_as(ctypes.c_void_p), src.size)
    assert rc == 0, L.evr_last_error()
    return dst


def test_pack_matches_numpy_bit_for_bit():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4096) * 10.0 ** rng.integers(-6, 6, 4096), [0.0, -0.0, 1.0, -1.0, 3.0e38, 1e-30, 0.5, 255.0]]).astype(np.float32)
    got = _call('evr_split_bf16_pack', x)
    np.testing.assert_array_equal(got.view(np.uint32), _ref_pack(x).view(np.uint32))


def test_roundtrip_keeps_16_significant_bits():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(8192) * 10.0 ** rng.integers(-4, 4, 8192)).astype(np.float32)
    y = _call('evr_split_bf16_unpack', _call('evr_split_bf16_pack', x))
    rel = np.abs(y - x) / np.maximum(np.abs(x), 1e-30)
    assert rel.max() <= 2.0 ** -16, rel.max()          # hi carries 8 bits, lo 8 more (+ sign of lo): <= 2^-17 typical
    assert np.median(rel) < 2.0 ** -18
    # values with <= 16 significant bits survive exactly
    z = (rng.integers(-32768, 32768, 4096).astype(np.float32) * np.float32(2.0) ** rng.integers(-20, 20, 4096)).astype(np.float32)
    np.testing.assert_array_equal(_call('evr_split_bf16_unpack', _call('evr_split_bf16_pack', z)), z)

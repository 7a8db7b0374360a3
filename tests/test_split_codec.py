"""The PACKED (split) storage codec of the default arithmetic mode, checked on the CPU against numpy / torch dtypes.

include/evreal_hip.h: every 16 values -> 16 f16 'hi' (RNE, saturating) | 16 e4m3 'lo8' = RNE((v - hi) * 2^12) | 16 e4m3 'x8' = RNE(v).
"""
import ctypes

import numpy as np
import torch

from evreal_amd import lib as _lib


def _e4m3_bits(x):
    t = torch.from_numpy(np.clip(x, -448.0, 448.0).astype(np.float32)).to(torch.float8_e4m3fn)
    return t.view(torch.uint8).numpy()


def _ref_pack(x):
    x = x.astype(np.float32).reshape(-1, 16)
    c = np.clip(x, -65504.0, 65504.0)
    hi = c.astype(np.float16)
    lo8 = _e4m3_bits((c - hi.astype(np.float32)) * np.float32(4096.0))
    x8 = _e4m3_bits(x)
    out = np.concatenate([hi.view(np.uint8).reshape(-1, 32), lo8.reshape(-1, 16), x8.reshape(-1, 16)], axis=1)
    return out.reshape(-1).view(np.float32)


def _call(name, src):
    L = _lib.load()          # host-only entry points: no GPU needed
    dst = np.empty_like(src)
    rc = getattr(L, name)(src.ctypes.data_as(ctypes.c_void_p), dst.ctypes.data_as(ctypes.c_void_p), src.size)
    assert rc == 0, L.evr_last_error()
    return dst


def test_pack_matches_numpy_and_torch_dtypes_bit_for_bit():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4096) * 10.0 ** rng.integers(-9, 4, 4096),
                        [0.0, -0.0, 1.0, -1.0, 3.0e38, 1e-30, 0.5, 255.0, 447.9, 448.0, 449.0, 65504.0, 65519.0, 7e4, 2.0 ** -14, 2.0 ** -15]]).astype(np.float32)
    got = _call('evr_split_pack', x)
    np.testing.assert_array_equal(got.view(np.uint8), _ref_pack(x).view(np.uint8))


def test_roundtrip_keeps_15_significant_bits():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(8192) * 10.0 ** rng.integers(-3, 2, 8192)).astype(np.float32)
    x = x[np.abs(x) > 2.0 ** -6][:8192 // 16 * 8].copy()
    x = np.resize(x, (x.size // 16) * 16)
    y = _call('evr_split_unpack', _call('evr_split_pack', x))
    rel = np.abs(y - x) / np.abs(x)
    assert rel.max() <= 2.0 ** -15, rel.max()      # hi: 11 bits (ulp/2 = 2^-12 rel. at most), lo8: 4 more
    assert np.median(rel) < 2.0 ** -17
    # values a half holds survive exactly
    z = (rng.integers(-2048, 2048, 4096).astype(np.float32) * np.float32(2.0) ** rng.integers(-12, 4, 4096)).astype(np.float32)
    np.testing.assert_array_equal(_call('evr_split_unpack', _call('evr_split_pack', z)), z)


def test_weight_packing_matches_numpy_and_keeps_both_fp8_pieces_in_range():
    L = _lib.load()
    rng = np.random.default_rng(2)
    for scale in (1e-3, 0.05, 1.0, 7.0, 300.0):
        w = (rng.uniform(-1, 1, 4096) * scale).astype(np.float32)
        w[::97] = 0.0
        got = np.empty_like(w); e = ctypes.c_int(0)
        assert L.evr_split_pack_weights(w.ctypes.data_as(ctypes.c_void_p), got.ctypes.data_as(ctypes.c_void_p), w.size, ctypes.byref(e)) == 0
        ex = e.value
        mx = float(np.abs(w).max())
        assert mx * 2.0 ** ex <= 224.0 < mx * 2.0 ** (ex + 1)
        g = w.reshape(-1, 16)
        hi = g.astype(np.float16)
        w8 = _e4m3_bits(g * np.float32(2.0 ** ex))
        wlo8 = _e4m3_bits((g - hi.astype(np.float32)) * np.float32(2.0 ** (ex + 12)))
        want = np.concatenate([hi.view(np.uint8).reshape(-1, 32), w8.reshape(-1, 16), wlo8.reshape(-1, 16)], axis=1).reshape(-1)
        np.testing.assert_array_equal(got.view(np.uint8), want)
        # neither fp8 piece saturates: decoding hi + wlo8 * 2^-(e+12) recovers w to 2^-15 of the tensor's largest weight
        t8 = torch.from_numpy(wlo8.copy()).view(torch.float8_e4m3fn).float().numpy().reshape(-1, 16)
        back = hi.astype(np.float32) + t8 * np.float32(2.0 ** -(ex + 12))
        assert np.abs(back - g).max() <= mx * 2.0 ** -15


def test_h2_codec_matches_numpy_and_keeps_22_bits():
    """The fp32-grade mode's H2 format (EVR_ARITH=h3): hi = f16(v 2^e), lo = f16(v 2^e - hi), both IEEE halves (subnormals kept:
    the f16 MFMA honours them, tools/mfma_denorm_probe.hip).  Bit-for-bit against numpy's float16, then the precision claim."""
    L = _lib.load()
    a = L.evr_h2_act_exponent()
    assert a == 4
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(4096) * 10.0 ** rng.integers(-8, 3, 4096),
                        [0.0, -0.0, 1.0, -1.0, 4093.9, 4094.0, 4095.0, 1e5, -1e5, 2.0 ** -20, 2.0 ** -28, 3e-9, 0.1, 255.0, 1e-30, 65504.0]]).astype(np.float32)
    got = _call('evr_h2_pack', x)
    c = np.clip(x.reshape(-1, 16) * np.float32(2.0 ** a), -65504.0, 65504.0).astype(np.float32)
    hi = c.astype(np.float16)
    lo = (c - hi.astype(np.float32)).astype(np.float16)
    want = np.concatenate([hi.view(np.uint8).reshape(-1, 32), lo.view(np.uint8).reshape(-1, 32)], axis=1).reshape(-1)
    np.testing.assert_array_equal(got.view(np.uint8), want)
    # decode: 22 significant bits wherever lo is a normal half (|v| >= 2^-7), an absolute 2^-29 below, clamp at +-4094
    y = np.empty_like(x)
    assert L.evr_h2_unpack(got.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), x.size, a) == 0
    inside = np.abs(x) <= 4094.0
    big = inside & (np.abs(x) >= 2.0 ** -7)
    assert (np.abs(y[big] - x[big]) / np.abs(x[big])).max() <= 2.0 ** -21
    assert np.abs(y[inside & ~big] - x[inside & ~big]).max() <= 2.0 ** -28
    assert np.abs(y[~inside]).max() <= 65504.0 / 16 + 1
    # weights: per-tensor exponent brings the largest magnitude to [2^13, 2^14); every weight keeps 2^-21 of it or better
    for scale in (1e-3, 0.05, 1.0, 300.0):
        w = (rng.uniform(-1, 1, 4096) * scale).astype(np.float32)
        got = np.empty_like(w); e = ctypes.c_int(0)
        assert L.evr_h2_pack_weights(w.ctypes.data_as(ctypes.c_void_p), got.ctypes.data_as(ctypes.c_void_p), w.size, ctypes.byref(e)) == 0
        mx = float(np.abs(w).max())
        assert 2.0 ** 13 <= mx * 2.0 ** e.value < 2.0 ** 14
        back = np.empty_like(w)
        assert L.evr_h2_unpack(got.ctypes.data_as(ctypes.c_void_p), back.ctypes.data_as(ctypes.c_void_p), w.size, e.value) == 0
        nz = np.abs(w) > mx * 2.0 ** -10
        assert (np.abs(back[nz] - w[nz]) / np.abs(w[nz])).max() <= 2.0 ** -21
        assert np.abs(back - w).max() <= mx * 2.0 ** -31 * 2 ** 10


def test_fastdiv_constants_divide_exactly():
    L = _lib.load()
    rng = np.random.default_rng(3)
    ds = [1, 2, 3, 5, 7, 44, 88, 176, 346, 352, 1452, 5808, 23232, 92928, 307200, (1 << 20) + 1, (1 << 31) - 1] + [int(v) for v in rng.integers(1, 1 << 22, 500)]
    for d in ds:
        mul, sh = ctypes.c_uint(0), ctypes.c_uint(0)
        assert L.evr_fastdiv_magic(d, ctypes.byref(mul), ctypes.byref(sh)) == 0
        n = np.concatenate([[0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 31) - 1, (1 << 31) - 2], rng.integers(0, 1 << 31, 300)]).astype(np.uint64)
        n = n[n < (1 << 31)]
        q = (((n * np.uint64(mul.value)) >> np.uint64(32)) + n) >> np.uint64(sh.value)
        np.testing.assert_array_equal(q, n // np.uint64(d), err_msg=str(d))


# ---- P6: the f16 + MX-fp6 mode's format (EVR_ARITH=mx6) ------------------------------------------------------------------
_E2M3 = np.array([m * 0.125 if m < 8 else (8 + (m & 7)) * 2.0 ** ((m >> 3) - 4) for m in range(32)], np.float64)   # the 32 magnitudes


def _e2m3_codes(v):
    """Round to nearest (ties to the even code), saturating at 7.5 -- an independent numpy statement of the OCP e2m3 rounding."""
    a = np.abs(v.astype(np.float64))
    i = np.clip(np.searchsorted(_E2M3, a, side='left'), 1, 31)
    lo, hi = _E2M3[i - 1], _E2M3[i]
    pick_hi = (a - lo > hi - a) | ((a - lo == hi - a) & ((i & 1) == 0))
    code = np.where(pick_hi, i, i - 1)
    code = np.where(a >= 7.5, 31, np.where(a == 0, 0, code))
    return (code | (np.signbit(v).astype(np.int64) << 5)).astype(np.uint8)


def _ref_pack_p6(x, e=0, weights=False):
    c = np.clip(np.ldexp(x.astype(np.float32).reshape(-1, 16), e), -65504.0, 65504.0).astype(np.float32)
    hi = c.astype(np.float16)
    lo = ((c - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float32)
    mx = np.abs(c).max(axis=1)
    eb = (mx.view(np.uint32) >> 23).astype(np.int64)
    eb = np.where(eb > 3, eb - 2, 1)
    s = np.ldexp(np.float64(1.0), eb - 127)[:, None]
    cv, cl = _e2m3_codes(c / s), _e2m3_codes(lo / s)
    codes = np.empty((c.shape[0], 32), np.uint8)
    codes[:, 0::2] = cl if weights else cv
    codes[:, 1::2] = cv if weights else cl
    bits = np.zeros((c.shape[0], 24), np.uint8)
    for j in range(32):
        v16 = codes[:, j].astype(np.uint16) << ((6 * j) & 7)
        bits[:, (6 * j) >> 3] |= (v16 & 0xff).astype(np.uint8)
        if ((6 * j) & 7) > 2:
            bits[:, ((6 * j) >> 3) + 1] |= (v16 >> 8).astype(np.uint8)
    tail = np.zeros((c.shape[0], 8), np.uint8)
    tail[:, 0] = np.maximum(eb - 11, 1) if weights else eb
    return np.concatenate([hi.view(np.uint8).reshape(-1, 32), bits, tail], axis=1).reshape(-1)


def test_p6_codec_matches_numpy_and_keeps_the_split_precision():
    """hi = f16(v) | 32 e2m3 codes [v/S, (v - hi) 2^11 / S] with the group's scale S = 2^(E - 2) | the E8M0 byte: bit for bit against
    numpy, then the precision the arithmetic relies on -- decoded activations within 2^-16 of the group's largest value."""
    L = _lib.load()
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.standard_normal(8192) * 10.0 ** rng.integers(-9, 4, 8192),
                        rng.standard_normal(4096) * 3.0,                                  # groups of similar magnitudes
                        [0.0, -0.0, 1.0, -1.0, 3.0e38, 1e-30, 0.5, 255.0, 7.5, 7.75, 8.0, 65504.0, 65519.0, 7e4, 2.0 ** -14, 2.0 ** -15],
                        np.zeros(16)]).astype(np.float32)
    got = _call('evr_p6_pack', x)
    np.testing.assert_array_equal(got.view(np.uint8), _ref_pack_p6(x))
    y = np.empty_like(x)
    assert L.evr_p6_unpack(got.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), x.size) == 0
    xs = np.clip(x, -65504.0, 65504.0).reshape(-1, 16); ys = y.reshape(-1, 16)
    gmax = np.abs(xs).max(axis=1, keepdims=True)
    ok = gmax[:, 0] >= 2.0 ** -10                          # (below that the f16 half is subnormal: absolute 2^-25)
    assert (np.abs(ys[ok] - xs[ok]) / gmax[ok]).max() <= 2.0 ** -16
    assert np.abs(ys[~ok] - xs[~ok]).max() <= 2.0 ** -24
    # weights: values times 2^e (largest in [2^13, 2^14)), code pairs swapped, scale byte lowered by 11
    for scale in (1e-3, 0.05, 1.0, 300.0):
        w = (rng.uniform(-1, 1, 4096) * scale).astype(np.float32)
        got = np.empty_like(w); e = ctypes.c_int(0)
        assert L.evr_p6_pack_weights(w.ctypes.data_as(ctypes.c_void_p), got.ctypes.data_as(ctypes.c_void_p), w.size, ctypes.byref(e)) == 0
        assert 2.0 ** 13 <= float(np.abs(w).max()) * 2.0 ** e.value < 2.0 ** 14
        np.testing.assert_array_equal(got.view(np.uint8), _ref_pack_p6(w, e.value, weights=True))

"""CPU: the host-side uint8 parameter builders (onnxstream_amd/csrc/host/qu8.h: quantisation parameters, dynamic input quantisation, Sigmoid /
InstanceNormalization / Softmax tables, XNNPACK's qu8 add multipliers, fp32 requantisation) against oracle/np_qu8.py -- which is itself
pinned code for code against the reference's own uint8 run (tests/test_qu8_oracle.py).  Bit-exact, random parameters."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import np_qu8 as Q

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u8p, f32p, i32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)


@pytest.fixture(scope="module")
def hooks():
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "libqu8_hooks.so")
        subprocess.check_call(["g++", "-std=c++20", "-O2", "-shared", "-fPIC", "-I", os.path.join(REPO, "onnxstream_amd", "csrc", "host"),
                               os.path.join(REPO, "tests", "stub", "qu8_host_hooks.cpp"), "-o", so])
        yield ctypes.CDLL(so)


def _p(a, t):
    return a.ctypes.data_as(t)


def _qp(rng):
    lo, hi = -abs(rng.normal()) * 4 - 0.1, abs(rng.normal()) * 4 + 0.1
    return Q.range_to_scale(lo, hi)


def test_range_to_scale_and_dynamic_quantisation(hooks):
    rng = np.random.default_rng(1)
    for _ in range(200):
        lo, hi = sorted(rng.normal(size=2) * 5)
        s, z = ctypes.c_float(), ctypes.c_int()
        hooks.h_range_to_scale(ctypes.c_float(lo), ctypes.c_float(hi), ctypes.byref(s), ctypes.byref(z))
        ws, wz = Q.range_to_scale(np.float32(lo), np.float32(hi))
        assert np.float32(s.value) == ws and z.value == wz
    for n, thr in [(16384 * 3 + 17, 1), (16384 * 3 + 17, 4), (1000, 7), (35, 3)]:
        x = (rng.standard_normal(n) * 3).astype(np.float32)
        out = np.empty(n, np.uint8)
        s, z = ctypes.c_float(), ctypes.c_int()
        assert hooks.h_quantize_dynamic(_p(x, f32p), ctypes.c_size_t(n), ctypes.c_size_t(thr), _p(out, u8p), ctypes.byref(s), ctypes.byref(z)) == 0
        q, ws, wz = Q.quantize_dynamic(x, threads=thr)
        assert np.float32(s.value) == ws and z.value == wz and np.array_equal(out, q)


def test_sigmoid_table_add_mul_requant_bias(hooks):
    rng = np.random.default_rng(2)
    codes = np.arange(256, dtype=np.uint8)
    for _ in range(50):
        (si, zi), (so, zo) = _qp(rng), Q.range_to_scale(0.0, 1.0)
        lut = np.empty(256, np.uint8)
        hooks.h_sigmoid_lut(ctypes.c_float(si), zi, ctypes.c_float(so), zo, _p(lut, u8p))
        assert np.array_equal(lut, Q.sigmoid_u8(codes, si, zi, so, zo))
    n = 20000
    for _ in range(30):
        a, b = rng.integers(0, 256, n, dtype=np.uint8), rng.integers(0, 256, n, dtype=np.uint8)
        (sa, za), (sb, zb), (so, zo) = _qp(rng), _qp(rng), _qp(rng)
        y = np.empty(n, np.uint8)
        hooks.h_add(_p(a, u8p), ctypes.c_float(sa), za, _p(b, u8p), ctypes.c_float(sb), zb, ctypes.c_float(so), zo, ctypes.c_size_t(n), _p(y, u8p))
        assert np.array_equal(y, Q.add_u8(a, sa, za, b, sb, zb, so, zo))
        hooks.h_mul(_p(a, u8p), ctypes.c_float(sa), za, _p(b, u8p), ctypes.c_float(sb), zb, ctypes.c_float(so), zo, ctypes.c_size_t(n), _p(y, u8p))
        assert np.array_equal(y, Q.mul_u8(a, sa, za, b, sb, zb, so, zo))
        acc = rng.integers(-2_000_000, 2_000_000, n, dtype=np.int32)
        hooks.h_requant(_p(acc, i32p), ctypes.c_float(sa), ctypes.c_float(sb), ctypes.c_float(so), zo, ctypes.c_size_t(n), _p(y, u8p))
        scale = np.float32(np.float32(sa) * np.float32(sb)) / np.float32(so)
        assert np.array_equal(y, Q.requant_fp32(acc, scale, zo))
        bias = (rng.standard_normal(64) * 0.1).astype(np.float32)
        bi = np.empty(64, np.int32)
        hooks.h_conv_bias(_p(bias, f32p), ctypes.c_float(sa), ctypes.c_float(sb), ctypes.c_size_t(64), _p(bi, i32p))
        assert np.array_equal(bi, Q.conv_bias_i32(bias, sa, sb))


def test_softmax_and_instance_norm_tables(hooks):
    rng = np.random.default_rng(3)
    for rows, ch in [(64, 256), (5, 77), (3, 1024)]:
        x = rng.integers(0, 256, (rows, ch), dtype=np.uint8)
        s_in = np.float32(abs(rng.normal()) * 0.05 + 0.005)
        y = np.empty_like(x)
        hooks.h_softmax(_p(x, u8p), ctypes.c_float(s_in), ctypes.c_size_t(rows), ctypes.c_size_t(ch), _p(y, u8p))
        want, so, zo = Q.softmax_u8(x, s_in)
        assert np.array_equal(y, want) and so == np.float32(1 / 256) and zo == 0
    for C, L in [(8, 1024), (4, 4096), (3, 77)]:
        x = np.clip(rng.normal(128, 30, (1, C, L)), 0, 255).astype(np.uint8)
        (si, zi), (so, zo) = _qp(rng), _qp(rng)
        sc, bi = (1 + rng.standard_normal(C) * 0.1).astype(np.float32), (rng.standard_normal(C) * 0.1).astype(np.float32)
        y = np.empty_like(x)
        hooks.h_instance_norm(_p(x, u8p), ctypes.c_size_t(C), ctypes.c_size_t(L), ctypes.c_float(si), zi, _p(sc, f32p), _p(bi, f32p),
                              ctypes.c_float(1e-5), ctypes.c_float(so), zo, _p(y, u8p))
        assert np.array_equal(y, Q.instance_norm_u8(x, si, zi, sc, bi, 1e-5, so, zo))


def test_parallel_percentiles_equal_the_sequential_ones(hooks):
    """percentiles_fast (chunks spread over host threads, nth_element instead of sort; used by the calibration pass) == percentiles"""
    rng = np.random.default_rng(11)
    hooks.h_percentiles.argtypes = [f32p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, f32p, f32p]
    for n, threads in ((100, 1), (16384, 1), (50000, 3), (200001, 8), (1 << 20, 16), (7, 4)):
        x = (rng.standard_normal(n) * rng.uniform(0.1, 30)).astype(np.float32)
        if n > 1000:
            x[rng.integers(0, n, 5)] = np.inf
        want = Q.percentiles(x, 0.001, 0.001, threads)
        res = []
        for workers in (0, 1, 5):
            lo, hi = ctypes.c_float(), ctypes.c_float()
            rc = hooks.h_percentiles(_p(x, f32p), n, threads, workers, ctypes.byref(lo), ctypes.byref(hi))
            res.append(None if rc else (lo.value, hi.value))
        assert res[0] == res[1] == res[2]
        assert (res[0] is None) == (want is None)
        if want is not None:
            assert res[0] == (float(want[0]), float(want[1]))

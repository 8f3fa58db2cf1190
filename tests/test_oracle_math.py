"""Oracle pinning, part 2: the deterministic transcendentals (oracle/detmath.h) against the platform
libm that rustlight itself would call through Rust's f32::{sin,cos,exp,ln,powf}.  CPU only."""
import ctypes
import ctypes.util

import numpy as np

from oracle import orc
from rustlight_amd import abi

libm = ctypes.CDLL(ctypes.util.find_library("m"))
for f in ("sinf", "cosf", "expf", "logf", "acosf"):
    getattr(libm, f).restype = ctypes.c_float
    getattr(libm, f).argtypes = [ctypes.c_float]
libm.powf.restype = ctypes.c_float
libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
libm.atan2f.restype = ctypes.c_float
libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]


def batch(fn, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, np.float32)
    out = np.zeros_like(a)
    orc.lib().orc_math_batch(fn, a.shape[0], abi.fptr(a), abi.fptr(b), abi.fptr(out))
    return out


def ulp_diff(x, y):
    xi = np.asarray(x, np.float32).view(np.int32).astype(np.int64)
    yi = np.asarray(y, np.float32).view(np.int32).astype(np.int64)
    xi = np.where(xi < 0, -(xi & 0x7fffffff), xi)
    yi = np.where(yi < 0, -(yi & 0x7fffffff), yi)
    return np.abs(xi - yi)


def check(fn, name, a, b=None, min_equal=0.97):
    got = batch(fn, a, b)
    if b is None:
        ref = np.array([getattr(libm, name)(float(x)) for x in a], np.float32)
        exact = getattr(np, {"sinf": "sin", "cosf": "cos", "expf": "exp", "logf": "log", "acosf": "arccos"}[name])(a.astype(np.float64))
    else:
        ref = np.array([getattr(libm, name)(float(x), float(y)) for x, y in zip(a, b)], np.float32)
        exact = np.power(a.astype(np.float64), b.astype(np.float64)) if name == "powf" else np.arctan2(a.astype(np.float64), b.astype(np.float64))
    d = ulp_diff(got, ref)
    assert d.max() <= 1, (name, d.max())
    assert (d == 0).mean() >= min_equal, (name, (d == 0).mean())
    # and it is the correctly rounded value (f64 numpy result rounded to f32) almost everywhere
    cr = exact.astype(np.float32)
    assert (ulp_diff(got, cr) == 0).mean() >= 0.9999, name


def test_sin_cos(built):
    rng = np.random.default_rng(1)
    a = np.concatenate([rng.uniform(-np.pi, 2 * np.pi, 20000), rng.uniform(-0.8, 2.4, 20000), [0.0, 1e-8, -1e-8, np.pi / 4, np.pi / 2]]).astype(np.float32)
    check(0, "sinf", a, min_equal=0.97)
    check(1, "cosf", a, min_equal=0.97)


def test_exp_log_pow(built):
    rng = np.random.default_rng(2)
    check(2, "expf", rng.uniform(-60, 20, 20000).astype(np.float32), min_equal=0.97)
    check(3, "logf", np.exp(rng.uniform(-40, 40, 20000)).astype(np.float32), min_equal=0.97)
    x = rng.uniform(0, 1, 20000).astype(np.float32)
    y = rng.uniform(0.01, 200, 20000).astype(np.float32)
    check(4, "powf", x, y, min_equal=0.97)


def test_acos_atan2(built):
    rng = np.random.default_rng(3)
    check(5, "acosf", rng.uniform(-1, 1, 20000).astype(np.float32), min_equal=0.9)   # glibc acosf itself is only ~92 % correctly rounded
    check(6, "atan2f", rng.uniform(-3, 3, 20000).astype(np.float32), rng.uniform(-3, 3, 20000).astype(np.float32), min_equal=0.8)


def test_special_values(built):
    assert batch(2, [-200.0])[0] == 0.0 and np.isinf(batch(2, [100.0])[0])
    assert np.isneginf(batch(3, [0.0])[0]) and np.isnan(batch(3, [-1.0])[0])
    assert batch(4, [0.0], [2.0])[0] == 0.0 and batch(4, [0.3], [0.0])[0] == 1.0 and batch(4, [1.0], [77.0])[0] == 1.0
    assert np.isnan(batch(0, [np.inf])[0])

"""The libm functions on the path, as the DEVICE computes them (hprt_math.h, evaluated by a probe kernel), against the oracle's
restatements and against the host's glibc: sinf / cosf (Halton-driven direction sampling), acosf / atan2f (sphere and
environment parameterisations), logf (texture level of detail) and the double sin / cos of TrowbridgeReitzSample11's
normal-incidence branch (core/microfacet.cpp:243-245).  tests/test_oracle_pins.py holds the oracle's restatements against libm
exhaustively on the CPU; this file closes the loop on the GPU: 2^24 arguments per function — a stride through all float bit
patterns of the domain plus the arguments the path actually forms (2 pi u, u a multiple of 2^-24) — bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1 << 24


def _device(hprt, fn, x, y):
    o0 = np.empty(x.size, np.float64); o1 = np.empty(x.size, np.float64)
    f = hprt.lib.hprt_debug_device_math
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    rc = f(0, fn, x.ctypes.data, y.ctypes.data, x.size, o0.ctypes.data, o1.ctypes.data)
    assert rc == 0, hprt.lib.hprt_last_error()
    return o0, o1


def _oracle(orc, fn, x, y, libm):
    o0 = np.empty(x.size, np.float64); o1 = np.empty(x.size, np.float64)
    f = orc.lib.orc_math_eval
    f.restype = None
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    f(fn, int(libm), x.ctypes.data, y.ctypes.data, x.size, o0.ctypes.data, o1.ctypes.data)
    return o0, o1


def _patterns(lo, hi, n, seed):
    """n float bit patterns spread over [lo, hi] (bit-pattern order = value order for positive floats), both signs."""
    rng = np.random.default_rng(seed)
    a, b = np.float32(lo).view(np.uint32), np.float32(hi).view(np.uint32)
    u = (a + (rng.integers(0, int(b) - int(a) + 1, n, dtype=np.int64))).astype(np.uint32)
    return u.view(np.float32)


@pytest.mark.parametrize("fn,name", [(0, "sinf_cosf"), (1, "acosf"), (2, "atan2f"), (3, "logf"), (4, "sin_cos_double")])
def test_device_libm_restatements(hprt, orc, fn, name):
    rng = np.random.default_rng(fn)
    twopi_u = (np.float32(2) * np.float32(3.14159274101257324219) * (rng.integers(0, 1 << 24, N // 2).astype(np.float32) * np.float32(2.0 ** -24))).astype(np.float32)
    if fn == 0:
        x = np.concatenate([_patterns(1e-30, 119.9, N // 4, 1), -_patterns(1e-30, 119.9, N // 4, 2), twopi_u]); y = np.zeros_like(x)
    elif fn == 1:
        x = np.concatenate([_patterns(1e-30, 1.0, N // 2, 3), -_patterns(1e-30, 1.0, N // 2, 4)]); y = np.zeros_like(x)
    elif fn == 2:
        x = np.concatenate([_patterns(1e-20, 1e20, N // 2, 5), -_patterns(1e-20, 1e20, N // 2, 6)])
        y = np.concatenate([_patterns(1e-20, 1e20, N // 2, 7), -_patterns(1e-20, 1e20, N // 2, 8)])[rng.permutation(N)]
    elif fn == 3:
        x = _patterns(1e-45, 3e38, N, 9); y = np.zeros_like(x)
    else:
        phi = (6.28318530718 * (rng.integers(0, 1 << 24, N // 2).astype(np.float64) * 2.0 ** -24)).astype(np.float32)     # Float phi = 6.28318530718 * U2
        x = np.concatenate([_patterns(1e-45, 6.2831850051879883, N // 2, 10), phi]); y = np.zeros_like(x)
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
    d0, d1 = _device(hprt, fn, x, y)
    o0, o1 = _oracle(orc, fn, x, y, libm=False)
    assert np.array_equal(d0.view(np.uint64), o0.view(np.uint64)) and np.array_equal(d1.view(np.uint64), o1.view(np.uint64)), \
        (name, int((d0.view(np.uint64) != o0.view(np.uint64)).sum()), int((d1.view(np.uint64) != o1.view(np.uint64)).sum()))
    if " fma " in open("/proc/cpuinfo").read():        # x86-64 glibc then runs the FMA builds the restatements follow
        l0, l1 = _oracle(orc, fn, x, y, libm=True)
        assert np.array_equal(d0.view(np.uint64), l0.view(np.uint64)) and np.array_equal(d1.view(np.uint64), l1.view(np.uint64)), name
    assert np.isfinite(d0).all()

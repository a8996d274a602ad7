"""Floating-point identities the HIP kernel relies on to reproduce the
reference's mixed-precision expressions with cheaper instructions."""
import numpy as np


def test_f32_reciprocal_equals_double_rounded_reciprocal():
    """Helper.h:83-84 computes (float)(1.0 / (double)sqrtf(q)); the kernel uses
    the correctly rounded float division 1.0f / y.  Exhaustive over every float
    in [0.25, 4): q = |s|^2 stays within a few ulp of 1."""
    lo = np.float32(0.25).view(np.uint32)
    hi = np.float32(4.0).view(np.uint32)
    step = 1 << 22
    for start in range(int(lo), int(hi), step):
        y = np.arange(start, min(start + step, int(hi)), dtype=np.uint32).view(np.float32)
        a = (np.float64(1.0) / y.astype(np.float64)).astype(np.float32)
        b = np.float32(1.0) / y
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_tanf_equals_rounded_double_tan_on_the_angle_range():
    """Helper.h:409-410: s = tanf(1e-3f * a).  The kernel evaluates tan in f64
    and rounds; on the +-20 mrad range both give the same float.
    The C library's tanf / tan are called directly (numpy's float32 tan is its
    own SIMD routine, not tanf)."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.tanf.restype, libm.tanf.argtypes = ctypes.c_float, [ctypes.c_float]
    libm.tan.restype, libm.tan.argtypes = ctypes.c_double, [ctypes.c_double]
    a = np.arange(-20000, 20001, 4, dtype=np.float64) * 1e-3        # mrad, 4 urad pitch
    x = (np.float32(1e-3) * a.astype(np.float32)).astype(np.float32)
    t32 = np.array([libm.tanf(float(v)) for v in x], dtype=np.float32)
    t64 = np.array([libm.tan(float(v)) for v in x], dtype=np.float64).astype(np.float32)
    assert np.array_equal(t32.view(np.uint32), t64.view(np.uint32))

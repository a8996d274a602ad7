"""ctypes mirror of include/rt_hip.h (the C ABI of the HIP backend).

The same record types feed the product library (csrc/librt_hip.so) and, in
tests only, the CPU oracle -- so one marshalled problem is handed to both
sides.  Nothing here computes; it only lays numpy arrays out as the PODs the
header declares.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

RT_N_SUB = 3
RT_N_FAILED_MAX = 32

RT_OK, RT_ERR_ARG, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_NOMEM = range(5)

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)


class RtRay(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("a", C.c_float), ("b", C.c_float)]


#: numpy view of rt_ray[] (16-byte records, reference ray_struct layout)
RAY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("a", "<f4"), ("b", "<f4")])


class RtBeam(C.Structure):
    _fields_ = [
        ("nx", C.c_int32), ("ny", C.c_int32), ("na", C.c_int32), ("nb", C.c_int32),
        ("nv", C.c_int32),
        ("dx", C.c_double), ("dy", C.c_double), ("da", C.c_double), ("db", C.c_double),
        ("dz", C.c_double),
        ("x", c_double_p), ("y", c_double_p), ("a", c_double_p), ("b", c_double_p),
        ("dv", c_double_p),
    ]


class RtGain(C.Structure):
    _fields_ = [
        ("Nx", C.c_int32), ("Ny", C.c_int32), ("Nv", C.c_int32),
        ("x", c_double_p), ("y", c_double_p), ("n", c_double_p),
        ("g0", c_float_p), ("E0", c_float_p), ("gv", c_float_p),
    ]


class RtSeed(C.Structure):
    _fields_ = [
        ("dim", C.c_int32 * 5),
        ("x", c_double_p * 5),
        ("f", c_double_p * 5),
        ("f0", C.c_double),
    ]


class RtStats(C.Structure):
    _fields_ = [
        ("n_rays", C.c_uint64), ("cell_steps", C.c_uint64), ("n_escaped", C.c_uint64),
        ("n_skipped", C.c_uint64), ("kernel_ms", C.c_float), ("total_ms", C.c_float),
        ("march_ms", C.c_float), ("freq_ms", C.c_float),
    ]


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(c_double_p)


def _fp(a: np.ndarray | None):
    if a is None:
        return c_float_p()
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(c_float_p)


class Marshalled:
    """A Problem laid out as C records.  Holds references to every numpy array
    it points into, so it must outlive the native call."""

    def __init__(self, problem):
        from .problem import Problem  # noqa: F401  (type only)

        p = problem
        self.problem = p
        self._keep = []
        b = p.beam
        self.beam = RtBeam(b.nx, b.ny, b.na, b.nb, b.nv, b.dx, b.dy, b.da, b.db, b.dz,
                           _dp(b.x), _dp(b.y), _dp(b.a), _dp(b.b), _dp(b.dv))
        self._keep += [b.x, b.y, b.a, b.b, b.dv]
        self.N = len(p.gain)
        self.gain = (RtGain * self.N)()
        for i, g in enumerate(p.gain):
            self.gain[i] = RtGain(g.Nx, g.Ny, g.Nv, _dp(g.x), _dp(g.y), _dp(g.n),
                                  _fp(g.g0), _fp(g.E0), _fp(g.gv))
            self._keep += [g.x, g.y, g.n, g.g0, g.E0, g.gv]
        self.seed = None
        if p.seed is not None:
            s = RtSeed()
            for i in range(5):
                s.dim[i] = int(p.seed.x[i].shape[0])
                s.x[i] = _dp(p.seed.x[i])
                s.f[i] = _dp(p.seed.f[i])
                self._keep += [p.seed.x[i], p.seed.f[i]]
            s.f0 = p.seed.f0
            self.seed = s

    @property
    def seed_ref(self):
        return C.byref(self.seed) if self.seed is not None else None


def rays_ptr(rays: np.ndarray):
    assert rays.dtype == RAY_DTYPE and rays.flags.c_contiguous
    return rays.ctypes.data_as(C.POINTER(RtRay))


def declare_hip_api(lib: C.CDLL) -> None:
    """Attach argtypes/restype for every symbol include/rt_hip.h declares."""
    P = C.POINTER
    vp = C.c_void_p
    lib.rt_hip_device_count.argtypes = []
    lib.rt_hip_device_count.restype = C.c_int
    lib.rt_hip_last_error.argtypes = []
    lib.rt_hip_last_error.restype = C.c_char_p
    lib.rt_hip_selftest.argtypes = [C.c_int, P(C.c_ulonglong), P(C.c_ulonglong)]
    lib.rt_hip_selftest.restype = C.c_int
    lib.rt_hip_image_loop.argtypes = [
        C.c_int, C.c_int, P(RtBeam), P(RtGain), P(RtSeed), C.c_int, P(RtRay), C.c_size_t,
        C.c_double, c_double_p, c_double_p, P(C.c_uint), P(RtRay), C.c_int, P(C.c_int),
        P(RtStats)]
    lib.rt_hip_image_loop.restype = C.c_int
    lib.rt_hip_multi_image_loop.argtypes = list(lib.rt_hip_image_loop.argtypes)
    lib.rt_hip_multi_image_loop.restype = C.c_int
    lib.rt_hip_multi_last_mode.argtypes = []
    lib.rt_hip_multi_last_mode.restype = C.c_int
    lib.rt_hip_ray_list_grid_dims.argtypes = [P(RtRay), C.c_size_t, P(C.c_int * 4)]
    lib.rt_hip_ray_list_grid_dims.restype = C.c_int
    lib.rt_hip_host_libm_mode.argtypes = [C.c_int]
    lib.rt_hip_host_libm_mode.restype = C.c_int
    lib.rt_hip_pool_trim.argtypes = []
    lib.rt_hip_pool_trim.restype = None
    lib.rt_hip_plan_create.argtypes = [P(vp), C.c_int, C.c_int, P(RtBeam), P(RtGain), P(RtSeed),
                                       C.c_int, C.c_double]
    lib.rt_hip_plan_create.restype = C.c_int
    lib.rt_hip_plan_set_rays.argtypes = [vp, P(RtRay), C.c_size_t]
    lib.rt_hip_plan_set_rays.restype = C.c_int
    lib.rt_hip_plan_set_ray_grid.argtypes = [vp, c_double_p, C.c_int, c_double_p, C.c_int,
                                             c_double_p, C.c_int, c_double_p, C.c_int,
                                             C.c_int64, C.c_int64, C.c_int64]
    lib.rt_hip_plan_set_ray_grid.restype = C.c_int
    lib.rt_hip_plan_run.argtypes = [vp, vp, vp, vp]
    lib.rt_hip_plan_run.restype = C.c_int
    lib.rt_hip_plan_fetch.argtypes = [vp, c_double_p, c_double_p, P(C.c_uint), P(RtRay), C.c_int,
                                      P(C.c_int), P(RtStats)]
    lib.rt_hip_plan_fetch.restype = C.c_int
    lib.rt_hip_plan_kernel_ms.argtypes = [vp, P(C.c_float)]
    lib.rt_hip_plan_kernel_ms.restype = C.c_int
    lib.rt_hip_plan_kernel_times.argtypes = [vp, P(C.c_float), P(C.c_float)]
    lib.rt_hip_plan_kernel_times.restype = C.c_int
    lib.rt_hip_plan_last_fused.argtypes = [vp]
    lib.rt_hip_plan_last_fused.restype = C.c_int
    lib.rt_hip_plan_set_timing_ring.argtypes = [vp, C.c_int]
    lib.rt_hip_plan_set_timing_ring.restype = C.c_int
    lib.rt_hip_plan_ring_times.argtypes = [vp, c_float_p, c_float_p, C.c_int, P(C.c_int)]
    lib.rt_hip_plan_ring_times.restype = C.c_int
    lib.rt_hip_plan_image_ptr.argtypes = [vp]
    lib.rt_hip_plan_image_ptr.restype = vp
    lib.rt_hip_plan_iang_ptr.argtypes = [vp]
    lib.rt_hip_plan_iang_ptr.restype = vp
    lib.rt_hip_plan_enable_probe.argtypes = [vp, C.c_int]
    lib.rt_hip_plan_enable_probe.restype = C.c_int
    lib.rt_hip_plan_fetch_probe.argtypes = [vp, c_float_p, c_float_p, P(C.c_int32), P(RtRay),
                                            P(C.c_uint32), P(C.c_uint32)]
    lib.rt_hip_plan_fetch_probe.restype = C.c_int
    lib.rt_hip_plan_set_exact_emission.argtypes = [vp, C.c_int]
    lib.rt_hip_plan_set_exact_emission.restype = C.c_int
    lib.rt_hip_plan_set_step_factor.argtypes = [vp, C.c_double]
    lib.rt_hip_plan_set_step_factor.restype = C.c_int
    lib.rt_hip_plan_enable_path.argtypes = [vp, C.c_int]
    lib.rt_hip_plan_enable_path.restype = C.c_int
    lib.rt_hip_plan_fetch_path.argtypes = [vp, c_float_p, P(C.c_int32)]
    lib.rt_hip_plan_fetch_path.restype = C.c_int
    lib.rt_hip_plan_set_debug.argtypes = [vp, C.c_uint]
    lib.rt_hip_plan_set_debug.restype = C.c_int
    lib.rt_hip_plan_destroy.argtypes = [vp]
    lib.rt_hip_plan_destroy.restype = None


#: every symbol the header declares -- checked by tests/test_cabi_exports.py
HIP_API_SYMBOLS = [
    "rt_hip_device_count", "rt_hip_last_error", "rt_hip_selftest", "rt_hip_image_loop", "rt_hip_multi_image_loop",
    "rt_hip_multi_last_mode", "rt_hip_ray_list_grid_dims", "rt_hip_host_libm_mode", "rt_hip_pool_trim", "rt_hip_plan_create",
    "rt_hip_plan_set_rays", "rt_hip_plan_set_ray_grid", "rt_hip_plan_run", "rt_hip_plan_fetch",
    "rt_hip_plan_kernel_ms", "rt_hip_plan_kernel_times", "rt_hip_plan_last_fused", "rt_hip_plan_set_timing_ring", "rt_hip_plan_ring_times", "rt_hip_plan_image_ptr", "rt_hip_plan_iang_ptr", "rt_hip_plan_enable_probe",
    "rt_hip_plan_fetch_probe", "rt_hip_plan_set_exact_emission", "rt_hip_plan_set_step_factor", "rt_hip_plan_enable_path",
    "rt_hip_plan_fetch_path", "rt_hip_plan_set_debug", "rt_hip_plan_destroy",
]

"""Host-side binding of the HIP backend (csrc/librt_hip.so) and the Python
mirror of the reference's operator interface for this path.

  HipLibrary            loads the in-tree shared library; fails loudly when it
                        is missing -- there is NO CPU fallback in the product.
  Plan                  device-resident problem (rt_hip_plan_* of include/rt_hip.h)
  image_loop(...)       the back-end loop, same meaning as RayTraceImageCudaLoop
                        (src/RayTraceImageCuda.cu:145-221) behind the signature of
                        src/RayTraceImage.cpp:47-75
  create_image(p, method)
                        mirror of RayTrace::create_image (src/RayTraceImage.cpp:227-434):
                        checks, mode select, ray list, dispatch on the method
                        string ("hip", "hip-multigpu", "auto"), failure handling.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import time
from pathlib import Path

import numpy as np

from . import cabi
from .problem import Problem

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = CSRC / "librt_hip.so"


class RayTraceError(RuntimeError):
    """Raised where the reference calls RAY_ERROR (utilities/RayUtilityMacros.h:88-91)."""


def build_library(force: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    if force and LIB_PATH.exists():
        LIB_PATH.unlink()
    subprocess.run(["make", "-s", "-C", str(CSRC)], check=True)
    return LIB_PATH


class HipLibrary:
    _instance = None

    def __init__(self, path: Path = LIB_PATH):
        if not Path(path).exists():
            raise RayTraceError(
                f"HIP backend library {path} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        self.path = Path(path)
        self.lib = C.CDLL(str(path))
        cabi.declare_hip_api(self.lib)

    @classmethod
    def get(cls) -> "HipLibrary":
        if cls._instance is None:
            if not LIB_PATH.exists():
                # a fresh checkout (built files are not in history): compile in-tree with hipcc;
                # if that is impossible the constructor below raises -- there is no other path
                try:
                    build_library()
                except Exception as exc:  # noqa: BLE001
                    raise RayTraceError(f"HIP backend library {LIB_PATH} is missing and could not be built: {exc}") from exc
            cls._instance = HipLibrary()
        return cls._instance

    def device_count(self) -> int:
        return int(self.lib.rt_hip_device_count())

    def selftest(self, device: int = 0) -> tuple[int, int]:
        """(values checked, mismatches) of the on-device known-answer test of the march's exact shortcuts."""
        n, bad = C.c_ulonglong(0), C.c_ulonglong(0)
        self.check(self.lib.rt_hip_selftest(device, C.byref(n), C.byref(bad)), "rt_hip_selftest")
        return int(n.value), int(bad.value)

    def check(self, rc: int, what: str) -> None:
        if rc != cabi.RT_OK:
            msg = self.lib.rt_hip_last_error().decode(errors="replace")
            raise RayTraceError(f"{what} failed (status {rc}): {msg}")


class Plan:
    """A problem resident in HBM: tables uploaded once, runnable many times."""

    def __init__(self, problem: Problem, device: int = 0, lib: HipLibrary | None = None,
                 method: int | None = None):
        self.hl = lib or HipLibrary.get()
        self.problem = problem
        self.device = device
        self._m = cabi.Marshalled(problem)
        self._h = C.c_void_p()
        rc = self.hl.lib.rt_hip_plan_create(C.byref(self._h), device, self._m.N, C.byref(self._m.beam),
                                            self._m.gain, self._m.seed_ref,
                                            problem.method if method is None else method, problem.scale)
        self.hl.check(rc, "rt_hip_plan_create")
        self.n_rays = 0

    # -- rays ---------------------------------------------------------------
    def set_rays(self, rays: np.ndarray) -> "Plan":
        rays = np.ascontiguousarray(rays, dtype=cabi.RAY_DTYPE)
        self.hl.check(self.hl.lib.rt_hip_plan_set_rays(self._h, cabi.rays_ptr(rays), len(rays)),
                      "rt_hip_plan_set_rays")
        self.n_rays = len(rays)
        return self

    def set_ray_grid(self, first: int | None = None, stride: int | None = None,
                     count: int | None = None, grids=None) -> "Plan":
        """Rays generated on the device from the problem's ray grid
        (RayTraceImage.cpp:300-328); defaults follow N_start / N_parallel.
        `grids` = (x, y, a, b) overrides the problem's ray grid."""
        p = self.problem
        gx, gy, ga, gb = p.ray_grid if grids is None else [np.ascontiguousarray(g, np.float64) for g in grids]
        first = p.N_start if first is None else first
        stride = p.N_parallel if stride is None else stride
        if count is None:
            nt = len(gx) * len(gy) * len(ga) * len(gb)
            count = 0 if first >= nt else (nt - first + stride - 1) // stride
        self._grids = (gx, gy, ga, gb)
        self.hl.check(self.hl.lib.rt_hip_plan_set_ray_grid(
            self._h, cabi._dp(gx), len(gx), cabi._dp(gy), len(gy), cabi._dp(ga), len(ga),
            cabi._dp(gb), len(gb), first, stride, count), "rt_hip_plan_set_ray_grid")
        self.n_rays = count
        return self

    # -- run ----------------------------------------------------------------
    def enable_probe(self, on: bool = True) -> "Plan":
        self.hl.check(self.hl.lib.rt_hip_plan_enable_probe(self._h, int(on)), "rt_hip_plan_enable_probe")
        return self

    def run(self, stream: int = 0, image_ptr: int = 0, iang_ptr: int = 0) -> "Plan":
        """Asynchronous: zero outputs + trace kernel on `stream` (a hipStream_t value)."""
        self.hl.check(self.hl.lib.rt_hip_plan_run(self._h, C.c_void_p(stream), C.c_void_p(image_ptr),
                                                  C.c_void_p(iang_ptr)), "rt_hip_plan_run")
        return self

    def fetch(self, want_image: bool = True) -> dict:
        b = self.problem.beam
        image = np.empty(b.nx * b.ny * b.nv) if want_image else None
        iang = np.empty(b.na * b.nb) if want_image else None
        code = C.c_uint(0)
        nf = C.c_int(0)
        failed = np.zeros(cabi.RT_N_FAILED_MAX, dtype=cabi.RAY_DTYPE)
        st = cabi.RtStats()
        rc = self.hl.lib.rt_hip_plan_fetch(
            self._h, cabi._dp(image) if want_image else None, cabi._dp(iang) if want_image else None,
            C.byref(code), cabi.rays_ptr(failed), cabi.RT_N_FAILED_MAX, C.byref(nf), C.byref(st))
        self.hl.check(rc, "rt_hip_plan_fetch")
        return dict(image=image, I_ang=iang, failure_code=code.value, failed_rays=failed[:nf.value].copy(),
                    stats={k: getattr(st, k) for k, _ in cabi.RtStats._fields_})

    def set_exact_emission(self, on: bool = True) -> "Plan":
        """Emission mode: the CPU's per-frequency el/gl instead of the per-sub-segment ratio (include/rt_hip.h)."""
        self.hl.check(self.hl.lib.rt_hip_plan_set_exact_emission(self._h, int(on)), "rt_hip_plan_set_exact_emission")
        return self

    def set_step_factor(self, c: float) -> "Plan":
        self.hl.check(self.hl.lib.rt_hip_plan_set_step_factor(self._h, float(c)), "rt_hip_plan_set_step_factor")
        return self

    def set_debug(self, bits: int) -> "Plan":
        """Profiling aid: bit 0 skips the frequency kernel, bit 1 skips the march, bit 2 the I_ang flush (include/rt_hip.h)."""
        self.hl.check(self.hl.lib.rt_hip_plan_set_debug(self._h, int(bits)), "rt_hip_plan_set_debug")
        return self

    def enable_path(self, on: bool = True) -> "Plan":
        self.hl.check(self.hl.lib.rt_hip_plan_enable_path(self._h, int(on)), "rt_hip_plan_enable_path")
        return self

    def fetch_path(self) -> dict:
        """{x, y, I}: [n_rays][3(N-1)+1] float arrays, err: [n_rays] return codes."""
        n = self.n_rays
        N2 = (self.problem.N - 1) * cabi.RT_N_SUB + 1
        path = np.zeros((n, N2, 3), np.float32)
        err = np.zeros(n, np.int32)
        self.hl.check(self.hl.lib.rt_hip_plan_fetch_path(self._h, cabi._fp(path), err.ctypes.data_as(C.POINTER(C.c_int32))),
                      "rt_hip_plan_fetch_path")
        return dict(x=path[:, :, 0].copy(), y=path[:, :, 1].copy(), I=path[:, :, 2].copy(), err=err)

    def kernel_ms(self) -> float:
        """Device time of the last run's trace kernel (waits for it)."""
        ms = C.c_float(0)
        self.hl.check(self.hl.lib.rt_hip_plan_kernel_ms(self._h, C.byref(ms)), "rt_hip_plan_kernel_ms")
        return float(ms.value)

    def kernel_times(self) -> tuple:
        """(march_ms, freq_ms) of the last run (waits for it)."""
        a, f = C.c_float(0), C.c_float(0)
        self.hl.check(self.hl.lib.rt_hip_plan_kernel_times(self._h, C.byref(a), C.byref(f)),
                      "rt_hip_plan_kernel_times")
        return float(a.value), float(f.value)

    def last_fused(self) -> bool:
        """The last run took the whole path in one launch (rt_fused.hip): kernel_times() = (launch, 0)."""
        return bool(self.hl.lib.rt_hip_plan_last_fused(self._h))

    def set_timing_ring(self, n_runs: int) -> "Plan":
        """Keep the kernel-event triples of the last n_runs runs (include/rt_hip.h)."""
        self.hl.check(self.hl.lib.rt_hip_plan_set_timing_ring(self._h, int(n_runs)), "rt_hip_plan_set_timing_ring")
        self._ring = int(n_runs)
        return self

    def ring_times(self) -> list:
        """[(march_ms, freq_ms)] of the most recent runs, oldest first (waits for the last run)."""
        n = getattr(self, "_ring", 0)
        if n < 1:
            return []
        a = np.zeros(n, np.float32)
        f = np.zeros(n, np.float32)
        got = C.c_int(0)
        self.hl.check(self.hl.lib.rt_hip_plan_ring_times(self._h, cabi._fp(a), cabi._fp(f), n, C.byref(got)),
                      "rt_hip_plan_ring_times")
        return [(float(a[i]), float(f[i])) for i in range(got.value)]

    def fetch_probe(self) -> dict:
        n = self.n_rays
        S = (self.problem.N - 1) * cabi.RT_N_SUB
        out = dict(gvl=np.zeros((n, S), np.float32), evl=np.zeros((n, S), np.float32),
                   ivl=np.zeros((n, S), np.int32), ray2=np.zeros(n, cabi.RAY_DTYPE),
                   flags=np.zeros(n, np.uint32), steps=np.zeros(n, np.uint32))
        P = C.POINTER
        rc = self.hl.lib.rt_hip_plan_fetch_probe(
            self._h, cabi._fp(out["gvl"]), cabi._fp(out["evl"]), out["ivl"].ctypes.data_as(P(C.c_int32)),
            cabi.rays_ptr(out["ray2"]), out["flags"].ctypes.data_as(P(C.c_uint32)),
            out["steps"].ctypes.data_as(P(C.c_uint32)))
        self.hl.check(rc, "rt_hip_plan_fetch_probe")
        return out

    @property
    def image_ptr(self) -> int:
        return int(self.hl.lib.rt_hip_plan_image_ptr(self._h) or 0)

    @property
    def iang_ptr(self) -> int:
        return int(self.hl.lib.rt_hip_plan_iang_ptr(self._h) or 0)

    def close(self) -> None:
        if self._h:
            self.hl.lib.rt_hip_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def image_loop(problem: Problem, rays: np.ndarray | None = None, device: int = 0) -> dict:
    """RayTraceImageHipLoop through the host-pointer entry point
    (rt_hip_image_loop): upload, trace, download.  Returns image, I_ang,
    failure_code, failed_rays, stats."""
    hl = HipLibrary.get()
    m = cabi.Marshalled(problem)
    if rays is None:
        rays = problem.build_rays()
    rays = np.ascontiguousarray(rays, dtype=cabi.RAY_DTYPE)
    b = problem.beam
    image = np.zeros(b.nx * b.ny * b.nv)
    iang = np.zeros(b.na * b.nb)
    code = C.c_uint(0)
    nf = C.c_int(0)
    failed = np.zeros(cabi.RT_N_FAILED_MAX, dtype=cabi.RAY_DTYPE)
    st = cabi.RtStats()
    t0 = time.perf_counter()
    rc = hl.lib.rt_hip_image_loop(device, m.N, C.byref(m.beam), m.gain, m.seed_ref, problem.method,
                                  cabi.rays_ptr(rays), len(rays), problem.scale, cabi._dp(image),
                                  cabi._dp(iang), C.byref(code), cabi.rays_ptr(failed),
                                  cabi.RT_N_FAILED_MAX, C.byref(nf), C.byref(st))
    call_ms = (time.perf_counter() - t0) * 1e3  # the C call alone (what a C++ caller of the adapter waits for)
    hl.check(rc, "rt_hip_image_loop")
    return dict(image=image, I_ang=iang, failure_code=code.value, failed_rays=failed[:nf.value].copy(),
                stats={k: getattr(st, k) for k, _ in cabi.RtStats._fields_}, call_ms=call_ms)


def multi_image_loop(problem: Problem, rays: np.ndarray | None = None, n_devices: int = 0) -> dict:
    """RayTraceImageHipMultiGPULoop through rt_hip_multi_image_loop: all devices of the node, one RCCL
    collective per image (include/rt_hip.h).  n_devices <= 0: every device.  Returns what image_loop
    returns plus `mode` (1 = pixel-column tiles + gather, 2 = ray chunks + sum-reduce)."""
    hl = HipLibrary.get()
    m = cabi.Marshalled(problem)
    if rays is None:
        rays = problem.build_rays()
    rays = np.ascontiguousarray(rays, dtype=cabi.RAY_DTYPE)
    b = problem.beam
    image = np.zeros(b.nx * b.ny * b.nv)
    iang = np.zeros(b.na * b.nb)
    code = C.c_uint(0)
    nf = C.c_int(0)
    failed = np.zeros(cabi.RT_N_FAILED_MAX, dtype=cabi.RAY_DTYPE)
    st = cabi.RtStats()
    t0 = time.perf_counter()
    rc = hl.lib.rt_hip_multi_image_loop(n_devices, m.N, C.byref(m.beam), m.gain, m.seed_ref, problem.method,
                                        cabi.rays_ptr(rays), len(rays), problem.scale, cabi._dp(image),
                                        cabi._dp(iang), C.byref(code), cabi.rays_ptr(failed),
                                        cabi.RT_N_FAILED_MAX, C.byref(nf), C.byref(st))
    call_ms = (time.perf_counter() - t0) * 1e3
    hl.check(rc, "rt_hip_multi_image_loop")
    return dict(image=image, I_ang=iang, failure_code=code.value, failed_rays=failed[:nf.value].copy(),
                stats={k: getattr(st, k) for k, _ in cabi.RtStats._fields_}, mode=int(hl.lib.rt_hip_multi_last_mode()),
                call_ms=call_ms)


def ray_list_grid_dims(rays: np.ndarray):
    """(nx, ny, na, nb) if the list is a whole tensor grid in create_image's order, else None (host only)."""
    hl = HipLibrary.get()
    rays = np.ascontiguousarray(rays, dtype=cabi.RAY_DTYPE)
    dims = (C.c_int * 4)()
    ok = hl.lib.rt_hip_ray_list_grid_dims(cabi.rays_ptr(rays), len(rays), C.byref(dims))
    return tuple(dims) if ok else None


_FAILURE_TEXT = {1: "Invalid ray detected", 2: "Negitive intensity detected", 3: "NaNs detected in intensity"}


def create_image(problem: Problem, method: str = "auto", device: int = 0, device_rays: bool = True) -> dict:
    """Mirror of RayTrace::create_image (src/RayTraceImage.cpp:227-434) with the
    arms this backend adds: "hip" (one device) and "hip-multigpu" (all devices of
    the node behind rt_hip_multi_image_loop: pixel-column tiles + RCCL gather for
    ASE, ray chunks + RCCL sum-reduce otherwise).  "auto" resolves to "hip".  Any other method string
    is an error -- the CPU/OpenMP/CUDA arms belong to the reference.

    Returns dict(image [ny][nx][nv] flat, I_ang, stats)."""
    problem.validate()
    m = method.lower()
    if m == "auto":
        m = "hip"
    if m == "hip":
        with Plan(problem, device) as plan:
            if device_rays:
                plan.set_ray_grid()
            else:
                plan.set_rays(problem.build_rays())
            out = plan.run().fetch()
    elif m == "hip-multigpu":
        out = multi_image_loop(problem)
    else:
        raise RayTraceError("Unknown method: " + m)
    if out["failure_code"] != 0:
        msgs = [t for bit, t in _FAILURE_TEXT.items() if out["failure_code"] & (1 << bit)]
        raise RayTraceError("Some rays failed: " + "; ".join(msgs))
    return out


def calc_ray_path(problem: Problem, x, y, a, b, method: int | None = None, c: float = 0.5, device: int = 0):
    """Mirror of RayTrace::calc_ray_path (src/RayTraceImage.cpp:440-477): trace the rays of the
    tensor grid x * y * a * b through the problem's tables and return the path of each ray.

    Returns (xr, yr, Ir, n_errors): float32 arrays laid out as the reference lays them out,
    index = N2 * (i + j*Nx + k*Nx*Ny + m*Nx*Ny*Na) + step with N2 = 3 (N-1) + 1, i.e. shape
    [Nb][Na][Ny][Nx][N2] in C order."""
    grids = [np.ascontiguousarray(g, np.float64) for g in (x, y, a, b)]
    nx, ny, na, nb = (len(g) for g in grids)
    with Plan(problem, device, method=method) as plan:
        plan.set_step_factor(c).enable_path().set_ray_grid(0, 1, nx * ny * na * nb, grids=grids)
        out = plan.run().fetch_path()
    N2 = out["x"].shape[1]

    def lay(v):  # ray order is i, j, k, m with m fastest
        return np.ascontiguousarray(v.reshape(nx, ny, na, nb, N2).transpose(3, 2, 1, 0, 4))

    return lay(out["x"]), lay(out["y"]), lay(out["I"]), int((out["err"] != 0).sum())

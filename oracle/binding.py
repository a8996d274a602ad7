"""ctypes bindings of the CPU parity oracle and of the compiled reference.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.

  Oracle()    -> oracle/librt_oracle.so   (our C restatement, rt_oracle.c)
  Reference() -> oracle/_ref/librt_ref.so (the unmodified reference CPU path,
                 built by `make -C oracle ref` where /root/reference exists;
                 the built file travels to the GPU box, the sources do not)
"""
from __future__ import annotations

import ctypes as C
import importlib
import subprocess
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_pkg = importlib.import_module("raytrace-miniapp_amd")
cabi = _pkg.cabi

P = C.POINTER


class Counters(C.Structure):
    _fields_ = [("n_rays", C.c_uint64), ("cell_steps", C.c_uint64), ("cross_iters", C.c_uint64),
                ("inner_iters", C.c_uint64), ("n_escaped", C.c_uint64)]


def build(ref: bool = True) -> None:
    """Compile the oracle (and the reference library when its tree is present)."""
    subprocess.run(["make", "-s", "-C", str(HERE)], check=True)
    if ref and Path("/root/reference/src").is_dir():
        subprocess.run(["make", "-s", "-C", str(HERE), "ref"], check=True)


class Oracle:
    def __init__(self, path: Path | None = None):
        path = Path(path) if path else HERE / "librt_oracle.so"
        if not path.exists():
            build(ref=False)
        self.lib = C.CDLL(str(path))
        L = self.lib
        L.rt_oracle_image_loop.argtypes = [
            C.c_int, P(cabi.RtBeam), P(cabi.RtGain), P(cabi.RtSeed), C.c_int, P(cabi.RtRay),
            C.c_size_t, C.c_double, cabi.c_double_p, cabi.c_double_p, P(C.c_uint), P(cabi.RtRay),
            C.c_int, P(C.c_int), P(Counters), C.c_int]
        L.rt_oracle_image_loop.restype = C.c_int
        L.rt_oracle_probe.argtypes = [
            C.c_int, P(cabi.RtBeam), P(cabi.RtGain), P(cabi.RtSeed), C.c_int, P(cabi.RtRay),
            C.c_size_t, cabi.c_float_p, cabi.c_float_p, P(C.c_int32), P(cabi.RtRay),
            P(C.c_uint32), P(C.c_uint32), cabi.c_double_p, P(C.c_int32)]
        L.rt_oracle_probe.restype = C.c_int
        L.rt_oracle_calc_ray_path.argtypes = [
            C.c_int, P(cabi.RtBeam), P(cabi.RtGain), P(cabi.RtSeed), C.c_int, C.c_float, P(cabi.RtRay),
            C.c_size_t, cabi.c_float_p, P(C.c_int32)]
        L.rt_oracle_calc_ray_path.restype = C.c_int

    def image_loop(self, problem, rays=None, n_threads: int = 1):
        """Returns dict(image, I_ang, failure_code, failed_rays, counters, seconds)."""
        m = cabi.Marshalled(problem)
        if rays is None:
            rays = problem.build_rays()
        b = problem.beam
        image = np.zeros(b.nx * b.ny * b.nv)
        iang = np.zeros(b.na * b.nb)
        code = C.c_uint(0)
        nf = C.c_int(0)
        failed = np.zeros(cabi.RT_N_FAILED_MAX, dtype=cabi.RAY_DTYPE)
        cnt = Counters()
        t0 = time.perf_counter()
        rc = self.lib.rt_oracle_image_loop(
            m.N, C.byref(m.beam), m.gain, m.seed_ref, problem.method, cabi.rays_ptr(rays),
            len(rays), problem.scale, cabi._dp(image), cabi._dp(iang), C.byref(code),
            cabi.rays_ptr(failed), cabi.RT_N_FAILED_MAX, C.byref(nf), C.byref(cnt), n_threads)
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(f"rt_oracle_image_loop failed: {rc}")
        return dict(image=image, I_ang=iang, failure_code=code.value,
                    failed_rays=failed[:nf.value].copy(), seconds=dt,
                    counters={k: getattr(cnt, k) for k, _ in Counters._fields_})

    def probe(self, problem, rays, want_Iv: bool = True):
        m = cabi.Marshalled(problem)
        n = len(rays)
        S = (problem.N - 1) * cabi.RT_N_SUB
        K = problem.beam.nv
        out = dict(gvl=np.zeros((n, S), np.float32), evl=np.zeros((n, S), np.float32),
                   ivl=np.zeros((n, S), np.int32), ray2=np.zeros(n, cabi.RAY_DTYPE),
                   flags=np.zeros(n, np.uint32), steps=np.zeros(n, np.uint32),
                   err=np.zeros(n, np.int32))
        Iv = np.zeros((n, K)) if want_Iv else None
        rc = self.lib.rt_oracle_probe(
            m.N, C.byref(m.beam), m.gain, m.seed_ref, problem.method, cabi.rays_ptr(rays), n,
            cabi._fp(out["gvl"]), cabi._fp(out["evl"]), out["ivl"].ctypes.data_as(P(C.c_int32)),
            cabi.rays_ptr(out["ray2"]), out["flags"].ctypes.data_as(P(C.c_uint32)),
            out["steps"].ctypes.data_as(P(C.c_uint32)),
            cabi._dp(Iv) if want_Iv else None, out["err"].ctypes.data_as(P(C.c_int32)))
        if rc != 0:
            raise RuntimeError(f"rt_oracle_probe failed: {rc}")
        out["Iv"] = Iv
        return out


    def calc_ray_path(self, problem, rays, c: float = 0.5):
        """Per ray: x, y, I at the 3(N-1)+1 sub-segment boundaries -> dict(x, y, I [n][N2], err)."""
        m = cabi.Marshalled(problem)
        n = len(rays)
        N2 = (problem.N - 1) * cabi.RT_N_SUB + 1
        path = np.zeros((n, N2, 3), np.float32)
        err = np.zeros(n, np.int32)
        rc = self.lib.rt_oracle_calc_ray_path(m.N, C.byref(m.beam), m.gain, m.seed_ref, problem.method, c,
                                              cabi.rays_ptr(rays), n, cabi._fp(path),
                                              err.ctypes.data_as(P(C.c_int32)))
        if rc != 0:
            raise RuntimeError("rt_oracle_calc_ray_path failed")
        return dict(x=path[:, :, 0].copy(), y=path[:, :, 1].copy(), I=path[:, :, 2].copy(), err=err)


class Reference:
    """The compiled, unmodified reference CPU path (None-safe: `available()`)."""

    PATH = HERE / "_ref" / "librt_ref.so"

    @classmethod
    def available(cls) -> bool:
        return cls.PATH.exists()

    def __init__(self):
        if not self.PATH.exists():
            raise FileNotFoundError(f"{self.PATH} not built (make -C oracle ref)")
        self.lib = C.CDLL(str(self.PATH))
        L = self.lib
        L.ref_cpu_loop.argtypes = [
            C.c_int, P(cabi.RtBeam), P(cabi.RtGain), P(cabi.RtSeed), C.c_int, P(cabi.RtRay),
            C.c_size_t, C.c_double, cabi.c_double_p, cabi.c_double_p, P(C.c_uint), P(C.c_int)]
        L.ref_cpu_loop.restype = C.c_int
        L.ref_file_dims.argtypes = [C.c_char_p, P(C.c_int)]
        L.ref_file_dims.restype = C.c_int
        L.ref_create_image_file.argtypes = [C.c_char_p, C.c_char_p, cabi.c_double_p,
                                            cabi.c_double_p, cabi.c_double_p, cabi.c_double_p,
                                            cabi.c_double_p]
        L.ref_create_image_file.restype = C.c_int
        L.ref_calc_rays_file.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, cabi.c_double_p,
                                         cabi.c_double_p, P(C.c_int), cabi.c_double_p]
        L.ref_calc_rays_file.restype = C.c_int
        L.ref_calc_ray_path_file.argtypes = [C.c_char_p, P(C.c_int), P(C.c_int), C.c_double, cabi.c_float_p,
                                             cabi.c_float_p, cabi.c_float_p]
        L.ref_calc_ray_path_file.restype = C.c_int

    def scale_file(self, path, scale: float) -> dict:
        """The reference's scale_problem (src/CreateImageHelpers.cpp:104-150) applied to a .dat file: the grids it leaves
        in euv_beam and (if present) seed_beam."""
        L = self.lib
        L.ref_scale_file.argtypes = [C.c_char_p, C.c_double, C.c_int, P(C.c_int), cabi.c_double_p, cabi.c_double_p]
        L.ref_scale_file.restype = C.c_int
        out = {}
        for which, key in ((0, "beam"), (1, "seed_beam")):
            dims = (C.c_int * 4)()
            d = np.zeros(4)
            rc = L.ref_scale_file(str(path).encode(), scale, which, dims, cabi._dp(d), None)
            if rc < 0:
                raise FileNotFoundError(path)
            if rc == 1:
                continue
            n = list(dims)
            g = np.zeros(sum(n))
            L.ref_scale_file(str(path).encode(), scale, which, dims, cabi._dp(d), cabi._dp(g))
            o = np.cumsum([0] + n)
            out[key] = dict(n=np.array(n), d=d.copy(), x=g[o[0]:o[1]].copy(), y=g[o[1]:o[2]].copy(), a=g[o[2]:o[3]].copy(),
                            b=g[o[3]:o[4]].copy())
        return out

    def cpu_loop_sliced(self, problem, rays=None, max_nv: int = 96):
        """RayTraceImageCPULoop on a problem with more frequencies than the reference's K_MAX = 100
        (src/common/RayTraceImageHelper.h:30, src/RayTraceImage.cpp:231) allows: the frequencies are independent given
        a ray's march (SURVEY.md 8(c) iii), so the frequency axis is cut into slices of at most max_nv, each slice is a
        problem of its own for the reference, the images are concatenated along k and I_ang is the sum over the slices
        (returned per slice as well)."""
        import copy
        rt = importlib.import_module("raytrace-miniapp_amd")
        b = problem.beam
        K = b.nv
        if rays is None:
            rays = problem.build_rays()
        image = np.zeros((b.nx * b.ny, K))
        parts, code, secs = [], 0, 0.0
        for k0 in range(0, K, max_nv):
            k1 = min(K, k0 + max_nv)
            q = copy.copy(problem)
            qb = copy.copy(b)
            qb.dv = np.ascontiguousarray(b.dv[k0:k1])
            if getattr(b, "v", None) is not None and len(b.v) == K:
                qb.v = np.ascontiguousarray(b.v[k0:k1])
            q.beam = qb
            q.gain = [rt.Gain(g.x, g.y, g.n, g.g0, g.E0, np.ascontiguousarray(g.gv.reshape(-1, K)[:, k0:k1]).reshape(-1), k1 - k0)
                      for g in problem.gain]
            if problem.seed is not None:
                sd = problem.seed
                q.seed = rt.Seed(sd.x[:4] + [np.ascontiguousarray(sd.x[4][k0:k1])], sd.f[:4] + [np.ascontiguousarray(sd.f[4][k0:k1])], sd.f0)
            r = self.cpu_loop(q, rays)
            image[:, k0:k1] = r["image"].reshape(-1, k1 - k0)
            parts.append(r["I_ang"])
            code |= r["failure_code"]
            secs += r["seconds"]
        return dict(image=image.reshape(-1), I_ang=np.sum(parts, axis=0), I_ang_slices=np.array(parts), failure_code=code,
                    seconds=secs, slices=[(k0, min(K, k0 + max_nv)) for k0 in range(0, K, max_nv)])

    def cpu_loop(self, problem, rays=None):
        m = cabi.Marshalled(problem)
        if rays is None:
            rays = problem.build_rays()
        b = problem.beam
        image = np.zeros(b.nx * b.ny * b.nv)
        iang = np.zeros(b.na * b.nb)
        code = C.c_uint(0)
        nf = C.c_int(0)
        t0 = time.perf_counter()
        self.lib.ref_cpu_loop(m.N, C.byref(m.beam), m.gain, m.seed_ref, problem.method,
                              cabi.rays_ptr(rays), len(rays), problem.scale, cabi._dp(image),
                              cabi._dp(iang), C.byref(code), C.byref(nf))
        dt = time.perf_counter() - t0
        return dict(image=image, I_ang=iang, failure_code=code.value, n_failed=nf.value, seconds=dt)

    def file_dims(self, path) -> dict:
        d = (C.c_int * 12)()
        if self.lib.ref_file_dims(str(path).encode(), d) != 0:
            raise FileNotFoundError(path)
        keys = ["N", "N_start", "N_parallel", "nx", "ny", "na", "nb", "nv", "has_seed",
                "seed_nx", "gain_Nx", "gain_Ny"]
        return dict(zip(keys, list(d)))

    def create_image_file(self, path, method: str = "cpu"):
        d = self.file_dims(path)
        image = np.zeros(d["nx"] * d["ny"] * d["nv"])
        iang = np.zeros(d["na"] * d["nb"])
        gimg = np.zeros_like(image)
        gang = np.zeros_like(iang)
        sec = C.c_double(0)
        rc = self.lib.ref_create_image_file(str(path).encode(), method.encode(), cabi._dp(image),
                                            cabi._dp(iang), cabi._dp(gimg), cabi._dp(gang),
                                            C.byref(sec))
        if rc != 0:
            raise RuntimeError("ref_create_image_file failed")
        return dict(image=image, I_ang=iang, golden_image=gimg, golden_I_ang=gang,
                    seconds=sec.value, dims=d)

    def calc_ray_path_file(self, path, i0, n, c: float = 0.5):
        """RayTrace::calc_ray_path on the sub-grid [i0, i0+n) of the file's ray grid.
        Returns x, y, I as [nb][na][ny][nx][N2] arrays (the reference's layout) and the error count."""
        d = self.file_dims(path)
        N2 = (d["N"] - 1) * cabi.RT_N_SUB + 1
        tot = N2 * n[0] * n[1] * n[2] * n[3]
        xr, yr, ir = (np.zeros(tot, np.float32) for _ in range(3))
        a0 = (C.c_int * 4)(*i0)
        an = (C.c_int * 4)(*n)
        nerr = self.lib.ref_calc_ray_path_file(str(path).encode(), a0, an, c, cabi._fp(xr), cabi._fp(yr), cabi._fp(ir))
        if nerr < 0:
            raise RuntimeError("ref_calc_ray_path_file failed")
        shp = (n[3], n[2], n[1], n[0], N2)
        return dict(x=xr.reshape(shp), y=yr.reshape(shp), I=ir.reshape(shp), n_errors=nerr)

    def calc_rays_file(self, path, stride: int, n: int):
        d = self.file_dims(path)
        Iv = np.zeros((n, d["nv"]))
        ray2 = np.zeros((n, 4))
        rin = np.zeros((n, 4))
        err = (C.c_int * n)()
        rc = self.lib.ref_calc_rays_file(str(path).encode(), stride, n, cabi._dp(Iv),
                                         cabi._dp(ray2), err, cabi._dp(rin))
        if rc != 0:
            raise RuntimeError("ref_calc_rays_file failed")
        return dict(Iv=Iv, ray2=ray2, rays=rin, err=np.array(list(err), np.int32))

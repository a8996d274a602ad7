"""Diagnostic (librt_hip_wt.so: -DRT_WAVETIMES): when the waves of a march launch run out of rays and when they end."""
import ctypes as C, importlib, sys
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
lib = be.HipLibrary(be.CSRC / "librt_hip_wt.so")
full = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
for name, p in (("N=1", full), ("N=8 shard", mg.shard(full, 0, 8))):
    with be.Plan(p, lib=lib) as plan:
        plan.set_ray_grid().set_debug(1)
        s = (C.c_ulonglong * 8)(); e = (C.c_ulonglong * 8192)(); d = (C.c_ulonglong * 8192)()
        for _ in range(3):
            plan.run(); plan.fetch(want_image=False)
            lib.lib.rt_hip_debug_wavetimes(s, e, d)
        plan.run(); st = plan.fetch(want_image=False)["stats"]
        lib.lib.rt_hip_debug_wavetimes(s, e, d)
    s = list(s); n = int(s[6]); t0 = s[0]
    raw_e = np.array(e[:n], dtype=np.uint64); wav = (raw_e >> np.uint64(56)).astype(np.int64); raw_e = raw_e & np.uint64((1 << 56) - 1)
    end = (raw_e.astype(np.float64) - t0) / 100.0; dry = (np.array(d[:n], dtype=np.float64) - t0) / 100.0   # microseconds
    print(f"{name}: march {st['march_ms']:.3f} ms, waves {n}; starts within {(s[1]-s[0])/100:.1f} us")
    print(f"   counter dry : first {dry.min():8.1f} us  median {np.median(dry):8.1f}  last {dry.max():8.1f}")
    print(f"   wave ends   : first {end.min():8.1f} us  median {np.median(end):8.1f}  last {end.max():8.1f}")
    q = np.percentile(end, [10, 25, 50, 75, 90, 99])
    print("   end percentiles 10/25/50/75/90/99 (us):", " ".join(f"{x:.0f}" for x in q))
    print(f"   mean idle at the end: {(end.max() - end).mean():.1f} us per wave = {(end.max() - end).mean() / end.max() * 100:.1f} % of the launch")
    for sl in range(4):
        m = (wav // 4) == sl
        if m.any():
            print(f"   waves in SIMD slot {sl}: counter dry median {np.median(dry[m]):8.1f} us, end median {np.median(end[m]):8.1f}, last {end[m].max():8.1f}")

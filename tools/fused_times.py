"""Diagnostic (librt_hip_wt.so: -DRT_WAVETIMES): time line of the one-launch run -- when the ray counters run dry,
when the waves leave the march, when work-groups finish their frequency phase, and how the tiles are spread."""
import ctypes as C, importlib, sys
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
lib = be.HipLibrary(be.CSRC / "librt_hip_wt.so")
full = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
small = rt.datfile.load('tests/golden/ASE_small.dat.xz')
for name, p in (("N=1", full), ("N=8 shard", mg.shard(full, 0, 8)), ("ASE_small.dat", small)):
    with be.Plan(p, lib=lib) as plan:
        plan.set_ray_grid()
        s = (C.c_ulonglong * 8)(); e = (C.c_ulonglong * 8192)(); d = (C.c_ulonglong * 8192)()
        ft = (C.c_ulonglong * (6 * 8192))(); nw = C.c_uint(0)
        for _ in range(3):
            plan.run(); plan.fetch(want_image=False)
            lib.lib.rt_hip_debug_wavetimes(s, e, d); lib.lib.rt_hip_debug_freqtimes(ft, C.byref(nw))
        plan.run(); st = plan.fetch(want_image=False)["stats"]; assert plan.last_fused()
        lib.lib.rt_hip_debug_wavetimes(s, e, d); lib.lib.rt_hip_debug_freqtimes(ft, C.byref(nw))
    t0 = list(s)[0]; n = nw.value
    F = np.array(ft, dtype=np.uint64).reshape(6, 8192)[:, :n]
    us = lambda a: (a.astype(np.float64) - t0) / 100.0
    left, buf, first, end = us(F[0]), us(F[1]), us(F[2]), us(F[3])
    wg = (F[4] & np.uint64(0xffff)).astype(np.int64); tiles = F[5].astype(np.int64)
    dry = (np.array(d[:int(list(s)[6])], dtype=np.float64) - t0) / 100.0
    print(f"{name}: launch {st['march_ms']:.3f} ms, waves {n}")
    k0 = list(s)[2]
    if k0 and k0 != 0xffffffffffffffff:
        print(f"   first work-group enters the kernel {(t0 - k0) / 100.0:.1f} us before the first wave starts marching; "
              f"last wave starts marching at {(list(s)[1] - k0) / 100.0:.1f} us")
    pc = lambda a: " ".join(f"{x:7.0f}" for x in np.percentile(a, [0, 10, 50, 90, 100]))
    print(f"   counters dry (us)      min/10/50/90/max: {pc(dry)}")
    print(f"   wave leaves the march  min/10/50/90/max: {pc(left)}")
    print(f"   waiting for a buffer   min/10/50/90/max: {pc(buf - left)}   waves that waited > 5 us: {(buf - left > 5).sum()}")
    print(f"   wave ends              min/10/50/90/max: {pc(end)}")
    wgs = np.unique(wg)
    wg_end = np.array([end[wg == g].max() for g in wgs]); wg_tiles = np.array([tiles[wg == g].sum() for g in wgs])
    wg_left = np.array([left[wg == g].max() for g in wgs])
    print(f"   work-group ends        min/10/50/90/max: {pc(wg_end)}")
    print(f"   last wave of a work-group leaves the march: {pc(wg_left)}")
    print(f"   tiles per work-group   min/10/50/90/max: {pc(wg_tiles)}   per wave: {pc(tiles)}")
    print(f"   idle at the end: {(end.max() - end).mean():.1f} us per wave = {(end.max() - end).mean() / end.max() * 100:.1f} % of the launch")
    k = np.argsort(wg_end)[-3:]
    for i in k:
        print(f"   late work-group {wgs[i]}: ends {wg_end[i]:.0f} us, tiles {wg_tiles[i]}, last wave left the march at {wg_left[i]:.0f} us")

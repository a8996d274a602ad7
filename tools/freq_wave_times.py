"""Diagnostic (librt_hip_wt.so: -DRT_WAVETIMES): where the time of a frequency launch goes -- per wave: start, tables
ready, first tile done, last tile done -- and the gap between the end of the march and the first frequency wave."""
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
        plan.set_ray_grid()
        s = (C.c_ulonglong * 8)(); e = (C.c_ulonglong * 8192)(); d = (C.c_ulonglong * 8192)()
        ft = (C.c_ulonglong * (6 * 8192))(); nw = C.c_uint(0)
        for _ in range(4):
            plan.run(); st = plan.fetch(want_image=False)["stats"]
            lib.lib.rt_hip_debug_wavetimes(s, e, d)
            lib.lib.rt_hip_debug_freqtimes(ft, C.byref(nw))
    n = nw.value
    raw = np.array(ft, dtype=np.uint64).reshape(6, 8192)[:, :n]
    t = raw[:4].astype(np.float64)
    m_end = np.array(e[:int(s[6])], dtype=np.float64).max()
    t0 = t[0].min()
    us = lambda x: x / 100.0
    print(f"{name}: march {st['march_ms']:.3f} ms, freq {st['freq_ms']:.3f} ms, frequency waves {n}")
    print(f"   last march wave end -> first frequency wave start: {us(t0 - m_end):.1f} us")
    for k, label in enumerate(("start", "tables ready", "first tile done", "last tile done")):
        x = us(t[k] - t0)
        print(f"   {label:16s}: min {x.min():7.1f}  median {np.median(x):7.1f}  max {x.max():7.1f} us")
    print(f"   first tile (ready -> done), median {np.median(us(t[2] - t[1])):.1f} us; "
          f"tiles after the first, per wave median {np.median(us(t[3] - t[2])):.1f} us")
    where = raw[4]; tiles = raw[5].astype(np.int64)
    blk = (where & np.uint64(0xffff)).astype(np.int64); xcc = ((where >> np.uint64(24)) & np.uint64(0xf)).astype(np.int64)
    hwid = (where >> np.uint64(32)).astype(np.int64); wav = ((where >> np.uint64(16)) & np.uint64(0xff)).astype(np.int64)
    cu = (hwid >> 8) & 0xf; se = (hwid >> 13) & 0x7; simd = (hwid >> 4) & 0x3
    dur = us(t[3] - t[1]); first = us(t[2] - t[1]); per_tile = dur / np.maximum(tiles, 1)
    print(f"   tiles per wave: min {tiles.min()} median {int(np.median(tiles))} max {tiles.max()}; per-tile time (whole wave): "
          f"min {per_tile.min():.1f} median {np.median(per_tile):.1f} max {per_tile.max():.1f} us")
    for label, key in (("XCC", xcc), ("block % 8", blk % 8), ("SE", se), ("CU", cu), ("SIMD", simd), ("block / 256", blk // 256), ("wave in group", wav), ("wave / 4", wav // 4)):
        ks = np.unique(key)
        print(f"   by {label}: " + "  ".join(f"{k}: n {np.sum(key == k)} tile {np.median(per_tile[key == k]):.1f} end {np.median(us(t[3] - t0)[key == k]):.0f}" for k in ks[:16]))

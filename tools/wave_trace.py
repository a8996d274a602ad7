"""Diagnostic (librt_hip_wt.so): per-wave trace of the march of the one-launch run -- time, live lanes and runs of
blocks [A] / [B] every 16 loop iterations.  Prints the slowest waves' traces and the mean iteration time by live-lane
count.   python tools/wave_trace.py [shard8|standin|small]  (env RT_HIP_EXPRESS_* apply)"""
import ctypes as C, importlib, sys
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
lib = be.HipLibrary(be.CSRC / "librt_hip_wt.so")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
full = rt.scale_problem(base, 16.0)
case = sys.argv[1] if len(sys.argv) > 1 else "shard8"
p = {"small": base, "standin": full}.get(case) or mg.shard(full, 0, int(case[5:]))
NS = 64
with be.Plan(p, lib=lib) as plan:
    plan.set_ray_grid()
    s = (C.c_ulonglong * 8)(); e = (C.c_ulonglong * 8192)(); d = (C.c_ulonglong * 8192)()
    tr = (C.c_ulonglong * (8192 * NS))(); bl = (C.c_uint * (8192 * NS * 6))()
    for _ in range(3):
        plan.run(); plan.fetch(want_image=False)
        lib.lib.rt_hip_debug_wavetimes(s, e, d); lib.lib.rt_hip_debug_wavetrace(tr, NS, bl)
    pub = (C.c_ulonglong * (8192 * 4))(); vm = (C.c_ulonglong * 8192)()
    lib.lib.rt_hip_debug_publish(pub, vm)
    plan.run(); st = plan.fetch(want_image=False)["stats"]
    lib.lib.rt_hip_debug_wavetimes(s, e, d); lib.lib.rt_hip_debug_wavetrace(tr, NS, bl)
    lib.lib.rt_hip_debug_publish(pub, vm)
PUB = np.array(pub, dtype=np.float64).reshape(8192, 4); VM = np.array(vm, dtype=np.float64)
m = PUB[:, 1] > 0
pc = lambda a: " ".join(f"{x:8.1f}" for x in np.percentile(a, [0, 10, 50, 90, 99, 100]))
print(f"publishing waves {m.sum()}: tiles pushed per wave (min/10/50/90/99/max) {pc(PUB[m, 1])}")
print(f"   us in tile_publish per wave      {pc(PUB[m, 0] / 100)}")
print(f"   us per publish (mean of a wave)  {pc(PUB[m, 0] / PUB[m, 1] / 100)}")
print(f"   longest publish of a wave (us)   {pc(PUB[m, 3] / 100)}")
print(f"   failed compare-exchanges / wave  {pc(PUB[m, 2])}")
print(f"   us waiting for record stores     {pc(VM[m] / 100)}")
t0 = list(s)[0] & 0xffffffffff
T = np.array(tr, dtype=np.uint64).reshape(8192, NS)
BL = np.array(bl, dtype=np.uint32).reshape(8192, NS, 6)
n = T[:, 0].astype(np.int64)
used = np.nonzero(n)[0]
print(f"{case}: launch {st['march_ms']:.3f} ms, traced waves {len(used)}, samples per wave min/med/max {n[used].min()} {int(np.median(n[used]))} {n[used].max()} (x16 iterations)")
time = lambda v: ((v & np.uint64(0xffffffffff)).astype(np.float64) - t0) / 100.0
live = lambda v: ((v >> np.uint64(40)) & np.uint64(0xff)).astype(np.int64)
aruns = lambda v: ((v >> np.uint64(48)) & np.uint64(0xff)).astype(np.int64)
bruns = lambda v: ((v >> np.uint64(56)) & np.uint64(0xff)).astype(np.int64)
# iteration time by live-lane count (of the sample that ends the 16 iterations)
rows = []
for w in used:
    k = min(n[w], NS - 1)
    if k < 2:
        continue
    v = T[w, 1:k + 1]
    t = time(v); dt = np.diff(t) / 16.0
    rows.append(np.stack([dt, live(v)[1:], aruns(v)[1:], bruns(v)[1:], t[1:]], axis=1))
R = np.concatenate(rows)
print("live lanes   samples   us/iteration (mean)   [A] runs/16   [B] runs/16")
for lo, hi in ((1, 1), (2, 3), (4, 7), (8, 15), (16, 31), (32, 47), (48, 59), (60, 64)):
    m = (R[:, 1] >= lo) & (R[:, 1] <= hi)
    if m.sum():
        print(f"  {lo:2d}-{hi:2d}    {m.sum():8d}   {R[m, 0].mean():8.3f}            {R[m, 2].mean():6.1f}       {R[m, 3].mean():6.1f}")
for name, lo, hi in (("busy phase (t < 200 us)", 0, 200), ("200-350 us", 200, 350), ("after 350 us", 350, 1e9)):
    m = (R[:, 4] >= lo) & (R[:, 4] < hi)
    if m.sum():
        print(f"  {name}: samples {m.sum()}, us/iteration {R[m, 0].mean():.3f}, live {R[m, 1].mean():.1f}")
last = sorted(used, key=lambda w: time(T[w, min(n[w], NS - 1)]))[-4:]
for w in last:
    k = min(n[w], NS - 1)
    v = T[w, 1:k + 1]
    print(f"wave {w}: iterations >= {16 * n[w]}")
    print("   t(us):  " + " ".join(f"{x:5.0f}" for x in time(v)))
    print("   live:   " + " ".join(f"{x:5d}" for x in live(v)))
    print("   [A]/16: " + " ".join(f"{x:5d}" for x in aruns(v)))
    print("   [B]/16: " + " ".join(f"{x:5d}" for x in bruns(v)))
    for b, nm in enumerate(("head+refill", "[A1]", "[A2]", "retire+publish", "[B]", "[C]")):
        print(f"   kcycles {nm:15s}: " + " ".join(f"{x / 1000:5.1f}" for x in BL[w, 1:k + 1, b]))
# long intervals: who, when
import collections
nw = int(__import__("os").environ.get("RT_HIP_MARCH_THREADS", "1024")) // 64
ev = []
for w in used:
    k = min(n[w], NS - 1)
    if k < 2:
        continue
    t = time(T[w, 1:k + 1]); dt = np.diff(t)
    for i in np.nonzero(dt > 80.0)[0]:
        ev.append((w, t[i], dt[i], live(T[w, 1:k + 1])[i]))
print(f"16-iteration intervals longer than 80 us: {len(ev)} in {len(set(e[0] for e in ev))} of {len(used)} waves")
if ev:
    E = np.array(ev)
    print("   by wave of the work-group:", dict(sorted(collections.Counter((E[:, 0].astype(int) % nw).tolist()).items())))
    print("   by SIMD (wave % 4):       ", dict(sorted(collections.Counter((E[:, 0].astype(int) % 4).tolist()).items())))
    print("   start time (us) percentiles 0/10/50/90/100:", np.percentile(E[:, 1], [0, 10, 50, 90, 100]).round(0))
    print("   length (us) percentiles:", np.percentile(E[:, 2], [0, 10, 50, 90, 100]).round(0), " live lanes at start:", np.percentile(E[:, 3], [0, 10, 50, 90, 100]))
    wgs = collections.Counter((E[:, 0].astype(int) // nw).tolist())
    print(f"   work-groups affected: {len(wgs)}; events per affected work-group max {max(wgs.values())}")

"""A/B of library builds on the config-5 shape (n x n pixels, 512 frequencies, one ray per pixel):
python tools/config5_ab.py [n]"""
import importlib, sys, glob
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
pm = importlib.import_module("raytrace-miniapp_amd.problem")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
p = pm.regrid_beam(pm.resample_frequency(base, 512), nx=n, ny=n, a_centre=-1.0, b_centre=-4.5)
libs = [be.CSRC / "librt_hip.so"] + sorted(glob.glob(str(be.CSRC / "librt_hip_abl_*.so")))
plans = [be.Plan(p, lib=be.HipLibrary(path)).set_ray_grid() for path in libs]
best = [(1e9, 1e9)] * len(libs)
for rnd in range(3):
    for i, plan in enumerate(plans):
        plan.run(); st = plan.fetch(want_image=False)["stats"]
        best[i] = (min(best[i][0], st["march_ms"]), min(best[i][1], st["freq_ms"]))
for path, b in zip(libs, best):
    print(f"config5 {n}^2 {str(path).split('/')[-1]:30s} march {b[0]:7.3f} ms  freq {b[1]:7.3f} ms", flush=True)

"""A/B of library builds on a seeded workload four times the size of seed_small (30.7 M rays)."""
import importlib, sys, glob
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
base = rt.datfile.load('tests/golden/seed_small.dat.xz')
p = rt.scale_problem(base, 4.0)
print("rays", p.n_rays_total)
libs = [be.CSRC / "librt_hip.so"] + sorted(glob.glob(str(be.CSRC / "librt_hip_abl_*.so")))
plans = [be.Plan(p, lib=be.HipLibrary(path)).set_ray_grid() for path in libs]
best = [(1e9, 1e9)] * len(libs)
for rnd in range(4):
    for i, plan in enumerate(plans):
        plan.run(); st = plan.fetch(want_image=False)["stats"]
        best[i] = (min(best[i][0], st["march_ms"]), min(best[i][1], st["freq_ms"]))
for path, b in zip(libs, best):
    print(f"seed x4 {str(path).split('/')[-1]:30s} march {b[0]:7.3f} ms  freq {b[1]:7.3f} ms", flush=True)

"""Profiling-only: time the kernels of ablated builds (make -C raytrace-miniapp_amd/csrc ablate).
The builds are timed round-robin (clock ramp-up and box drift hit all of them alike); best of all rounds."""
import importlib, sys, glob
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
which = sys.argv[1] if len(sys.argv) > 1 else "ase"
p = rt.scale_problem(base, 16.0) if which == "ase" else rt.datfile.load('tests/golden/seed_small.dat.xz')
libs = [be.CSRC / "librt_hip.so"] + sorted(glob.glob(str(be.CSRC / "librt_hip_abl_*.so")))
plans = []
for path in libs:
    plan = be.Plan(p, lib=be.HipLibrary(path))
    plan.set_ray_grid()
    plans.append(plan)
best = [(1e9, 1e9)] * len(libs)
for rnd in range(6):
    for i, plan in enumerate(plans):
        for _ in range(2):
            plan.run()
            st = plan.fetch(want_image=False)["stats"]
            best[i] = (min(best[i][0], st["march_ms"]), min(best[i][1], st["freq_ms"]))
for path, b in zip(libs, best):
    print(f"{str(path).split('/')[-1]:32s} march {b[0]:7.3f} ms  freq {b[1]:7.3f} ms")
for plan in plans:
    plan.close()

"""Experiment: sweep RT_HIP_MARCH_PARK (lane parking threshold of block [A] of the march)."""
import importlib, os, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
# CASE=seed: the seeded input (seed_small.dat) instead of the ASE stand-in
if os.environ.get("CASE") == "seed":
    p = rt.datfile.load('tests/golden/seed_small.dat.xz')
else:
    p = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
for park in [int(a) for a in sys.argv[1:]] or [1, 8, 16, 24, 32]:
    os.environ["RT_HIP_MARCH_PARK"] = str(park)
    with be.Plan(p) as plan:
        plan.set_ray_grid()
        best = 1e9
        for _ in range(6):
            plan.run(); st = plan.fetch(want_image=False)["stats"]; best = min(best, st["march_ms"])
        print(f"park {park:3d}  march {best:.3f} ms  steps {st['cell_steps']}", flush=True)

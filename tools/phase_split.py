"""Diagnostic: kernel times of the march and frequency kernels on the three stock workloads."""
import importlib, sys, os
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
cases = {"ASE_medium_standin": rt.scale_problem(base, 16.0), "ASE_small": base,
         "seed_small": rt.datfile.load('tests/golden/seed_small.dat.xz')}
only = sys.argv[1:] or list(cases)
for name in only:
    p = cases[name]
    with be.Plan(p) as plan:
        plan.set_ray_grid()
        rows = []
        for i in range(5):
            plan.run()
            st = plan.fetch(want_image=False)["stats"]
            rows.append((st["kernel_ms"], st["march_ms"], st["freq_ms"]))
    best = min(rows)
    print(f"{name:20s} kernel {best[0]:8.3f} ms  march {best[1]:8.3f}  freq {best[2]:8.3f}   "
          f"steps {st['cell_steps']} skipped {st['n_skipped']} escaped {st['n_escaped']}  "
          f"Gsteps/s {st['cell_steps'] / best[0] / 1e6:.2f}")

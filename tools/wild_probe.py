"""Diagnostic: one ray with a NaN / infinite start per process (argv[1] = case), through the list path."""
import importlib, sys
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
p = rt.datfile.load('tests/golden/ASE_small.dat.xz')
case = sys.argv[1]
good = p.build_rays(np.arange(0, p.n_rays_total, 1999, dtype=np.int64))
r = good[:1].copy()
r["x"], r["y"] = 0.5 * (p.gain[1].x[0] + p.gain[1].x[-1]), 0.5 * p.gain[1].y[-1]
if case == "a_nan": r["a"] = np.nan
elif case == "a_inf": r["a"] = np.inf
elif case == "b_ninf": r["b"] = -np.inf
elif case == "x_nan": r["x"] = np.nan
elif case == "a_big": r["a"] = 5000.0          # 5 rad: the f64 tangent stands in (tan_wide)
elif case == "outside_nan": r["x"], r["a"] = 10.0, np.nan
elif case == "outside_inf": r["x"], r["a"] = 10.0, np.inf
elif case == "a_huge": r["a"] = 1e30
elif case == "outside_huge": r["x"], r["a"] = 10.0, -1e30
elif case == "good": r = good[:1].copy()
rays = np.concatenate([good, r]) if "--mixed" in sys.argv else r
with be.Plan(p) as plan:
    plan.set_rays(rays)
    print(case, "rays set", flush=True)
    plan.run()
    print(case, "launched", flush=True)
    out = plan.fetch()
print(case, "failure_code", out["failure_code"], "failed", len(out["failed_rays"]), "steps", out["stats"]["cell_steps"], flush=True)

import copy, importlib, sys, time, os
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
hip = importlib.import_module("raytrace-miniapp_amd.backend")
pm = importlib.import_module("raytrace-miniapp_amd.problem")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
p = copy.copy(base)
p.gain = [base.gain[0]] + [rt.Gain(g.x, g.y, np.ones_like(g.n), g.g0, g.E0, g.gv, g.Nv) for g in base.gain[1:]]
p.beam = copy.copy(base.beam); p.beam.dz = float("inf")
rays = np.zeros(3, dtype=rt.cabi.RAY_DTYPE)
rays["x"] = 0.5 * (base.gain[1].x[0] + base.gain[1].x[-1])
rays["y"] = [0.3 * base.gain[1].y[-1], 0.5 * base.gain[1].y[-1], 0.7 * base.gain[1].y[-1]]
q = pm.regrid_beam(p, nx=12, ny=6)
q.beam.a = q.beam.da * (np.arange(7) - 3.0); q.beam.b = q.beam.db * (np.arange(5) - 2.0)
for lim in (4096, 1 << 16, 1 << 20):
    os.environ["RT_HIP_MARCH_SPIN_LIMIT"] = str(lim)
    with hip.Plan(p) as plan:
        t = time.perf_counter(); out = plan.set_rays(rays).run().fetch(); dt = time.perf_counter() - t
    print("LIST limit", lim, "code", out["failure_code"], "failed", len(out["failed_rays"]), "kernel ms", round(out["stats"]["kernel_ms"], 2), flush=True)
    with hip.Plan(q) as plan:
        out = plan.set_ray_grid().run().fetch()
        print("GRID limit", lim, "code", out["failure_code"], "rays", out["stats"]["n_rays"], "fused", plan.last_fused(), "kernel ms", round(out["stats"]["kernel_ms"], 2), "failed", len(out["failed_rays"]), flush=True)

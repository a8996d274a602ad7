"""Extended validation: every march record of the ASE_medium stand-in (6.38 M rays; `seed`: of seed_small,
7.8 M rays) against the oracle, bit for bit."""
import importlib, sys, time
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
from oracle.binding import Oracle
which = sys.argv[1] if len(sys.argv) > 1 else "ase"
p = (rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0) if which == "ase"
     else rt.datfile.load('tests/golden/seed_small.dat.xz'))
rays = p.build_rays()
with be.Plan(p) as plan:
    plan.set_ray_grid().enable_probe().run(); out = plan.fetch(); pr = plan.fetch_probe()
t0 = time.time()
ora = Oracle()
n = len(rays); bad = 0
for a in range(0, n, 800000):          # oracle probe in slices (memory)
    b = min(n, a + 800000)
    o = ora.probe(p, rays[a:b], want_Iv=False)
    for key in ("gvl", "evl"):
        bad += int((pr[key][a:b].view(np.uint32) != o[key].view(np.uint32)).any(axis=1).sum())
    bad += int((pr["ivl"][a:b] != o["ivl"]).any(axis=1).sum()) + int((pr["steps"][a:b] != o["steps"]).sum())
    bad += int(((pr["flags"][a:b] & 3) != (o["flags"] & 3)).sum())
    okm = o["err"] == 0
    for key in "xyab":
        bad += int((pr["ray2"][key][a:b][okm].view(np.uint32) != o["ray2"][key][okm].view(np.uint32)).sum())
print(f"rays {n}, mismatching records/fields {bad}, oracle time {time.time()-t0:.1f} s")

"""A/B of one tuning env var inside one process: python tools/sweep_env.py VAR v1,v2,... case"""
import importlib, sys, os
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
var, vals, case = sys.argv[1], sys.argv[2].split(','), sys.argv[3]
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
p = {"medium": rt.scale_problem(base, 16.0), "small": base, "seed": rt.datfile.load('tests/golden/seed_small.dat.xz')}[case]
with be.Plan(p) as plan:
    plan.set_ray_grid()
    res = {v: [] for v in vals}
    for rep in range(4):
        for v in vals:
            os.environ[var] = v
            plan.run(); st = plan.fetch(want_image=False)["stats"]
            res[v].append((st["march_ms"], st["freq_ms"]))
    for v in vals:
        print(case, var, v, "march min %.3f  freq min %.3f  freq all %s" % (min(a for a, _ in res[v]), min(b for _, b in res[v]), [round(b, 2) for _, b in res[v]]))

"""Diagnostic: kernel times of the stand-in under values of one environment variable.
python tools/sweep_env.py VAR v1 v2 ...   (each value in a fresh plan, round-robin, best of 5)"""
import importlib, os, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
var, vals = sys.argv[1], sys.argv[2:]
case = os.environ.get("RT_CASE", "ase")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
if case.startswith("shard"):   # rank-0 pixel-column shard of an N-rank run of the stand-in
    p = mg.shard(rt.scale_problem(base, 16.0), 0, int(case[5:]))
else:
    p = {"ase": lambda: rt.scale_problem(base, 16.0), "small": lambda: base,
         "seed": lambda: rt.datfile.load('tests/golden/seed_small.dat.xz')}[case]()
best = {v: (1e9, 1e9) for v in vals}
with be.Plan(p) as plan:
    plan.set_ray_grid()
    for rnd in range(5):
        for v in vals:
            os.environ[var] = v
            for _ in range(2):
                plan.run()
                st = plan.fetch(want_image=False)["stats"]
                best[v] = (min(best[v][0], st["march_ms"]), min(best[v][1], st["freq_ms"]))
for v in vals:
    print(f"{var}={v:8s} march {best[v][0]:7.3f} ms  freq {best[v][1]:7.3f} ms   steps {st['cell_steps']}")

import importlib, sys, os, time
sys.path.insert(0,'.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
med = rt.scale_problem(base, 16.0)
seed = rt.datfile.load('tests/golden/seed_small.dat.xz')
for name, p in (("ASE_medium_standin", med), ("ASE_small", base), ("seed_small", seed)):
    for dbg in ("0", "1"):
        os.environ["RT_HIP_DEBUG"] = dbg
        with be.Plan(p) as plan:
            plan.set_ray_grid()
            ts=[]
            for i in range(4):
                plan.run(); ts.append(plan.kernel_ms())
            st = plan.fetch(want_image=False)["stats"]
        print(name, "skipB" if dbg=="1" else "full ", "kernel ms", [round(t,3) for t in ts], "steps", st["cell_steps"], "skipped", st["n_skipped"], "escaped", st["n_escaped"])

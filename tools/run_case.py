"""Run one stock workload a few times (for rocprofv3): python tools/run_case.py <ASE_small|seed_small|ASE_medium_standin|seed_medium> [n] [library.so]"""
import importlib, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
name = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
seed = rt.datfile.load('tests/golden/seed_small.dat.xz')
p = {"ASE_small": base, "seed_small": seed, "ASE_medium_standin": rt.scale_problem(base, 16.0),
     "seed_medium": rt.scale_problem(seed, 16.0)}[name]
lib = be.HipLibrary(sys.argv[3]) if len(sys.argv) > 3 else None
with (be.Plan(p, lib=lib) if lib else be.Plan(p)) as plan:
    plan.set_ray_grid()
    for i in range(n):
        plan.run(); st = plan.fetch(want_image=False)["stats"]
    print(name, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})

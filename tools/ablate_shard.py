"""A/B of library builds on the 8-rank shard and on the whole stand-in (round-robin, best of 12).  ORDER=reverse puts
the product library last -- position matters: see DESIGN.md 4.3."""
import importlib, os, sys, glob
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
full = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
cases = [("N=8 shard", mg.shard(full, 0, 8)), ("N=1", full)]
if os.environ.get("SHARDS"):
    cases = [(f"N={w} shard", mg.shard(full, 0, int(w)) if int(w) > 1 else full) for w in os.environ["SHARDS"].split(",")]
for name, p in cases:
    libs = [be.CSRC / "librt_hip.so"] + sorted(glob.glob(str(be.CSRC / "librt_hip_abl_*.so")))
    if os.environ.get("ORDER") == "reverse":
        libs = libs[::-1]
    plans = []
    for path in libs:
        plan = be.Plan(p, lib=be.HipLibrary(path)); plan.set_ray_grid(); plans.append(plan)
    best = [(1e9, 1e9)] * len(libs)
    for rnd in range(6):
        for i, plan in enumerate(plans):
            for _ in range(2):
                plan.run(); st = plan.fetch(want_image=False)["stats"]
                best[i] = (min(best[i][0], st["march_ms"]), min(best[i][1], st["freq_ms"]))
    for path, b in zip(libs, best):
        print(f"{name:10s} {str(path).split('/')[-1]:36s} march {b[0]:7.3f} ms  freq {b[1]:7.3f} ms", flush=True)
    for plan in plans: plan.close()

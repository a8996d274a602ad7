"""What the march's record stores cost (they reach HBM as partial lines, 2.9x the record bytes): the march with
and without them (librt_hip_abl_norec.so: -DRT_ABL_NOSTORE -DRT_ABL_NOMETA) on the stand-in, on the rank-0 shard of
an 8-rank run and on BASELINE config 5 (march only: without records the frequency pass has nothing to read)."""
import importlib, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
pm = importlib.import_module("raytrace-miniapp_amd.problem")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
full = rt.scale_problem(base, 16.0)
cases = {"stand-in N=1": full, "stand-in N=8 rank 0": mg.shard(full, 0, 8),
         "config 5 (4096^2 x 512)": pm.regrid_beam(pm.resample_frequency(base, 512), nx=4096, ny=4096, a_centre=-1.0, b_centre=-4.5)}
libs = {"with stores": be.CSRC / "librt_hip.so", "no stores": be.CSRC / "librt_hip_abl_norec.so"}
for name, p in cases.items():
    plans = {}
    for lab, path in libs.items():
        plan = be.Plan(p, lib=be.HipLibrary(path))
        plan.set_ray_grid().set_debug(1)      # march only
        plans[lab] = plan
    best = {lab: 1e9 for lab in libs}
    for rnd in range(5):
        for lab, plan in plans.items():
            for _ in range(2):
                plan.run()
                best[lab] = min(best[lab], plan.fetch(want_image=False)["stats"]["march_ms"])
    for plan in plans.values():
        plan.close()
    a, b = best["with stores"], best["no stores"]
    print(f"{name:26s} march {a:8.3f} ms with record stores, {b:8.3f} ms without: {100 * (a - b) / a:5.1f} %", flush=True)

import importlib, os, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
pm = importlib.import_module("raytrace-miniapp_amd.problem")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
p = pm.regrid_beam(pm.resample_frequency(base, 512), nx=4096, ny=4096, a_centre=-1.0, b_centre=-4.5)
with be.Plan(p) as plan:
    plan.set_ray_grid()
    for w in ("12", "16", "8", "12", "16"):
        os.environ["RT_HIP_FREQ_WG_WAVES"] = w
        t = []
        for _ in range(3):
            plan.run(); st = plan.fetch(want_image=False)["stats"]; t.append(st["freq_ms"])
        print("waves per work-group", w, "freq", round(min(t), 3), "ms", flush=True)

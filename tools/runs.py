import importlib, sys, glob, time
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
p = rt.scale_problem(base, 16.0)
for path in [be.CSRC / "librt_hip.so", be.CSRC / "librt_hip_abl_div32.so", be.CSRC / "librt_hip.so"]:
    lib = be.HipLibrary(path)
    with be.Plan(p, lib=lib) as plan:
        plan.set_ray_grid()
        rows = []
        for i in range(8):
            plan.run(); st = plan.fetch(want_image=False)["stats"]
            rows.append((round(st["march_ms"],2), round(st["freq_ms"],2)))
            if i == 3: time.sleep(1.0)
    print(str(path).split('/')[-1], rows)

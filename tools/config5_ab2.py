"""A/B of two library builds on the config-5 shape (default 4096^2 x 512, one ray per pixel) and on the stand-in:
python tools/config5_ab2.py [n] libA.so libB.so"""
import importlib, sys
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
pm = importlib.import_module("raytrace-miniapp_amd.problem")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
args = sys.argv[1:]
n = int(args.pop(0)) if args and args[0].isdigit() else 4096
libs = args
p = pm.regrid_beam(pm.resample_frequency(base, 512), nx=n, ny=n, a_centre=-1.0, b_centre=-4.5)
imgs = []
for path in libs:
    with be.Plan(p, lib=be.HipLibrary(path)) as plan:
        plan.set_ray_grid()
        t = []
        for _ in range(4):
            plan.run(); st = plan.fetch(want_image=False)["stats"]; t.append((st["march_ms"], st["freq_ms"]))
        out = plan.fetch() if n <= 1024 else None
    print(f"config5 {n}^2 {path.split('/')[-1]:24s} march {min(a for a, _ in t):7.3f} ms  freq {min(b for _, b in t):7.3f} ms", flush=True)
    imgs.append(out)
if imgs[0] is not None and len(imgs) > 1:
    print("max |d image| between the builds:", float(np.abs(imgs[0]["image"] - imgs[1]["image"]).max()))

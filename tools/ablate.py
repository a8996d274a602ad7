"""Profiling-only: time the kernels of ablated builds (make -C raytrace-miniapp_amd/csrc ablate)."""
import importlib, sys, glob
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
p = rt.scale_problem(base, 16.0)
libs = [be.CSRC / "librt_hip.so"] + sorted(glob.glob(str(be.CSRC / "librt_hip_abl_*.so")))
for path in libs:
    lib = be.HipLibrary(path)
    with be.Plan(p, lib=lib) as plan:
        plan.set_ray_grid()
        rows = []
        for i in range(4):
            plan.run(); st = plan.fetch(want_image=False)["stats"]
            rows.append((st["march_ms"], st["freq_ms"]))
    b = min(rows)
    print(f"{str(path).split('/')[-1]:32s} march {b[0]:7.3f} ms  freq {b[1]:7.3f} ms  steps {st['cell_steps']}")

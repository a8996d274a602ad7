"""BASELINE config 5 shape (synthetic nx x ny pixels, nv = 512, na = nb = 1): timing + properties."""
import importlib, sys, time
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
pm = importlib.import_module("raytrace-miniapp_amd.problem")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
p = pm.regrid_beam(pm.resample_frequency(base, 512), nx=n, ny=n, a_centre=-1.0, b_centre=-4.5)
print("rays", p.n_rays_total, "image GB", p.n_rays_total * 512 * 8 / 1e9)
with be.Plan(p) as plan:
    plan.set_ray_grid()
    for i in range(3):
        t0 = time.perf_counter(); plan.run(); st = plan.fetch(want_image=False)["stats"]; dt = time.perf_counter() - t0
        print(f"run {i}: wall {dt*1e3:.1f} ms kernels {st['kernel_ms']:.2f} (march {st['march_ms']:.2f} freq {st['freq_ms']:.2f}) steps {st['cell_steps']} skipped {st['n_skipped']}")

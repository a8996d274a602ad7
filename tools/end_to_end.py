"""ms/image of the host-pointer entry point (rt_hip_image_loop: pack + upload + kernels +
download), as the reference harness times create_image -- the PCIe-inclusive number."""
import importlib, sys, time
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
cases = {"ASE_small": base, "ASE_medium_standin": rt.scale_problem(base, 16.0),
         "seed_small": rt.datfile.load('tests/golden/seed_small.dat.xz')}
for name, p in cases.items():
    rays = p.build_rays()
    be.image_loop(p, rays)
    t, c = [], []
    for i in range(3):
        t0 = time.perf_counter(); out = be.image_loop(p, rays); t.append((time.perf_counter() - t0) * 1e3); c.append(out["call_ms"])
    with be.Plan(p) as plan:
        plan.set_ray_grid()
        g = []
        for i in range(3):
            t0 = time.perf_counter(); plan.run(); o2 = plan.fetch(); g.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:20s} rays {len(rays):9d}  image_loop (host ray list handed over) C call {min(c):6.2f} ms, from Python {min(t):6.2f} ms   "
          f"plan.run+fetch (device ray grid, tables resident) {min(g):7.2f} ms   kernels {out['stats']['kernel_ms']:.2f} ms")

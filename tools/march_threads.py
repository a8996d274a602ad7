"""March work-group size (512 ... 1024 threads) and rays per counter fetch on the whole stand-in."""
import importlib, os, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
p = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
for thr in (512, 640, 768, 896, 1024):
    for chunk in (128, 192, 384):
        os.environ["RT_HIP_MARCH_THREADS"] = str(thr); os.environ["RT_HIP_MARCH_CHUNK"] = str(chunk)
        with be.Plan(p) as plan:
            plan.set_ray_grid().set_timing_ring(10)
            for _ in range(14): plan.run()
            t = plan.ring_times()
        print(f"threads {thr:4d} chunk {chunk:4d}: march {min(a for a,_ in t):.3f} ms", flush=True)

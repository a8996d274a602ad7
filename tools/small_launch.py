"""Fixed cost of a march / frequency launch: ASE_small ray subsets of growing size (device ray grid)."""
import importlib, os, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
p = rt.datfile.load('tests/golden/ASE_small.dat.xz')
counts = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024, 2048, 4096, 16384, 65536, 131072, 399000]
with be.Plan(p) as plan:
    for count in counts:
        plan.set_ray_grid(0, 1, count).set_timing_ring(8)
        for _ in range(10):
            plan.run()
        t = plan.ring_times()
        print(f"rays {count:7d}: march {min(a for a, _ in t):.3f} ms  freq {min(b for _, b in t):.3f} ms", flush=True)

"""What one rank of an N-rank strong-scaling run costs: the stand-in's pixel-column shard of rank 0 for
N = 1, 2, 4, 8 traced alone on this GPU (kernel times, no collective).  Predicts the kernel-side scaling."""
import importlib, os, sys, time
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
full = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
base = None
for world in (1, 2, 4, 8):
    worst = (0, 0, 0)
    for rank in sorted({0, world - 1, world // 2}):
        p = mg.shard(full, rank, world)
        with be.Plan(p) as plan:
            plan.set_ray_grid().set_timing_ring(12)
            for _ in range(15):
                plan.run()
            t = plan.ring_times()
            wall0 = time.perf_counter()
            for _ in range(20):
                plan.run()
            plan.ring_times()
            wall = (time.perf_counter() - wall0) / 20 * 1e3
        m = min(a for a, _ in t); f = min(b for _, b in t)
        if m + f > worst[0] + worst[1]:
            worst = (m, f, wall)
    if base is None:
        base = worst[2]
    print(f"N {world}: rays/rank {p.n_rays_total:8d}  march {worst[0]:.3f}  freq {worst[1]:.3f}  sum {worst[0]+worst[1]:.3f} ms  "
          f"step wall {worst[2]:.3f} ms  -> kernel-side speed-up {base / worst[2]:.2f}x", flush=True)

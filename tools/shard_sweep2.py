"""Sweep rays per counter fetch (chunk) and threads on the strong-scaling shards and the full stand-in."""
import importlib, os, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
full = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
for world, thr_list in ((8, (768, 1024)), (4, (896, 1024)), (1, (1024,))):
    p = mg.shard(full, 0, world)
    for thr in thr_list:
        for chunk in (16, 32, 48, 64, 96, 128, 192, 256):
            os.environ["RT_HIP_MARCH_THREADS"] = str(thr)
            os.environ["RT_HIP_MARCH_CHUNK"] = str(chunk)
            with be.Plan(p) as plan:
                plan.set_ray_grid().set_timing_ring(10)
                for _ in range(12):
                    plan.run()
                t = plan.ring_times()
            print(f"N {world} threads {thr:4d} chunk {chunk:4d}: march {min(a for a, _ in t):.3f} ms", flush=True)

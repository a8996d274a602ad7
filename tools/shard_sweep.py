"""Sweep march launch parameters on the strong-scaling shards (rank 0 of N = 4, 8)."""
import importlib, os, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
full = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
for world in (8, 4, 2):
    p = mg.shard(full, 0, world)
    for thr in (512, 640, 768, 896, 1024):
        for park in (1, 12):
            for chunk in (0, 64):
                os.environ["RT_HIP_MARCH_THREADS"] = str(thr)
                os.environ["RT_HIP_MARCH_PARK"] = str(park)
                if chunk:
                    os.environ["RT_HIP_MARCH_CHUNK"] = str(chunk)
                else:
                    os.environ.pop("RT_HIP_MARCH_CHUNK", None)
                with be.Plan(p) as plan:
                    plan.set_ray_grid().set_timing_ring(10)
                    for _ in range(12):
                        plan.run()
                    t = plan.ring_times()
                print(f"N {world} threads {thr:4d} park {park:2d} chunk {chunk or 'auto':>4}: march {min(a for a, _ in t):.3f} ms", flush=True)

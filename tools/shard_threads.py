"""March work-group size and rays per counter fetch on pixel-column shards of the stand-in (8, 6 and 16 ranks): the
data behind the thresholds in plan_launch_run (rt_launch.hip)."""
import importlib, os, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
full = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
import os as _os
for world in [int(w) for w in _os.environ.get('SHARDS', '8,6,16').split(',')]:
    p = mg.shard(full, 0, world) if world > 1 else full
    for thr in [int(t) for t in _os.environ.get('THREADS', '512,640,768,1024').split(',')]:
        for chunk in [int(c) for c in _os.environ.get('CHUNKS', '32,48,64,80,96').split(',')]:
            os.environ["RT_HIP_MARCH_THREADS"] = str(thr); os.environ["RT_HIP_MARCH_CHUNK"] = str(chunk)
            with be.Plan(p) as plan:
                plan.set_ray_grid().set_timing_ring(10)
                for _ in range(14): plan.run()
                t = plan.ring_times()
            print(f"N={world} threads {thr:4d} chunk {chunk:4d}: march {min(a for a,_ in t):.3f} ms", flush=True)

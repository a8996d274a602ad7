import importlib, os, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
full = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
p = mg.shard(full, 0, 8)
pads = [0, 256, 512, 1024, 2048, 4096, 8192, 12288, 16384, 32768, 65536, 131072, 262144]
best = {k: 1e9 for k in pads}
with be.Plan(p) as plan:
    plan.set_ray_grid()
    for rnd in range(5):
        for k in pads:
            os.environ["RT_HIP_IANG_PAD"] = str(k)
            for _ in range(2):
                plan.run(); st = plan.fetch(want_image=False)["stats"]
                best[k] = min(best[k], st["freq_ms"])
for k in pads: print(f"pad {k:7d}: freq {best[k]:.3f} ms")

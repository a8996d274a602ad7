"""What the end-of-launch I_ang flush of the frequency kernel costs on the 8-rank shard, per plan: the same plan with
and without the flush (debug bit 2), for every library build present (copies of one build show the placement effect:
DESIGN.md 6)."""
import importlib, os, sys, glob
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
full = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
p = mg.shard(full, 0, 8)
libs = [be.CSRC / "librt_hip.so"] + sorted(glob.glob(str(be.CSRC / "librt_hip_abl_*.so")))
plans = [be.Plan(p, lib=be.HipLibrary(path)).set_ray_grid() for path in libs]
best = {}
for rnd in range(5):
    for i, plan in enumerate(plans):
        for dbg in (0, 4):
            plan.set_debug(dbg)
            for _ in range(2):
                plan.run(); st = plan.fetch(want_image=False)["stats"]
                best[(i, dbg)] = min(best.get((i, dbg), 1e9), st["freq_ms"])
for i in range(len(plans)):
    print(f"plan {i}: with flush {best[(i,0)]:.3f} ms, without {best[(i,4)]:.3f} ms")

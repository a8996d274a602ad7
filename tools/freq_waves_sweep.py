"""Frequency-kernel occupancy on small launches: work-group size x work-groups per CU (RT_HIP_FREQ_WG_WAVES,
RT_HIP_FREQ_WGS) on pixel-column shards of the stand-in."""
import importlib, os, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
full = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
combos = [("16", "1"), ("12", "1"), ("8", "1"), ("4", "1"), ("4", "2"), ("4", "3")]
for world in (8, 4, 2, 1):
    p = mg.shard(full, 0, world) if world > 1 else full
    best = {c: 1e9 for c in combos}
    with be.Plan(p) as plan:
        plan.set_ray_grid()
        for rnd in range(4):
            for c in combos:
                os.environ["RT_HIP_FREQ_WG_WAVES"], os.environ["RT_HIP_FREQ_WGS"] = c
                for _ in range(2):
                    plan.run(); st = plan.fetch(want_image=False)["stats"]
                    best[c] = min(best[c], st["freq_ms"])
    print(f"N={world}: " + "  ".join(f"{int(c[0]) * int(c[1]):2d} waves/CU ({c[0]}x{c[1]}) {best[c]:.3f}" for c in combos), flush=True)

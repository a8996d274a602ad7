"""Diagnostic: frequency-kernel time of the emission mode, source-function form vs the CPU's formula."""
import importlib, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
p = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
with be.Plan(p) as plan:
    plan.set_ray_grid()
    for mode in (False, True, False, True):
        plan.set_exact_emission(mode)
        best = 1e9
        for _ in range(4):
            plan.run(); st = plan.fetch(want_image=False)["stats"]; best = min(best, st["freq_ms"])
        print("exact" if mode else "default", f"freq {best:.3f} ms  march {st['march_ms']:.3f}")

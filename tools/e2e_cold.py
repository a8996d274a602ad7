"""Why are the kernels of an rt_hip_image_loop call slower than the same kernels in a resident plan?"""
import importlib, sys, time
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
p = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
rays = p.build_rays()
for i in range(6):
    out = be.image_loop(p, rays); print("image_loop kernel_ms", round(out['stats']['kernel_ms'], 3), flush=True)
# resident plan, back to back
with be.Plan(p) as plan:
    plan.set_ray_grid()
    for i in range(6):
        plan.run(); st = plan.fetch(want_image=False)["stats"]; print("plan.run kernel_ms", round(st['kernel_ms'], 3), "fused", plan.last_fused())
    # resident plan with a pause before each run (as long as a call's host work)
    for pause in (0.0005, 0.002, 0.01):
        for i in range(3):
            time.sleep(pause); plan.run(); st = plan.fetch(want_image=False)["stats"]; print("pause", pause, "kernel_ms", round(st['kernel_ms'], 3))
# a new plan per run (fresh tables), no list
for i in range(4):
    with be.Plan(p) as plan:
        plan.set_ray_grid(); plan.run(); st = plan.fetch(want_image=False)["stats"]; print("new plan kernel_ms", round(st['kernel_ms'], 3))

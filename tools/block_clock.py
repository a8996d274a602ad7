"""Diagnostic: wave clock per block of the march loop (make -C raytrace-miniapp_amd/csrc librt_hip_time.so)."""
import ctypes as C, importlib, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
lib = be.HipLibrary(be.CSRC / "librt_hip_time.so")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
p = rt.scale_problem(base, 16.0)
with be.Plan(p, lib=lib) as plan:
    plan.set_ray_grid().run()
    out = (C.c_ulonglong * 8)()
    lib.lib.rt_hip_debug_counters(out)      # discard the warm-up
    plan.run()
    st = plan.fetch(want_image=False)["stats"]
    lib.lib.rt_hip_debug_counters(out)
v = list(out)
tot = sum(v[:6])
print("march_ms", round(st["march_ms"], 3), "wave iterations", v[7])
for lab, c in zip(("refill", "A1", "A2", "DONE", "B", "C"), v[:6]):
    print(f"  {lab:7s} {100.0 * c / tot:5.1f} %   {c / max(v[7], 1):8.1f} clk/iter")

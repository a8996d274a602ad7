"""Diagnostic: lane occupancy of the three nested march loops (instrumented build,
`make -C raytrace-miniapp_amd/csrc librt_hip_instr.so`)."""
import ctypes as C, importlib, sys
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
lib = be.HipLibrary(be.CSRC / "librt_hip_instr.so")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
cases = {"ASE_medium_standin": rt.scale_problem(base, 16.0), "ASE_small": base,
         "seed_small": rt.datfile.load('tests/golden/seed_small.dat.xz')}
for name, p in cases.items():
    with be.Plan(p, lib=lib) as plan:
        plan.set_ray_grid().run()
        st = plan.fetch(want_image=False)["stats"]
    out = (C.c_ulonglong * 8)()
    lib.lib.rt_hip_debug_counters(out)
    v = list(out)
    print(name, "rays", st["n_rays"], "kernel_ms", round(st["kernel_ms"], 3))
    for i, lab in enumerate(("inner", "cross", "cell ")):
        w, a = v[2 * i], v[2 * i + 1]
        print(f"   {lab}: wave-iters {w:>12d}  lane-iters {a:>12d}  occupancy {a / (64.0 * max(w, 1)):.3f}  per-ray lane-iters {a / st['n_rays']:.2f}")
    print(f"   tiny-dividend block of [C]: entered in {v[6]} of {v[0]} wave-iterations ({100.0 * v[6] / max(v[0], 1):.1f} %)")

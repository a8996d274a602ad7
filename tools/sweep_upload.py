"""Diagnostic: rt_hip_image_loop wall time under RT_HIP_UPLOAD_SLICES values."""
import importlib, os, sys, time
sys.path.insert(0, '.')
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
p = rt.scale_problem(base, 16.0)
rays = p.build_rays()
for v in sys.argv[1:]:
    os.environ["RT_HIP_UPLOAD_SLICES"] = v
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter(); out = be.image_loop(p, rays, device=0); best = min(best, time.perf_counter() - t0)
    print(f"slices {v}: image_loop {best*1e3:.2f} ms  kernels {out['stats']['kernel_ms']:.2f}")

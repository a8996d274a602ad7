"""Diagnostic (librt_hip_instr.so): loop iterations per ray of the march on the stand-in -- which rays are the long
ones?  Statistics by pixel column / row and by angle cell; saved to gpurun_out/ray_iters_standin.npz."""
import ctypes as C, importlib, sys
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
lib = be.HipLibrary(be.CSRC / "librt_hip_instr.so")
p = rt.scale_problem(rt.datfile.load('tests/golden/ASE_small.dat.xz'), 16.0)
b = p.beam
n = p.n_rays_total
with be.Plan(p, lib=lib) as plan:
    plan.set_ray_grid().set_debug(1).run()
    plan.fetch(want_image=False)
    out = np.zeros(n, np.uint16)
    lib.lib.rt_hip_debug_ray_iters(out.ctypes.data_as(C.POINTER(C.c_ushort)), C.c_ulonglong(n))
it = out.reshape(b.nx, b.ny, b.na, b.nb).astype(np.float64)
print("iterations per ray: mean %.1f  median %.0f  99%% %.0f  99.9%% %.0f  max %.0f" % (it.mean(), np.median(it), np.percentile(it, 99), np.percentile(it, 99.9), it.max()))
np.set_printoptions(linewidth=220, precision=0, suppress=True)
print("max over pixels, by angle cell (rows a index 0..%d step 4, cols b index 0..%d step 3):" % (b.na - 1, b.nb - 1))
print(it.max(axis=(0, 1))[::4, ::3])
print("mean over pixels, by angle cell:")
print(it.mean(axis=(0, 1))[::4, ::3])
print("max over angles, by pixel (rows x index step 10, cols y index step 5):")
print(it.max(axis=(2, 3))[::10, ::5])
print("fraction of rays above 150 iterations: %.4f; above 200: %.4f" % ((it > 150).mean(), (it > 200).mean()))
long = it > 150
print("rays > 150 by a index:", long.sum(axis=(0, 1, 3)))
print("rays > 150 by b index:", long.sum(axis=(0, 1, 2)))
print("rays > 150 by x index (step 6):", long.sum(axis=(1, 2, 3))[::6])
print("rays > 150 by y index (step 2):", long.sum(axis=(0, 2, 3))[::2])
np.savez_compressed("gpurun_out/ray_iters_standin.npz", iters=out, shape=np.array([b.nx, b.ny, b.na, b.nb]))

"""Fixed and per-ray cost of the two kernels: the stand-in at 1/8 ... 2x its size (pixel-column shards, and
scale_problem(32) for the 2x point), T = a + b n."""
import importlib, sys
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
full = rt.scale_problem(base, 16.0)
pts = []
for name, p in [("1/8", mg.shard(full, 0, 8)), ("1/4", mg.shard(full, 0, 4)), ("1/2", mg.shard(full, 0, 2)), ("1", full),
                ("2", rt.scale_problem(base, 32.0))]:
    with be.Plan(p) as plan:
        plan.set_ray_grid().set_timing_ring(10)
        for _ in range(14):
            plan.run()
        t = plan.ring_times()
        st = plan.fetch(want_image=False)["stats"]
    m, f = min(a for a, _ in t), min(b for _, b in t)
    pts.append((st["n_rays"] / 1e6, m, f))
    print(f"x{name:4s} rays {st['n_rays']:9d} steps/ray {st['cell_steps'] / st['n_rays']:.2f}  march {m:.3f} ms  freq {f:.3f} ms", flush=True)
n = np.array([p[0] for p in pts]); A = np.vstack([np.ones_like(n), n]).T
for lab, col in (("march", 1), ("freq", 2)):
    a, b = np.linalg.lstsq(A, np.array([p[col] for p in pts]), rcond=None)[0]
    print(f"{lab}: fixed {a:.3f} ms + {b:.4f} ms per million rays")

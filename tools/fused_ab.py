"""One-launch run (rt_fused.hip) against the two-kernel run of the same plan: images, counters, kernel time.

  python tools/fused_ab.py [standin|small|shard8] ..."""
import importlib, os, sys, time
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
cases = [a for a in sys.argv[1:] if not a.startswith("-")] or ["small", "standin"]
for case in cases:
    if case == "small":
        p = base
    elif case == "standin":
        p = rt.scale_problem(base, 16.0)
    elif case.startswith("shard"):
        n = int(case[5:])
        p = mg.shard(rt.scale_problem(base, 16.0), 0, n)
    res = {}
    for mode in ("2", "1"):
        os.environ["RT_HIP_FUSED"] = mode
        with be.Plan(p) as plan:
            plan.set_ray_grid()
            plan.run()
            out = plan.fetch()
            fused = plan.last_fused()
            t = []
            for _ in range(12):
                plan.run()
                m, f = plan.kernel_times()
                t.append(m + f)
            res[mode] = (out, fused, min(t), float(np.median(t)))
        print(f"{case:8s} RT_HIP_FUSED={mode} fused={fused} kernels min {min(t):.3f} ms median {np.median(t):.3f} ms "
              f"steps {out['stats']['cell_steps']} failure {out['failure_code']}", flush=True)
    a, b = res["2"][0], res["1"][0]
    s = np.abs(a["image"]).max()
    print(f"{case:8s} fused vs two-kernel: max|d image|/max {np.abs(a['image'] - b['image']).max() / s:.2e}  "
          f"max|d I_ang|/max {np.abs(a['I_ang'] - b['I_ang']).max() / np.abs(a['I_ang']).max():.2e}  "
          f"speed-up {res['2'][2] / res['1'][2]:.3f}", flush=True)
    assert res["1"][1] and not res["2"][1]

"""Express waves (rt_march.hip: RT_HIP_EXPRESS_AGE / RT_HIP_EXPRESS_HOLD) against the plain run on one plan per
workload: kernel time (min / median of 12 runs after 3), image against the plain run.

  python tools/express_ab.py [shard8 shard4 shard2 standin small] [--settings age:hold,age:hold,...]"""
import importlib, os, sys
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
full = rt.scale_problem(base, 16.0)
args = sys.argv[1:]
settings = "0:0,16:0,32:0,64:0,16:1,32:1,64:1,128:1"
if "--settings" in args:
    i = args.index("--settings")
    settings = args[i + 1]
    del args[i:i + 2]
extra = {}
if "--env" in args:
    i = args.index("--env")
    for kv in args[i + 1].split(","):
        k, v = kv.split("=")
        extra[k] = v
    del args[i:i + 2]
cases = args or ["shard8", "standin"]
for case in cases:
    if case == "small":
        p = base
    elif case == "standin":
        p = full
    else:
        n = int(case[5:].split(".")[0])
        r = int(case.split(".")[1]) if "." in case else 0
        p = mg.shard(full, r, n)
    os.environ.update(extra)
    with be.Plan(p) as plan:
        plan.set_ray_grid()
        ref = None
        for s in settings.split(","):
            age, hold, *rest = s.split(":")
            os.environ["RT_HIP_EXPRESS_AGE"] = age
            os.environ["RT_HIP_EXPRESS_HOLD"] = hold
            os.environ["RT_HIP_EXPRESS_PARK"] = rest[0] if rest else "0"
            os.environ["RT_HIP_EXPRESS_TAIL"] = rest[1] if len(rest) > 1 else "0"
            hold = hold + "".join(":" + r for r in rest)
            for _ in range(3):
                plan.run()
            out = plan.fetch()
            t = []
            for _ in range(12):
                plan.run()
                m, f = plan.kernel_times()
                t.append(m + f)
            if ref is None:
                ref = out
            d = np.abs(out["image"] - ref["image"]).max() / np.abs(ref["image"]).max()
            print(f"{case:9s} age {age:>4s} hold {hold}  min {min(t):.4f} ms  median {np.median(t):.4f} ms  fused {plan.last_fused()}  "
                  f"steps {out['stats']['cell_steps']}  max|d image|/max {d:.1e}", flush=True)

"""A/B of run-time switches of the library on one plan per workload: every env-set in turn (`;`-separated, `,` inside,
`-` = defaults), kernel time min / median of 12 runs after 3, image against the first set's.

  python tools/env_ab.py shard8 standin --envs "-;RT_HIP_MARCH_THREADS=1024,RT_HIP_FUSED_CONSUMERS=4" [--rounds 2]"""
import importlib, os, sys
sys.path.insert(0, '.')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
base = rt.datfile.load('tests/golden/ASE_small.dat.xz')
full = rt.scale_problem(base, 16.0)
args = sys.argv[1:]
def opt(name, default):
    if name in args:
        i = args.index(name); v = args[i + 1]; del args[i:i + 2]; return v
    return default
envs = opt("--envs", "-").split(";")
rounds = int(opt("--rounds", "1"))
cases = args or ["shard8"]
touched = set()
original = dict(os.environ)
def reset_env():
    for k in touched:
        if k in original:
            os.environ[k] = original[k]
        else:
            os.environ.pop(k, None)
for case in cases:
    if case == "small":
        p = base
    elif case == "standin":
        p = full
    elif case == "seed":
        p = rt.datfile.load('tests/golden/seed_small.dat.xz')
    elif case == "seedmed":
        p = rt.scale_problem(rt.datfile.load('tests/golden/seed_small.dat.xz'), 16.0)
    elif case.startswith("seedx"):
        p = rt.scale_problem(rt.datfile.load('tests/golden/seed_small.dat.xz'), float(case[5:]))
    else:
        n = int(case[5:].split(".")[0]); r = int(case.split(".")[1]) if "." in case else 0
        p = mg.shard(full, r, n)
    with be.Plan(p) as plan:
        plan.set_ray_grid()
        ref = None
        for rnd in range(rounds):
            for e in envs:
                reset_env()
                if e != "-":
                    for kv in e.split(","):
                        k, v = kv.split("="); os.environ[k] = v; touched.add(k)
                for _ in range(3):
                    plan.run()
                out = plan.fetch()
                t = []
                for _ in range(12):
                    plan.run(); m, f = plan.kernel_times(); t.append(m + f)
                if ref is None:
                    ref = out
                d = np.abs(out["image"] - ref["image"]).max() / np.abs(ref["image"]).max()
                da = np.abs(out["I_ang"] - ref["I_ang"]).max() / np.abs(ref["I_ang"]).max()
                print(f"{case:9s} {e:60s} min {min(t):.4f}  median {np.median(t):.4f} ms  fused {plan.last_fused()}  steps {out['stats']['cell_steps']}  "
                      f"d image {d:.1e} d I_ang {da:.1e} fail {out['failure_code']}", flush=True)
    reset_env()

"""Randomised grid-mode parity (the path create_image takes): uniform ray grids of random sizes, random N / K / dz,
no probe -- own-cell deposits, the exclusive mode (na = nb = 1), few-runs / row-cache / scan deposits, launches with
fewer tiles than counter shards, through plan.run() and through the host-pointer entry image_loop().
image, I_ang, failure code and ray-step count against the oracle:  python tools/fuzz_grid.py first last"""
import copy, importlib, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
pm = importlib.import_module("raytrace-miniapp_amd.problem")
from oracle.binding import Oracle

a = rt.datfile.load('tests/golden/ASE_small.dat.xz'); s = rt.datfile.load('tests/golden/seed_small.dat.xz')
ora = Oracle()
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
worst = {False: 0.0, True: 0.0}
for seed in range(first, last):
    rng = np.random.default_rng(77000 + seed)
    seeded = rng.random() < 0.4
    p = copy.copy(s if seeded else a)
    N = int(rng.integers(2, 5))
    gains = [p.gain[0]]
    for i in range(N - 1):
        g = p.gain[1 + int(rng.integers(0, 2))]
        gains.append(rt.Gain(g.x, g.y, g.n, g.g0 * np.float32(rng.uniform(0.3, 1.5)), g.E0, g.gv, g.Nv))
    p.gain = gains
    if rng.random() < 0.5:
        p = pm.resample_frequency(p, int(rng.choice([3, 5, 18, 52, 64, 66, 100])))
    if rng.random() < 0.3:
        p.beam = copy.copy(p.beam)
        p.beam.dz = float(p.beam.dz * rng.uniform(0.5, 2.0))
    shape = rng.choice(["tiny", "flat", "one_angle", "wide"])
    if shape == "tiny":
        n = dict(nx=int(rng.integers(1, 4)), ny=int(rng.integers(1, 4)), na=int(rng.integers(1, 4)), nb=int(rng.integers(1, 4)))
    elif shape == "flat":
        n = dict(nx=int(rng.integers(2, 30)), ny=int(rng.integers(1, 12)), na=int(rng.integers(1, 9)), nb=int(rng.integers(1, 9)))
    elif shape == "one_angle":
        n = dict(nx=int(rng.integers(3, 60)), ny=int(rng.integers(2, 40)), na=1, nb=1)
    else:
        n = dict(nx=int(rng.integers(1, 6)), ny=int(rng.integers(1, 6)), na=int(rng.integers(5, 40)), nb=int(rng.integers(5, 30)))
    p = pm.regrid_seed_beam(p, **n) if seeded else pm.regrid_beam(p, **n)
    rays = p.build_rays()
    ref = ora.image_loop(p, rays)
    outs = {}
    with be.Plan(p) as plan:
        outs["plan"] = plan.set_ray_grid().run().fetch()
    if seed % 3 == 0:
        outs["image_loop"] = be.image_loop(p, rays)
    for how, out in outs.items():
        ok = out["failure_code"] == ref["failure_code"]
        if "stats" in out and "cell_steps" in out["stats"]:
            ok = ok and out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
        err = 0.0
        if ok and ref["failure_code"] == 0:
            for key in ("image", "I_ang"):
                nr = np.linalg.norm(ref[key])
                if nr > 0:
                    e = float(np.linalg.norm(np.asarray(out[key]).ravel() - np.asarray(ref[key]).ravel()) / nr)
                    err = max(err, e)
                else:
                    ok = ok and not np.asarray(out[key]).any()
            ok = ok and err < (1e-10 if seeded else 2e-7)
            worst[seeded] = max(worst[seeded], err)
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, how, "seeded", seeded, "N", N, "K", p.beam.nv, n, "rays", len(rays), "err", err,
                  "codes", out["failure_code"], ref["failure_code"], flush=True)
print(f"cases {last - first}, mismatches {bad}, worst image / I_ang rel-L2: emission {worst[False]:.2e}, seeded {worst[True]:.2e}")

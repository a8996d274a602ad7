"""Randomised grid-mode parity (the path create_image takes): uniform ray grids of random sizes, random N / K / dz,
no probe -- own-cell deposits, the exclusive mode (na = nb = 1), few-runs / row-cache / scan deposits, launches with
fewer tiles than counter shards, through plan.run() and through the host-pointer entry image_loop().
image, I_ang, failure code and ray-step count against the oracle (the cases of
tests/test_gpu_fuzz.py::test_random_uniform_grids_match_oracle, any number of them):
    python tools/fuzz_grid.py first last"""
import importlib, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
from oracle.binding import Oracle
from test_gpu_fuzz import random_grid_case, check_grid_case

a = rt.datfile.load('tests/golden/ASE_small.dat.xz'); s = rt.datfile.load('tests/golden/seed_small.dat.xz')
ora = Oracle()
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
n_fused = 0   # cases that took the one-launch run (rt_fused.hip)
worst = {False: 0.0, True: 0.0}
for seed in range(first, last):
    rng = np.random.default_rng(77000 + seed)
    p, n = random_grid_case(rng, a, s)
    seeded = p.seed is not None
    rays = p.build_rays()
    ref = ora.image_loop(p, rays)
    outs = {}
    with be.Plan(p) as plan:
        outs["plan"] = plan.set_ray_grid().run().fetch()
        n_fused += plan.last_fused()
    if seed % 3 == 0:
        outs["image_loop"] = be.image_loop(p, rays)
    if (seed - first) % 2000 == 1999:
        print(f"... {seed - first + 1} cases, {bad} mismatches so far", flush=True)
    for how, out in outs.items():
        ok, err = check_grid_case(out, ref, seeded)
        worst[seeded] = max(worst[seeded], err)
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, how, "seeded", seeded, "N", p.N, "K", p.beam.nv, n, "rays", len(rays), "err", err,
                  "codes", out["failure_code"], ref["failure_code"], flush=True)
print(f"one-launch runs: {n_fused} of {last - first} plans")
print(f"cases {last - first}, mismatches {bad}, worst image / I_ang rel-L2: emission {worst[False]:.2e}, seeded {worst[True]:.2e}")

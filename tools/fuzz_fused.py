"""Randomised parity of the one-launch run (rt_fused.hip): emission-mode problems on the beam's own ray grid with at
least 32 rays per pixel -- random grid sizes (work-group sizes 512 / 768 / 1024, 2 or 3 pixel runs per tile), N = 2 / 3,
frequency counts, gains, dz, sub-ranges and strides of the ray grid -- image, I_ang, failure code and ray-step count
against the oracle; every plan must report the one-launch run.
    python tools/fuzz_fused.py first last"""
import copy, importlib, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
pm = importlib.import_module("raytrace-miniapp_amd.problem")
from oracle.binding import Oracle
from test_gpu_fuzz import check_grid_case

a = rt.datfile.load('tests/golden/ASE_small.dat.xz')
ora = Oracle()
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = n_fused = 0
worst = 0.0
for seed in range(first, last):
    rng = np.random.default_rng(91000 + seed)
    p = copy.copy(a)
    N = int(rng.integers(2, 4))
    gains = [p.gain[0]]
    for i in range(N - 1):
        g = p.gain[1 + int(rng.integers(0, 2))]
        gains.append(rt.Gain(g.x, g.y, g.n, g.g0 * np.float32(rng.uniform(0.3, 1.5)), g.E0, g.gv, g.Nv))
    p.gain = gains
    if rng.random() < 0.5:
        p = pm.resample_frequency(p, int(rng.choice([3, 5, 18, 52, 64, 66, 100, 130])))
    if rng.random() < 0.3:
        p.beam = copy.copy(p.beam)
        p.beam.dz = float(p.beam.dz * rng.uniform(0.5, 2.0))
    na, nb = int(rng.integers(4, 30)), int(rng.integers(4, 30))
    while na * nb < 32:
        na += 1
    big = rng.random() < 0.15                                   # now and then enough rays for 768 / 1024-thread work-groups
    nx = int(rng.integers(20, 60)) if big else int(rng.integers(1, 12))
    ny = int(rng.integers(10, 30)) if big else int(rng.integers(1, 8))
    p = pm.regrid_beam(p, nx=nx, ny=ny, na=na, nb=nb)
    total = p.n_rays_total
    kind = rng.integers(0, 4)
    first_r, stride = 0, 1
    if kind == 1:
        first_r = int(rng.integers(0, min(total, 500)))
    elif kind == 2:
        stride = int(rng.integers(2, 6))
    count = (total - first_r + stride - 1) // stride
    if kind == 3:
        count = int(rng.integers(1, count + 1))
    ids = first_r + stride * np.arange(count, dtype=np.int64)
    ref = ora.image_loop(p, p.build_rays(ids))
    with be.Plan(p) as plan:
        out = plan.set_ray_grid(first=first_r, stride=stride, count=count).run().fetch()
        n_fused += plan.last_fused()
    ok, err = check_grid_case(out, ref, False)
    worst = max(worst, err)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "N", N, "K", p.beam.nv, (nx, ny, na, nb), "range", (first_r, stride, count), "err", err,
              "codes", out["failure_code"], ref["failure_code"], flush=True)
    if (seed - first) % 500 == 499:
        print(f"... {seed - first + 1} cases, {bad} mismatches so far", flush=True)
print(f"one-launch runs: {n_fused} of {last - first} plans")
print(f"cases {last - first}, mismatches {bad}, worst image / I_ang rel-L2 {worst:.2e}")

"""Edge cases of the HIP path against the oracle, through the C ABI."""
import copy
import importlib
import os

import numpy as np
import pytest

from conftest import rel_l2

rt = importlib.import_module("raytrace-miniapp_amd")
problem_mod = importlib.import_module("raytrace-miniapp_amd.problem")
pytestmark = pytest.mark.gpu
TOL = 1e-5
TIGHT = 2e-7   # one float rounding of es/gs per sub-segment (frequency kernel); seeded gain-only mode stays ~1e-14


def run_hip(hip, p, rays=None, probe=False):
    with hip.Plan(p) as plan:
        if rays is None:
            plan.set_ray_grid()
        else:
            plan.set_rays(rays)
        if probe:
            plan.enable_probe()
        out = plan.run().fetch()
        if probe:
            out["probe"] = plan.fetch_probe()
    return out


def same_record(pr, ora):
    assert np.array_equal(pr["steps"], ora["steps"])
    assert np.array_equal(pr["flags"] & 3, ora["flags"] & 3)
    assert np.array_equal(pr["ivl"], ora["ivl"])
    assert np.array_equal(pr["gvl"].view(np.uint32), ora["gvl"].view(np.uint32))
    assert np.array_equal(pr["evl"].view(np.uint32), ora["evl"].view(np.uint32))


def test_empty_ray_list(hip, ase_small):
    out = run_hip(hip, ase_small, ase_small.build_rays(np.arange(0)))
    assert out["stats"]["n_rays"] == 0 and out["failure_code"] == 0
    assert not out["image"].any() and not out["I_ang"].any()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 130, 1000])
def test_ragged_ray_counts(hip, oracle, ase_small, n):
    ids = 200000 + np.arange(n, dtype=np.int64)
    rays = ase_small.build_rays(ids)
    out = run_hip(hip, ase_small, rays)
    ref = oracle.image_loop(ase_small, rays)
    assert out["stats"]["n_rays"] == n
    assert out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
    assert rel_l2(out["image"], ref["image"]) < TIGHT and rel_l2(out["I_ang"], ref["I_ang"]) < TIGHT


def test_rays_outside_plasma_and_outside_image(hip, oracle, ase_small):
    rays = np.zeros(5, dtype=rt.cabi.RAY_DTYPE)
    rays["x"] = [10.0, -10.0, 0.01, 0.01, 0.0]
    rays["y"] = [0.0, 0.0, 5.0, -5.0, 0.0]
    rays["a"] = [0, 0, 0, 0, 500.0]       # last: outside the angular grid
    out = run_hip(hip, ase_small, rays, probe=True)
    ref = oracle.image_loop(ase_small, rays)
    ora = oracle.probe(ase_small, rays, want_Iv=False)
    same_record(out["probe"], ora)
    assert np.array_equal(out["image"], ref["image"]) or rel_l2(out["image"], ref["image"]) < TIGHT
    assert out["failure_code"] == ref["failure_code"]


def test_ray_list_equals_device_generated_grid(hip, ase_small):
    p = rt.scale_problem(ase_small, 0.5)
    a = run_hip(hip, p)
    b = run_hip(hip, p, p.build_rays())
    assert a["stats"]["cell_steps"] == b["stats"]["cell_steps"]
    assert rel_l2(a["image"], b["image"]) < 1e-13 and rel_l2(a["I_ang"], b["I_ang"]) < 1e-13


def test_strided_decomposition_sums_to_whole(hip, ase_small):
    p = rt.scale_problem(ase_small, 0.5)
    whole = run_hip(hip, p)
    img = np.zeros_like(whole["image"])
    ang = np.zeros_like(whole["I_ang"])
    steps = 0
    for start in range(3):
        q = copy.copy(p)
        q.N_start, q.N_parallel = start, 3
        part = run_hip(hip, q)
        img += part["image"]
        ang += part["I_ang"]
        steps += part["stats"]["cell_steps"]
    assert steps == whole["stats"]["cell_steps"]
    assert rel_l2(img, whole["image"]) < 1e-13 and rel_l2(ang, whole["I_ang"]) < 1e-13


@pytest.mark.parametrize("nv", [64, 65, 130, 512])
def test_more_than_64_frequencies(hip, oracle, ase_small, nv):
    """K > 64 walks the frequency axis in 64-lane chunks (the reference stops at K_MAX = 100)."""
    p = problem_mod.resample_frequency(ase_small, nv)
    ids = np.arange(0, p.n_rays_total, 211, dtype=np.int64)
    rays = p.build_rays(ids)
    out = run_hip(hip, p, rays)
    ref = oracle.image_loop(p, rays)
    assert out["failure_code"] == 0
    assert rel_l2(out["image"], ref["image"]) < TIGHT and rel_l2(out["I_ang"], ref["I_ang"]) < TIGHT


@pytest.mark.parametrize("N", [2, 5, 9])
def test_other_numbers_of_lengths(hip, oracle, ase_small, N):
    p = copy.copy(ase_small)
    g = ase_small.gain
    p.gain = [g[0]] + [g[1 + (i % 2)] for i in range(N - 1)]
    ids = np.arange(0, p.n_rays_total, 397, dtype=np.int64)
    rays = p.build_rays(ids)
    out = run_hip(hip, p, rays, probe=True)
    ora = oracle.probe(p, rays, want_Iv=False)
    same_record(out["probe"], ora)
    ref = oracle.image_loop(p, rays)
    assert rel_l2(out["image"], ref["image"]) < TIGHT and rel_l2(out["I_ang"], ref["I_ang"]) < TIGHT


def test_two_sided_y_grid_takes_the_unmirrored_branch(hip, oracle, ase_small):
    """Helper.h:449-453: mirror_y only when the gain grid starts at y >= 0."""
    p = copy.copy(ase_small)
    gains = [ase_small.gain[0]]
    for g in ase_small.gain[1:]:
        Nx, Ny, K = g.Nx, g.Ny, g.Nv
        y2 = np.concatenate([-g.y[::-1] - 1e-6, g.y])
        def mir(a, w=1):
            a = a.reshape(Ny, Nx * w)
            return np.concatenate([a[::-1], a], axis=0).reshape(-1)
        gains.append(rt.Gain(g.x, y2, mir(g.n), mir(g.g0), mir(g.E0), mir(g.gv, K), K))
    p.gain = gains
    ids = np.arange(0, p.n_rays_total, 101, dtype=np.int64)
    rays = p.build_rays(ids)
    rays["y"][::2] *= -1
    out = run_hip(hip, p, rays, probe=True)
    ora = oracle.probe(p, rays, want_Iv=False)
    same_record(out["probe"], ora)
    ref = oracle.image_loop(p, rays)
    assert rel_l2(out["image"], ref["image"]) < TIGHT


def same_outputs_in_a_failing_run(out, ref, tol=1e-6):
    """Failing rays deposit nothing on the CPU (RayTraceImageCPU.cpp:29-36: `continue` before the deposit);
    the backend repeats its frequency pass without them (include/rt_hip.h), so image and I_ang of a failing
    run are the CPU's: same non-finite entries (an overflow to +inf is not an error on the CPU either), same
    finite values."""
    for key in ("image", "I_ang"):
        a, b = out[key], ref[key]
        fa, fb = np.isfinite(a), np.isfinite(b)
        assert np.array_equal(fa, fb), key
        assert np.array_equal(np.isnan(a), np.isnan(b)), key
        assert np.array_equal(a[~fa & ~np.isnan(a)], b[~fb & ~np.isnan(b)]), key   # signed infinities
        nb = np.linalg.norm(b[fb])
        assert np.linalg.norm(a[fa] - b[fb]) <= tol * nb if nb > 0 else np.all(a[fa] == 0), key


def test_failure_codes_match_the_cpu_loop(hip, oracle, ase_small):
    ids = np.arange(0, ase_small.n_rays_total, 997, dtype=np.int64)
    rays = ase_small.build_rays(ids)
    # error -1: a ray launched almost perpendicular to z (s.z^2 < 0.01, Helper.h:515)
    bad = rays.copy()
    bad["a"][7] = 1500.0
    out, ref = run_hip(hip, ase_small, bad), oracle.image_loop(ase_small, bad)
    assert out["failure_code"] == ref["failure_code"] == 1 << 1
    assert len(out["failed_rays"]) == 1 and out["failed_rays"][0] == bad[7]
    # error -3: NaNs in the lineshape
    p = copy.copy(ase_small)
    g = ase_small.gain[2]
    gv = g.gv.copy()
    gv[::7] = np.nan
    p.gain = ase_small.gain[:2] + [rt.Gain(g.x, g.y, g.n, g.g0, g.E0, gv, g.Nv)]
    out, ref = run_hip(hip, p, rays), oracle.image_loop(p, rays)
    assert ref["failure_code"] & (1 << 3) and out["failure_code"] == ref["failure_code"]
    same_outputs_in_a_failing_run(out, ref)
    # error -2: negative lineshape -> negative intensity
    gv = -np.abs(g.gv)
    p.gain = ase_small.gain[:2] + [rt.Gain(g.x, g.y, g.n, g.g0, g.E0, gv, g.Nv)]
    out, ref = run_hip(hip, p, rays), oracle.image_loop(p, rays)
    assert ref["failure_code"] & (1 << 2) and out["failure_code"] == ref["failure_code"]
    same_outputs_in_a_failing_run(out, ref)
    with pytest.raises(hip.RayTraceError, match="Some rays failed"):
        hip.create_image(p, "hip")


def test_failing_runs_through_the_host_pointer_entry(hip, oracle, ase_small):
    """rt_hip_image_loop queues the download of its (small) outputs behind the kernels; a run with failing rays repeats
    its frequency pass afterwards, so what was queued is stale and must not be what the caller gets -- for a list traced
    as a list and for the whole grid handed over as a list (recognised, generated on the device)."""
    p = copy.copy(ase_small)
    g = ase_small.gain[2]
    for gv in (-np.abs(g.gv), np.where(np.arange(g.gv.size) % 7 == 0, np.nan, g.gv).astype(np.float32)):
        p.gain = ase_small.gain[:2] + [rt.Gain(g.x, g.y, g.n, g.g0, g.E0, gv, g.Nv)]
        ids = np.arange(0, ase_small.n_rays_total, 997, dtype=np.int64)
        for rays in (ase_small.build_rays(ids), ase_small.build_rays()):
            ref = oracle.image_loop(p, rays, n_threads=8)
            out = hip.image_loop(p, rays)
            assert ref["failure_code"] & ((1 << 2) | (1 << 3)) and out["failure_code"] == ref["failure_code"]
            same_outputs_in_a_failing_run(out, ref)
    # and a clean call right after, on the staging the failing ones used
    ok = hip.image_loop(ase_small)
    want = run_hip(hip, ase_small)
    assert ok["failure_code"] == 0 and rel_l2(ok["image"], want["image"]) < 1e-12 and rel_l2(ok["I_ang"], want["I_ang"]) < 1e-12


_SPIN_CHILD = """
import copy, importlib, sys
sys.path.insert(0, {root!r})
import numpy as np
rt = importlib.import_module("raytrace-miniapp_amd")
hip = importlib.import_module("raytrace-miniapp_amd.backend")
base = rt.datfile.load({dat!r})
p = copy.copy(base)
# no refraction at all (n = 1 everywhere) and an infinite segment length: a ray along z never reaches the end of its
# sub-segment and never leaves the plasma -- Helper.h:279-311 steps 1250 cm at a time towards z = inf
p.gain = [base.gain[0]] + [rt.Gain(g.x, g.y, np.ones_like(g.n), g.g0, g.E0, g.gv, g.Nv) for g in base.gain[1:]]
p.beam = copy.copy(base.beam)
p.beam.dz = float("inf")
rays = np.zeros(3, dtype=rt.cabi.RAY_DTYPE)
rays["x"] = 0.5 * (base.gain[1].x[0] + base.gain[1].x[-1])
rays["y"] = [0.3 * base.gain[1].y[-1], 0.5 * base.gain[1].y[-1], 0.7 * base.gain[1].y[-1]]
with hip.Plan(p) as plan:
    out = plan.set_rays(rays).run().fetch()
print("LIST", out["failure_code"], len(out["failed_rays"]), float(np.abs(out["image"]).max()), flush=True)
# the same through the one-launch run: a whole grid whose launch angles include (0, 0)
pm = importlib.import_module("raytrace-miniapp_amd.problem")
q = pm.regrid_beam(p, nx=12, ny=6)
q.beam.a = q.beam.da * (np.arange(7) - 3.0)            # 7 x 5 launch angles around (0, 0): one ray per pixel spins
q.beam.b = q.beam.db * (np.arange(5) - 2.0)
with hip.Plan(q) as plan:
    out = plan.set_ray_grid().run().fetch()
    print("GRID", out["failure_code"], out["stats"]["n_rays"], plan.last_fused(), flush=True)
# slot exhaustion of the one-launch run (ADVICE round 4): 64 launch angles per pixel, (0, 0) among them, so that EVERY
# 64-ray tile holds exactly one ray that never advances; one wave per work-group and 40 tiles per wave, so that all 32
# tile slots of a wave end up held open by such tiles while eight or more of its lanes are idle -- refills that hand
# out nothing must not restart the watchdog
import os
os.environ["RT_HIP_MARCH_THREADS"] = "64"
q = pm.regrid_beam(p, nx=160, ny=64)                   # 10 240 pixels = tiles: 40 per wave on 256 CUs
q.beam.a = q.beam.da * (np.arange(8) - 4.0)
q.beam.b = q.beam.db * (np.arange(8) - 4.0)
with hip.Plan(q) as plan:
    out = plan.set_ray_grid().run().fetch()
    print("SLOTS", out["failure_code"], out["stats"]["n_rays"], plan.last_fused(), flush=True)
del os.environ["RT_HIP_MARCH_THREADS"]
# and the library is fine afterwards
ok = hip.image_loop(base)
print("AFTER", ok["failure_code"], ok["stats"]["cell_steps"], flush=True)
"""


def test_a_ray_that_never_advances_is_given_up_not_marched_for_ever():
    """The reference's loops have no iteration limit: a ray whose steps do not advance (here: an infinite dz in a medium
    without refraction) keeps the CPU busy for ever.  On the device that would be a hung GPU; the march gives a wave's
    rays up as invalid (error -1) after RT_HIP_MARCH_SPIN_LIMIT iterations without a single ray retiring or arriving
    (default 2^24; lowered here).  Run in a child process under a time limit: a hang is the failure this guards against."""
    import subprocess
    import sys
    from conftest import GOLDEN, ROOT
    env = dict(os.environ)
    env["RT_HIP_MARCH_SPIN_LIMIT"] = "4096"
    code = _SPIN_CHILD.format(root=str(ROOT), dat=str(GOLDEN / "ASE_small.dat.xz"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "LIST 2 3 0.0" in r.stdout, r.stdout
    grid = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("GRID")][0]
    assert int(grid[1]) & 2 and int(grid[2]) > 0, r.stdout
    slots = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("SLOTS")][0]
    assert int(slots[1]) & 2 and int(slots[2]) >= 32 * 64 * 256 and slots[3] == "True", r.stdout
    assert "AFTER 0 4768067" in r.stdout, r.stdout


def _ray_set(rays):
    return sorted(tuple(np.asarray(r.tolist(), dtype=np.float32).view(np.uint32).tolist()) for r in rays)


@pytest.mark.parametrize("bad_value", [np.nan, np.inf])
def test_non_finite_lineshape_in_row_zero_fails_rays_that_never_entered_the_plasma(hip, oracle, ase_small, bad_value):
    """Helper.h:543-594: the ASE loop multiplies gvl * gv[k + ivl * K] for EVERY sub-segment, entered or not; a ray
    that escaped at once has gvl = 0, ivl = 0 everywhere, reads row 0 of every length's table and fails with
    0 * NaN (or 0 * inf) -> error -3 on the CPU.  The march marks all-zero records `skip` (nothing to integrate)
    only while every lineshape value is finite; here the only non-finite value sits in row 0 of ONE length and the
    only rays that meet it are rays that never entered the plasma: failure code and failed rays must be the CPU's."""
    p = copy.copy(ase_small)
    g = ase_small.gain[1]
    gv = g.gv.copy()
    gv[3] = bad_value                                   # cell 0, frequency 3 of length 1 -- nothing else
    p.gain = [ase_small.gain[0], rt.Gain(g.x, g.y, g.n, g.g0, g.E0, gv, g.Nv), ase_small.gain[2]]
    rays = np.zeros(20, dtype=rt.cabi.RAY_DTYPE)
    rays["x"] = 10.0 + np.arange(20)                    # far outside the plasma: escape at the first test
    rays["y"] = 0.01
    rays["a"] = np.linspace(-3, 3, 20)
    out, ref = run_hip(hip, p, rays), oracle.image_loop(p, rays)
    assert ref["failure_code"] == 1 << 3, "the CPU loop fails these rays with error -3"
    assert out["failure_code"] == ref["failure_code"]
    assert _ray_set(out["failed_rays"]) == _ray_set(ref["failed_rays"])
    same_outputs_in_a_failing_run(out, ref)
    # the same table with rays that do march: those whose unvisited sub-segments (or visited cells) read row 0
    ids = np.arange(0, ase_small.n_rays_total, 1499, dtype=np.int64)
    mixed = np.concatenate([ase_small.build_rays(ids), rays[:3]])
    out, ref = run_hip(hip, p, mixed), oracle.image_loop(p, mixed)
    assert out["failure_code"] == ref["failure_code"]
    if len(ref["failed_rays"]) < rt.cabi.RT_N_FAILED_MAX:
        assert _ray_set(out["failed_rays"]) == _ray_set(ref["failed_rays"])
    same_outputs_in_a_failing_run(out, ref)
    # and a finite table still skips them: no failure, counters say so
    out = run_hip(hip, ase_small, rays)
    assert out["failure_code"] == 0 and out["stats"]["n_skipped"] == len(rays)


def test_rays_with_a_nan_start_are_reported_as_invalid_not_marched(hip, oracle, ase_small):
    """A NaN or infinite launch angle (tanf gives NaN for both) or a NaN position inside the plasma: every
    comparison of the reference's escape test and cell box fails on it, z never advances and its cell loop
    (Helper.h:463-504) never ends for N >= 3 -- there is no CPU result to compare with.  The backend reports such
    a ray as an invalid ray (error -1) and marches the others as if it were not there.  The same NaN ray
    launched outside the plasma escapes at once in the reference too and is no error."""
    ids = np.arange(0, ase_small.n_rays_total, 1999, dtype=np.int64)
    good = ase_small.build_rays(ids)
    wild = good[:4].copy()
    g1 = ase_small.gain[1]
    wild["x"], wild["y"] = 0.5 * (g1.x[0] + g1.x[-1]), 0.5 * g1.y[-1]       # inside the plasma
    wild["a"] = [np.nan, np.inf, 1.0, 1.0]
    wild["b"] = [0.0, 0.0, -np.inf, 0.0]
    wild["x"][3] = np.nan
    outside = good[:2].copy()
    outside["x"], outside["a"] = 10.0, [np.nan, np.inf]
    for march_ieee in (False, True):
        if march_ieee:
            os.environ["RT_HIP_MARCH_IEEE"] = "1"
        try:
            out = run_hip(hip, ase_small, np.concatenate([good, wild, outside]))
        finally:
            os.environ.pop("RT_HIP_MARCH_IEEE", None)
        ref = oracle.image_loop(ase_small, np.concatenate([good, outside]))
        assert out["failure_code"] == 1 << 1 and ref["failure_code"] == 0
        assert _ray_set(out["failed_rays"]) == _ray_set(wild)
        assert out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
        assert rel_l2(out["image"], ref["image"]) < TIGHT and rel_l2(out["I_ang"], ref["I_ang"]) < TIGHT


def test_host_pointer_entry_and_create_image_arms(hip, ase_small, ase_ref):
    p = rt.scale_problem(ase_small, 0.3)
    a = hip.image_loop(p)
    b = hip.create_image(p, "hip")
    c = hip.create_image(p, "Hip-MultiGPU")
    d = hip.create_image(p, "auto", device_rays=False)
    for o in (b, c, d):
        assert rel_l2(o["image"], a["image"]) < 1e-13 and rel_l2(o["I_ang"], a["I_ang"]) < 1e-13
    assert c["stats"]["n_rays"] == p.n_rays_total
    with pytest.raises(hip.RayTraceError, match="Unknown method"):
        hip.create_image(p, "cuda")
    full = hip.create_image(ase_small)
    assert rel_l2(full["image"], ase_ref["image"]) < TIGHT


def test_seeded_subsets_and_list_path(hip, oracle, seed_small):
    ids = np.arange(3, seed_small.n_rays_total, 499, dtype=np.int64)
    rays = seed_small.build_rays(ids)
    out = run_hip(hip, seed_small, rays, probe=True)
    ref = oracle.image_loop(seed_small, rays)
    ora = oracle.probe(seed_small, rays, want_Iv=False)
    same_record(out["probe"], ora)
    ok = ora["err"] == 0
    # exit ray: positions and angles bit-exact (atanf restated from the reference platform's libm)
    for key in "xyab":
        assert np.array_equal(out["probe"]["ray2"][key][ok].view(np.uint32), ora["ray2"][key][ok].view(np.uint32))
    assert rel_l2(out["image"], ref["image"]) < TOL and rel_l2(out["I_ang"], ref["I_ang"]) < TOL


def test_path_tracer_matches_calc_ray_path(hip, oracle):
    """(f-4) RayTrace::calc_ray_path on the GPU: positions bit-exact, intensities to float
    rounding, against the reference's own outputs (fixture) and the oracle, two step factors."""
    import importlib
    from conftest import GOLDEN
    for name in ("ASE_small", "seed_small"):
        p = rt.datfile.load(GOLDEN / f"{name}.dat.xz")
        fx = np.load(GOLDEN / f"{name}_ref_path.npz")
        i0, n = fx["i0"], fx["n"]
        gx, gy, ga, gb = p.ray_grid
        sub = [g[s:s + c] for g, s, c in zip((gx, gy, ga, gb), i0, n)]
        for c in (0.5, 0.25):
            xr, yr, Ir, nerr = hip.calc_ray_path(p, *sub, c=c)
            assert nerr == int(fx[f"nerr_c{c}"])
            assert np.array_equal(xr.view(np.uint32), fx[f"x_c{c}"].view(np.uint32))
            assert np.array_equal(yr.view(np.uint32), fx[f"y_c{c}"].view(np.uint32))
            assert np.allclose(Ir, fx[f"I_c{c}"], rtol=2e-6, atol=0)
            assert float(np.abs(fx[f"I_c{c}"]).max()) > 0


def test_path_tracer_escaped_and_failed_rays(hip, oracle, ase_small):
    rays = ase_small.build_rays(np.arange(0, ase_small.n_rays_total, 331, dtype=np.int64))
    rays["a"][5] = 1500.0      # error -1
    rays["x"][9] = 10.0        # outside the plasma from the start
    with hip.Plan(ase_small) as plan:
        got = plan.enable_path().set_rays(rays).run().fetch_path()
    want = oracle.calc_ray_path(ase_small, rays)
    assert np.array_equal(got["err"], want["err"]) and got["err"][5] == -1
    assert np.array_equal(got["x"].view(np.uint32), want["x"].view(np.uint32))
    assert np.array_equal(got["y"].view(np.uint32), want["y"].view(np.uint32))
    assert np.allclose(got["I"], want["I"], rtol=2e-6, atol=0)


def test_non_uniform_gain_grid_takes_the_bisection_fallback(hip, oracle, ase_small):
    """The arithmetic index guess is only exact on uniform grids; a stretched grid must
    give the same cells through the bisection fallback (Helper.h:131-143)."""
    p = copy.copy(ase_small)
    gains = [ase_small.gain[0]]
    for g in ase_small.gain[1:]:
        t = np.linspace(0.0, 1.0, g.Nx)
        x2 = g.x[0] + (g.x[-1] - g.x[0]) * (0.35 * t + 0.65 * t ** 3)       # strongly non-uniform, monotone
        s = np.linspace(0.0, 1.0, g.Ny)
        y2 = g.y[0] + (g.y[-1] - g.y[0]) * (0.5 * s + 0.5 * s ** 2)
        gains.append(rt.Gain(x2, y2, g.n, g.g0, g.E0, g.gv, g.Nv))
    p.gain = gains
    rays = p.build_rays(np.arange(0, p.n_rays_total, 173, dtype=np.int64))
    out = run_hip(hip, p, rays, probe=True)
    same_record(out["probe"], oracle.probe(p, rays, want_Iv=False))
    ref = oracle.image_loop(p, rays)
    assert rel_l2(out["image"], ref["image"]) < TIGHT and rel_l2(out["I_ang"], ref["I_ang"]) < TIGHT


@pytest.mark.parametrize("nv", [1, 3, 6, 50])
def test_small_and_odd_frequency_counts(hip, oracle, ase_small, nv):
    """VEC = 1 (odd K), 2 (K = 2 mod 4) and 4 code paths of the frequency kernel."""
    p = problem_mod.resample_frequency(ase_small, nv) if nv > 1 else None
    if p is None:
        p = copy.copy(ase_small)
        p.beam = copy.copy(ase_small.beam)
        p.beam.dv = np.ascontiguousarray(ase_small.beam.dv[20:21])
        p.gain = [rt.Gain(g.x, g.y, g.n, g.g0, g.E0, g.gv.reshape(-1, 52)[:, 20:21].copy(), 1) for g in ase_small.gain]
    rays = p.build_rays(np.arange(0, p.n_rays_total, 257, dtype=np.int64))
    out = run_hip(hip, p, rays)
    ref = oracle.image_loop(p, rays)
    assert out["failure_code"] == 0
    assert rel_l2(out["image"], ref["image"]) < TIGHT and rel_l2(out["I_ang"], ref["I_ang"]) < TIGHT


def test_deposit_modes_of_the_seeded_pass(hip, oracle, seed_small):
    """The three reductions behind the seeded deposit, each against the oracle:
    (i) rays in random order -- equal pixels are not contiguous in a tile, many distinct pixels:
        row cache keyed by distinct pixels, or the segmented scan when a tile has more than rows;
    (ii) a frequency axis too long for cache rows in LDS (nv = 300): segmented scan only;
    (iii) the grid-mode seed factor tables against the per-ray evaluation of the list path."""
    rng = np.random.default_rng(7)
    ids = np.sort(rng.permutation(seed_small.n_rays_total)[:60000]).astype(np.int64)
    shuffled = rng.permutation(ids)
    rays = seed_small.build_rays(shuffled)
    out, ref = run_hip(hip, seed_small, rays), oracle.image_loop(seed_small, rays)
    assert out["failure_code"] == ref["failure_code"] == 0
    assert rel_l2(out["image"], ref["image"]) < 1e-11 and rel_l2(out["I_ang"], ref["I_ang"]) < 1e-11
    wide = problem_mod.resample_frequency(seed_small, 300)
    rays = wide.build_rays(ids[::3])
    out, ref = run_hip(hip, wide, rays), oracle.image_loop(wide, rays)
    assert rel_l2(out["image"], ref["image"]) < 1e-11 and rel_l2(out["I_ang"], ref["I_ang"]) < 1e-11
    sub = copy.copy(seed_small)
    sub.seed_beam = copy.copy(seed_small.seed_beam)
    sub.seed_beam.x = seed_small.seed_beam.x[10:14].copy()     # 4 x 25 x 51 x 51 launch points and angles
    grid = run_hip(hip, sub)
    lst = run_hip(hip, sub, sub.build_rays())
    ref = oracle.image_loop(sub, n_threads=4)
    assert np.array_equal(grid["image"], lst["image"]) or rel_l2(grid["image"], lst["image"]) < 1e-13
    assert rel_l2(grid["image"], ref["image"]) < 1e-11 and rel_l2(grid["I_ang"], ref["I_ang"]) < 1e-11


def test_sliced_ray_upload_of_the_host_pointer_entry(hip, oracle, ase_small, monkeypatch):
    """rt_hip_image_loop uploads a long ray list in slices beside the march (one march launch per
    slice, own ray counter each); forced here on a short list, ragged slice sizes included."""
    ids = np.arange(5, ase_small.n_rays_total, 7, dtype=np.int64)[:50001]
    rays = ase_small.build_rays(ids)
    monkeypatch.setenv("RT_HIP_UPLOAD_SLICES", "1")
    whole = hip.image_loop(ase_small, rays)
    for n in ("2", "3", "7"):
        monkeypatch.setenv("RT_HIP_UPLOAD_SLICES", n)
        out = hip.image_loop(ase_small, rays)
        assert out["stats"]["n_rays"] == len(rays) and out["stats"]["cell_steps"] == whole["stats"]["cell_steps"]
        assert rel_l2(out["image"], whole["image"]) < 1e-13 and rel_l2(out["I_ang"], whole["I_ang"]) < 1e-13
    ref = oracle.image_loop(ase_small, rays)
    assert whole["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
    assert rel_l2(whole["image"], ref["image"]) < TIGHT


def test_gain_beyond_the_range_of_exp_matches_the_cpu_loop(hip, oracle, ase_small, seed_small):
    """|gvl * gv| > 708 (Helper.h:549-557, :575-579): e^gl overflows on the CPU -- +inf, or 0 * inf = NaN and
    error -3.  The fast form of the update is only taken below that range; beyond it the CPU's own formula
    runs, so failure code, failed-ray handling and the image equal the CPU loop's, in both modes."""
    def boosted(p, f):
        q = copy.copy(p)
        q.gain = [rt.Gain(g.x, g.y, g.n, g.g0 * np.float32(f), g.E0, g.gv, g.Nv) for g in p.gain]
        return q

    ids = np.arange(0, ase_small.n_rays_total, 499, dtype=np.int64)
    rays = ase_small.build_rays(ids)
    hit = 0
    for f in (40.0, 700.0, 1e30):
        p = boosted(ase_small, f)
        out, ref = run_hip(hip, p, rays), oracle.image_loop(p, rays)
        assert out["failure_code"] == ref["failure_code"], f
        assert out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
        same_outputs_in_a_failing_run(out, ref)
        hit += ref["failure_code"] != 0 or not np.isfinite(ref["image"]).all()
    assert hit >= 2, "the test must reach the overflow regime"
    ids = np.arange(1, seed_small.n_rays_total, 4999, dtype=np.int64)
    rays = seed_small.build_rays(ids)
    hit = 0
    for f in (40.0, 700.0):
        p = boosted(seed_small, f)
        out, ref = run_hip(hip, p, rays), oracle.image_loop(p, rays)
        assert out["failure_code"] == ref["failure_code"], f
        same_outputs_in_a_failing_run(out, ref)
        hit += ref["failure_code"] != 0 or not np.isfinite(ref["image"]).all()
    assert hit >= 1


def test_wide_launch_angles_in_list_mode(hip, oracle, ase_small):
    """List-mode launch angles of 200 ... 800 mrad (Helper.h:409-411: tanf(1e-3f * a)): beyond the range of
    the restated float kernel the device used the float-rounded f64 tangent, which is not always tanf.  The
    march record must equal the CPU loop's whatever the angle."""
    ids = np.arange(0, ase_small.n_rays_total, 211, dtype=np.int64)
    rays = ase_small.build_rays(ids)
    rng = np.random.default_rng(5)
    rays["a"] = rng.uniform(-800.0, 800.0, len(rays)).astype(np.float32)
    rays["b"] = rng.uniform(-800.0, 800.0, len(rays)).astype(np.float32)
    rays["a"][::3] = np.sign(rays["a"][::3]) * rng.uniform(200.0, 800.0, len(rays[::3])).astype(np.float32)
    out = run_hip(hip, ase_small, rays, probe=True)
    ora = oracle.probe(ase_small, rays, want_Iv=False)
    same_record(out["probe"], ora)
    ref = oracle.image_loop(ase_small, rays)
    assert out["failure_code"] == ref["failure_code"]
    same_outputs_in_a_failing_run(out, ref, tol=1e-5)


def test_ieee_division_variant_of_the_march_gives_the_same_records(hip, oracle, ase_small, monkeypatch):
    """The march has two instances (rt_march.hip, template BOUNDED): short division sequences under table ranges
    that rt_hip_plan_create verifies, full IEEE divisions otherwise.  RT_HIP_MARCH_IEEE=1 forces the second on
    the shipped tables: every march record must come out bit-identical to the default run and to the oracle."""
    ids = np.arange(0, ase_small.n_rays_total, 7, dtype=np.int64)
    rays = ase_small.build_rays(ids)
    fast = run_hip(hip, ase_small, rays, probe=True)
    monkeypatch.setenv("RT_HIP_MARCH_IEEE", "1")
    slow = run_hip(hip, ase_small, rays, probe=True)
    monkeypatch.delenv("RT_HIP_MARCH_IEEE")
    ora = oracle.probe(ase_small, rays, want_Iv=False)
    same_record(fast["probe"], ora)
    same_record(slow["probe"], ora)
    assert np.array_equal(fast["probe"]["ray2"].view(np.uint32), slow["probe"]["ray2"].view(np.uint32))
    assert rel_l2(fast["image"], slow["image"]) < 1e-14


def test_tables_outside_the_verified_ranges_take_the_ieee_march(hip, oracle, ase_small):
    """A refractive index far from 1 (here n * 6: outside [0.25, 4]) fails the range check of rt_hip_plan_create
    (tables_bounded), so the run takes the IEEE-division instance of the march by itself; records against the oracle."""
    p = copy.deepcopy(ase_small)
    for g in p.gain[1:]:
        g.n = g.n * 6.0
    ids = np.arange(0, p.n_rays_total, 23, dtype=np.int64)
    rays = p.build_rays(ids)
    out = run_hip(hip, p, rays, probe=True)
    ora = oracle.probe(p, rays, want_Iv=False)
    same_record(out["probe"], ora)
    ref = oracle.image_loop(p, rays)
    assert out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
    assert rel_l2(out["image"], ref["image"]) < TIGHT


@pytest.mark.parametrize("case", ["wide_angles", "long_rows", "angles_beyond_lds", "seeded_long_rows"])
def test_lds_layout_extremes_of_the_frequency_kernel(hip, oracle, ase_small, seed_small, case):
    """The frequency kernel's work-group keeps its tables, the I_ang histogram and every wave's scratch in one
    dynamic LDS block (rt_freq.hip: freq_lds_doubles); the corners of that layout against the oracle:
    a 64 x 64 I_ang histogram (the 32 KB that still fit), a frequency axis whose rows leave no room for a row
    cache, an I_ang too large for LDS (global atomics), and the seeded pass with long rows (segmented scan)."""
    if case == "wide_angles":
        p = problem_mod.regrid_beam(ase_small, nx=3, ny=2, na=64, nb=64)
    elif case == "long_rows":
        p = problem_mod.regrid_beam(problem_mod.resample_frequency(ase_small, 700), nx=4, ny=3, na=9, nb=7)
    elif case == "angles_beyond_lds":
        p = problem_mod.regrid_beam(ase_small, nx=2, ny=2, na=80, nb=70)
    else:
        p = problem_mod.regrid_seed_beam(problem_mod.resample_frequency(seed_small, 700), nx=6, ny=3, na=20, nb=20)
    rays = p.build_rays()
    ref = oracle.image_loop(p, rays, n_threads=8)
    with hip.Plan(p) as plan:
        out = plan.set_ray_grid().run().fetch()
    assert out["failure_code"] == ref["failure_code"] == 0
    tol = 1e-10 if p.seed is not None else TIGHT
    assert rel_l2(out["image"], ref["image"]) < tol and rel_l2(out["I_ang"], ref["I_ang"]) < tol
    lst = run_hip(hip, p, rays)      # the same through the ray list (no own cells, no grid tables)
    assert rel_l2(lst["image"], ref["image"]) < tol and rel_l2(lst["I_ang"], ref["I_ang"]) < tol

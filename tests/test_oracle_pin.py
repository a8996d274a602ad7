"""Pins the CPU oracle (oracle/rt_oracle.c) to the reference.

1. bit-identical to the committed outputs of the UNMODIFIED reference `cpu`
   method (tests/golden/*_ref_cpu.npz, made by tests/golden/make_golden.py);
2. within the reference harness' tolerance of the golden image embedded in the
   reference's own .dat files (src/CreateImageHelpers.cpp:66-100);
3. per-ray: RayTrace::calc_ray outputs of 400 strided rays per file, bit for bit;
4. when the compiled reference travelled with the repo (oracle/_ref/), live
   bitwise comparison on a slice of rays.
"""
import importlib

import numpy as np
import pytest

from conftest import GOLDEN, rel_l2

rt = importlib.import_module("raytrace-miniapp_amd")


def test_ase_small_bitwise_vs_reference_outputs(oracle, ase_small, ase_ref):
    out = oracle.image_loop(ase_small)
    assert out["failure_code"] == 0
    assert out["counters"]["n_rays"] == 399000
    assert out["counters"]["cell_steps"] == 4768067
    assert out["counters"]["cross_iters"] == 9569995
    assert out["counters"]["inner_iters"] == 13889342
    assert np.array_equal(out["image"], ase_ref["image"])
    assert np.array_equal(out["I_ang"], ase_ref["I_ang"])
    # the file's embedded golden image pins the image to ~5e-7 (one-sided norm gate 5e-6)
    assert rel_l2(out["image"], ase_small.golden_image) < 5.2e-7
    n0, n1 = np.linalg.norm(ase_small.golden_image), np.linalg.norm(out["image"])
    assert (n0 - n1) / n0 <= 5e-6
    # golden I_ang follows the GPU variants' mirrored range rule:
    # equal on the cells that survive it, zero elsewhere
    g = ase_small.golden_I_ang
    nz = g != 0
    assert nz.sum() == 85
    assert rel_l2(out["I_ang"][nz], g[nz]) < 5e-7


def test_seed_small_bitwise_vs_reference_outputs(oracle, seed_small, seed_ref):
    out = oracle.image_loop(seed_small)        # serial: the summation order matters for bit equality
    assert out["failure_code"] == 0
    assert out["counters"]["n_rays"] == 7803000
    assert out["counters"]["cell_steps"] == 53573880
    assert np.array_equal(out["image"], seed_ref["image"])
    assert np.array_equal(out["I_ang"], seed_ref["I_ang"])
    assert rel_l2(out["image"], seed_small.golden_image) < 5.2e-7
    g = seed_small.golden_I_ang
    nz = g != 0
    assert nz.sum() == 55
    assert rel_l2(out["I_ang"][nz], g[nz]) < 5e-7


@pytest.mark.parametrize("name", ["ASE_small", "seed_small"])
def test_per_ray_calc_ray_bitwise(oracle, name):
    p = rt.datfile.load(GOLDEN / f"{name}.dat.xz")
    fx = np.load(GOLDEN / f"{name}_ref_rays.npz")
    stride, n = int(fx["stride"]), fx["Iv"].shape[0]
    ids = np.arange(n, dtype=np.int64) * stride
    rays = p.build_rays(ids)
    # the reference built these rays itself: same coordinates
    for q, key in enumerate("xyab"):
        assert np.array_equal(rays[key], fx["rays"][:, q].astype(np.float32))
    pr = oracle.probe(p, rays)
    assert np.array_equal(pr["err"], fx["err"])
    ok = fx["err"] == 0
    assert np.array_equal(pr["Iv"][ok], fx["Iv"][ok])
    for q, key in enumerate("xyab"):
        assert np.array_equal(pr["ray2"][key][ok].astype(np.float64), fx["ray2"][ok, q])


def test_threads_equal_serial_to_rounding(oracle, ase_small, ase_ref):
    out = oracle.image_loop(ase_small, n_threads=4)
    assert rel_l2(out["image"], ase_ref["image"]) < 1e-14
    assert rel_l2(out["I_ang"], ase_ref["I_ang"]) < 1e-13
    assert out["counters"]["cell_steps"] == 4768067


def test_live_reference_slice(oracle, ase_small):
    from oracle.binding import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/librt_ref.so not built on this box")
    ref = Reference()
    ids = np.arange(100000, 140000, dtype=np.int64)
    rays = ase_small.build_rays(ids)
    a = oracle.image_loop(ase_small, rays)
    b = ref.cpu_loop(ase_small, rays)
    assert np.array_equal(a["image"], b["image"])
    assert np.array_equal(a["I_ang"], b["I_ang"])


def test_frequency_slicing_property(oracle, ase_small):
    """Frequencies are independent given the march record (SURVEY.md 8(c) iii):
    tracing with a slice of the frequency axis reproduces that slice of the image,
    and I_ang is the sum over slices."""
    import copy
    ids = np.arange(0, ase_small.n_rays_total, 53, dtype=np.int64)
    rays = ase_small.build_rays(ids)
    full = oracle.image_loop(ase_small, rays)
    K = ase_small.beam.nv
    img = np.zeros_like(full["image"]).reshape(-1, K)
    ang = np.zeros_like(full["I_ang"])
    for lo, hi in ((0, 20), (20, 52)):
        q = copy.copy(ase_small)
        q.beam = copy.copy(ase_small.beam)
        q.beam.dv = np.ascontiguousarray(ase_small.beam.dv[lo:hi])
        q.gain = [rt.Gain(g.x, g.y, g.n, g.g0, g.E0, g.gv.reshape(-1, K)[:, lo:hi].copy(), hi - lo)
                  for g in ase_small.gain]
        part = oracle.image_loop(q, rays)
        img[:, lo:hi] = part["image"].reshape(-1, hi - lo)
        ang += part["I_ang"]
    assert np.array_equal(img.reshape(-1), full["image"])
    assert rel_l2(ang, full["I_ang"]) < 1e-14


@pytest.mark.parametrize("name", ["ASE_small", "seed_small"])
def test_path_tracer_bitwise_vs_reference_calc_ray_path(oracle, name):
    """Oracle restatement of the `debug` path (Helper.h:419-426, 505-511, 536-566) against
    RayTrace::calc_ray_path outputs of the compiled reference (fixture), two step factors."""
    p = rt.datfile.load(GOLDEN / f"{name}.dat.xz")
    fx = np.load(GOLDEN / f"{name}_ref_path.npz")
    i0, n = fx["i0"], fx["n"]
    gx, gy, ga, gb = p.ray_grid
    I, J, K_, M = np.meshgrid(*[np.arange(c) + s for s, c in zip(i0, n)], indexing="ij")
    ids = ((I * len(gy) + J) * len(ga) + K_) * len(gb) + M
    rays = p.build_rays(ids.reshape(-1).astype(np.int64))
    for c in (0.5, 0.25):
        out = oracle.calc_ray_path(p, rays, c)
        N2 = out["x"].shape[1]
        for key in "xyI":
            mine = out[key].reshape(*n, N2).transpose(3, 2, 1, 0, 4)
            assert np.array_equal(mine.view(np.uint32), fx[f"{key}_c{c}"].view(np.uint32))
        assert int((out["err"] != 0).sum()) == int(fx[f"nerr_c{c}"])


# ---- the full-size configurations pinned to the reference itself (fixtures made by tests/golden/make_golden.py from
# ---- oracle/_ref/librt_ref.so: the reference's own scale_problem, and its CPU loop in frequency slices)
def config5_centre_tile(ase_small):
    """The centre 64 x 64-pixel tile of BASELINE config 5 (4096 x 4096 pixels, nv = 512, na = nb = 1) as a problem of its
    own: the full problem's cells and tables, the tile's pixels as deposit / ray grid."""
    import copy
    problem_mod = importlib.import_module("raytrace-miniapp_amd.problem")
    fx = np.load(GOLDEN / "config5_tile_ref.npz")
    n, T, K, i0, j0 = (int(fx[k]) for k in ("n", "T", "K", "i0", "j0"))
    p = problem_mod.regrid_beam(problem_mod.resample_frequency(ase_small, K), nx=n, ny=n, a_centre=-1.0, b_centre=-4.5)
    q = copy.copy(p)
    q.beam = copy.copy(p.beam)
    q.beam.x = np.ascontiguousarray(p.beam.x[i0:i0 + T])
    q.beam.y = np.ascontiguousarray(p.beam.y[j0:j0 + T])
    return q, fx


def frequency_slice(p, k0, k1):
    import copy
    K = p.beam.nv
    q = copy.copy(p)
    q.beam = copy.copy(p.beam)
    q.beam.dv = np.ascontiguousarray(p.beam.dv[k0:k1])
    q.gain = [rt.Gain(g.x, g.y, g.n, g.g0, g.E0, np.ascontiguousarray(g.gv.reshape(-1, K)[:, k0:k1]).reshape(-1), k1 - k0)
              for g in p.gain]
    return q


def test_config5_tile_bitwise_vs_reference_frequency_slices(oracle, ase_small):
    """BASELINE config 5 cannot run through the reference in one piece (nv = 512 >= K_MAX = 100,
    src/common/RayTraceImageHelper.h:30, src/RayTraceImage.cpp:231); SURVEY.md 8(d): frequencies are independent given a
    ray's march, so six slices of <= 96 through RayTraceImageCPULoop are the reference's answer.  The oracle's ONE pass over
    all 512 frequencies must give those very doubles (every 4th pixel's row is in the fixture, and every pixel's sum over k),
    and its own slices the reference's I_ang of each slice, bit for bit."""
    q, fx = config5_centre_tile(ase_small)
    T, K = int(fx["T"]), int(fx["K"])
    out = oracle.image_loop(q)
    assert out["failure_code"] == 0
    img = out["image"].reshape(T, T, K)
    assert np.array_equal(img[::4, ::4, :], fx["rows"])
    assert np.array_equal(img.sum(axis=2), fx["row_sums"])
    assert np.linalg.norm(fx["rows"]) > 0 and (fx["row_sums"] > 0).mean() > 0.5
    # I_ang: the reference adds one partial sum per slice and ray, the single pass one sum over all k per ray
    assert abs(out["I_ang"][0] - fx["I_ang"][0]) <= 1e-14 * abs(fx["I_ang"][0])
    for (k0, k1), want in zip(fx["slices"], fx["I_ang_slices"]):
        part = oracle.image_loop(frequency_slice(q, int(k0), int(k1)))
        assert np.array_equal(part["I_ang"], want)
        assert np.array_equal(part["image"].reshape(T, T, -1)[::4, ::4, :], fx["rows"][:, :, int(k0):int(k1)])


@pytest.mark.parametrize("name", ["ASE_small", "seed_small"])
def test_scale_problem_equals_the_references_own(name):
    """BASELINE config 3 is built with scale_problem(info, 16) (src/CreateImageHelpers.cpp:104-150): the Python mirror
    must leave the very grids the reference's function leaves -- sizes, spacings and every coordinate, bit for bit."""
    fx = np.load(GOLDEN / "scale16_ref_grids.npz")
    p = rt.scale_problem(rt.datfile.load(GOLDEN / f"{name}.dat.xz"), 16.0)
    beams = [("beam", p.beam)] + ([("seed_beam", p.seed_beam)] if p.seed_beam is not None else [])
    assert (f"{name}.seed_beam.x" in fx.files) == (p.seed_beam is not None)
    for key, b in beams:
        assert list(fx[f"{name}.{key}.n"]) == [len(b.x), len(b.y), len(b.a), len(b.b)]
        assert np.array_equal(fx[f"{name}.{key}.d"], np.array([b.dx, b.dy, b.da, b.db]))
        for ax in "xyab":
            assert np.array_equal(fx[f"{name}.{key}.{ax}"], getattr(b, ax)), (key, ax)
    if name == "ASE_small":
        assert p.n_rays_total == 6384000


@pytest.mark.skipif(not importlib.import_module("oracle.binding").Reference.available(), reason="oracle/_ref not built")
def test_live_reference_slices_equal_the_fixture(ase_small):
    """Where the compiled reference travelled with the repo: the fixture is what it computes now."""
    from oracle.binding import Reference
    q, fx = config5_centre_tile(ase_small)
    r = Reference().cpu_loop_sliced(q, max_nv=96)
    T, K = int(fx["T"]), int(fx["K"])
    assert np.array_equal(r["image"].reshape(T, T, K)[::4, ::4, :], fx["rows"])
    assert np.array_equal(r["I_ang_slices"], fx["I_ang_slices"])

"""Full-size workloads of BASELINE.json on the GPU: the ASE_medium stand-in
(config 3) against the multi-threaded oracle, and a tile of the synthetic
4096x4096x512 workload (config 5) against the oracle, plus size-independent
properties at full size."""
import importlib
import os

import numpy as np
import pytest

from conftest import rel_l2

rt = importlib.import_module("raytrace-miniapp_amd")
problem_mod = importlib.import_module("raytrace-miniapp_amd.problem")
pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_ase_medium_standin_vs_oracle(hip, oracle, ase_small):
    p = rt.scale_problem(ase_small, 16.0)
    assert p.n_rays_total == 6384000
    with hip.Plan(p) as plan:
        out = plan.set_ray_grid().run().fetch()
    ref = oracle.image_loop(p, n_threads=min(16, os.cpu_count() or 1))
    assert out["failure_code"] == 0 and ref["failure_code"] == 0
    assert out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
    assert rel_l2(out["image"], ref["image"]) < TOL and rel_l2(out["I_ang"], ref["I_ang"]) < TOL
    # ASE property (SURVEY.md 8(c) i): ray ijkm lands in angle cell (k, m), so every
    # I_ang cell is a sum over all pixels and sum(I_ang) = sum_pixels sum_k 2 dv_k image
    K = p.beam.nv
    lhs = out["I_ang"].sum()
    rhs = (out["image"].reshape(-1, K) * (2.0 * p.beam.dv)[None, :]).sum()
    assert abs(lhs - rhs) <= 1e-10 * abs(rhs)


def test_seed_medium_like_subsample_vs_oracle(hip, oracle, seed_small):
    p = rt.scale_problem(seed_small, 2.0)
    with hip.Plan(p) as plan:
        out = plan.set_ray_grid().run().fetch()
    ref = oracle.image_loop(p, n_threads=min(16, os.cpu_count() or 1))
    assert out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
    assert rel_l2(out["image"], ref["image"]) < TOL and rel_l2(out["I_ang"], ref["I_ang"]) < TOL


def test_synthetic_config5_tile_vs_oracle(hip, oracle, ase_small):
    """Config 5 shape (nv = 512, na = nb = 1, dense pixel grid), 96 x 64 pixel tile."""
    p = problem_mod.resample_frequency(ase_small, 512)
    p = problem_mod.regrid_beam(p, nx=96, ny=64, a_centre=-1.0, b_centre=-4.5)
    assert p.n_rays_total == 96 * 64
    with hip.Plan(p) as plan:
        out = plan.set_ray_grid().run().fetch()
    ref = oracle.image_loop(p, n_threads=8)
    assert out["failure_code"] == 0
    assert out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
    assert rel_l2(out["image"], ref["image"]) < TOL and rel_l2(out["I_ang"], ref["I_ang"]) < TOL
    assert np.linalg.norm(ref["image"]) > 0


def test_exclusive_pixel_mode_equals_atomic_mode(hip, oracle, ase_small):
    """na = nb = 1 on the beam's own grid: one ray per pixel, rows are stored, not added.
    Same image as the oracle, and as the list path (which never uses the store mode)."""
    p = problem_mod.resample_frequency(ase_small, 128)
    p = problem_mod.regrid_beam(p, nx=70, ny=33, a_centre=-1.0, b_centre=-4.5)
    with hip.Plan(p) as plan:
        a = plan.set_ray_grid().run().fetch()
        a2 = plan.run().fetch()                      # re-run: rows are rewritten, not accumulated
    with hip.Plan(p) as plan:
        b = plan.set_rays(p.build_rays()).run().fetch()
    ref = oracle.image_loop(p, n_threads=4)
    assert np.array_equal(a["image"], a2["image"])
    assert rel_l2(a["image"], ref["image"]) < 2e-7 and rel_l2(b["image"], ref["image"]) < 2e-7
    assert rel_l2(a["I_ang"], ref["I_ang"]) < 2e-7
    assert (a["image"] == 0).reshape(-1, 128).all(axis=1).sum() == (ref["image"] == 0).reshape(-1, 128).all(axis=1).sum()


def test_seed_medium_standin_full_size_vs_oracle(hip, oracle, seed_small):
    """BASELINE config 3, seeded half: seed_small x scale_problem(16) = 124,848,000 rays (the stand-in
    for the absent seed_medium.dat) against the 16-thread oracle: same ray-step count, rel-L2 <= 1e-5."""
    p = rt.scale_problem(seed_small, 16.0)
    assert p.n_rays_total == 124848000
    with hip.Plan(p) as plan:
        out = plan.set_ray_grid().run().fetch()
    ref = oracle.image_loop(p, n_threads=min(16, os.cpu_count() or 1))
    assert out["failure_code"] == 0 and ref["failure_code"] == 0
    assert out["stats"]["n_rays"] == 124848000
    assert out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
    assert rel_l2(out["image"], ref["image"]) < TOL and rel_l2(out["I_ang"], ref["I_ang"]) < TOL
    assert np.linalg.norm(ref["image"]) > 0


def test_config5_full_size_tiles_vs_oracle(hip, oracle, ase_small):
    """BASELINE config 5 at FULL size (4096 x 4096 pixels x 512 frequencies, 16,777,216 rays, 68.7 GB image
    that stays on the device): four corner tiles and the centre tile (64 x 64 pixels) are copied out and
    compared with the oracle on the matching sub-problem; the same tile traced as its own small plan must
    give the same rows bit for bit and the oracle's ray-step count; sum(I_ang) = sum 2 dv image holds on
    the device over the whole image."""
    import torch

    n, T, K = 4096, 64, 512
    p = problem_mod.regrid_beam(problem_mod.resample_frequency(ase_small, K), nx=n, ny=n, a_centre=-1.0, b_centre=-4.5)
    assert p.n_rays_total == n * n
    dev = torch.device("cuda", 0)
    image = torch.empty(n * n * K, dtype=torch.float64, device=dev)       # 68.7 GB, written once
    iang = torch.zeros(1, dtype=torch.float64, device=dev)
    with hip.Plan(p) as plan:
        plan.set_ray_grid().run(torch.cuda.current_stream().cuda_stream, image.data_ptr(), iang.data_ptr())
        st = plan.fetch(want_image=False)
    assert st["failure_code"] == 0 and st["stats"]["n_rays"] == n * n
    img = image.view(n, n, K)                                              # [iy][ix][k]
    dv2 = torch.tensor(2.0 * p.beam.dv, dtype=torch.float64, device=dev)
    lhs = float(iang.sum().item())
    rhs = float((image.view(-1, K) * dv2[None, :]).sum().item())
    assert abs(lhs - rhs) <= 1e-10 * abs(rhs) and rhs > 0
    steps_tiles = 0
    for (i0, j0) in [(0, 0), (n - T, 0), (0, n - T), (n - T, n - T), (n // 2 - T // 2, n // 2 - T // 2)]:
        tile = img[j0:j0 + T, i0:i0 + T, :].contiguous().cpu().numpy().reshape(-1)
        q = copy_beam_window(p, i0, j0, T)
        ref = oracle.image_loop(q, n_threads=8)
        with hip.Plan(q) as plan:
            small = plan.set_ray_grid().run().fetch()
        assert small["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
        assert np.array_equal(small["image"], tile), "full-size rows differ from the same tile traced alone"
        assert rel_l2(tile, ref["image"]) < TOL
        steps_tiles += ref["counters"]["cell_steps"]
    assert steps_tiles > 0
    del image, img


def test_config5_centre_tile_vs_reference_frequency_slices(hip, ase_small):
    """The centre 64 x 64 tile of config 5 against the REFERENCE ITSELF: RayTraceImageCPULoop cannot take nv = 512 in one
    piece (K_MAX = 100, src/common/RayTraceImageHelper.h:30), so the fixture holds its answer from six frequency slices of
    <= 96 (tests/golden/make_golden.py): every 4th pixel's row, every pixel's sum over k, I_ang."""
    from test_oracle_pin import config5_centre_tile
    q, fx = config5_centre_tile(ase_small)
    T, K = int(fx["T"]), int(fx["K"])
    with hip.Plan(q) as plan:
        out = plan.set_ray_grid().run().fetch()
    assert out["failure_code"] == 0
    img = out["image"].reshape(T, T, K)
    assert rel_l2(img[::4, ::4, :], fx["rows"]) < TOL
    assert rel_l2(img.sum(axis=2), fx["row_sums"]) < TOL
    assert rel_l2(out["I_ang"], fx["I_ang"]) < TOL
    assert np.linalg.norm(fx["rows"]) > 0


def copy_beam_window(p, i0, j0, T):
    """The sub-problem whose deposit / ray grid is the T x T pixel window at (i0, j0) of p's beam."""
    import copy
    q = copy.copy(p)
    b = copy.copy(p.beam)
    b.x = np.ascontiguousarray(p.beam.x[i0:i0 + T])
    b.y = np.ascontiguousarray(p.beam.y[j0:j0 + T])
    q.beam = b
    return q


def test_bench_two_rank_seeded_rehearsal(hip):
    """The seeded form of the N-rank bench (source columns sharded, ONE sum-reduce of image | I_ang): 124.8 M
    rays split over two ranks that share this box's GPU, collective staged through gloo."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "seed_medium", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["config"]["rays_total"] == 124848000
    assert line["config"]["parallelism"] == "source-columns x2"
    assert "reduce" in line["multi_gpu"]["collective"]


def test_bench_two_rank_launch_rehearsal(hip):
    """`python bench.py --gpus 2` must start its ranks itself (the driver launches it that way too).  This
    box has one GPU, so the rehearsal puts both ranks on device 0 and stages the collective through gloo;
    what is checked is the launcher, the strong-scaling shard, the one-collective assembly and the JSON line.
    (RCCL itself needs one device per rank: unmeasured here.)"""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["multi_gpu"]["ranks_seen"] == 2
    assert line["config"]["rays_total"] == 6384000 and line["config"]["ray_steps_total"] == 75601675
    assert line["config"]["rays_per_gpu"] == 3192000
    # the product's own multi-GPU arm on the same workload (rt_hip_multi_image_loop, ndev = 2; on this one-GPU box the
    # loop-back rehearsal): present, in tile mode, and equal to the single-device image in the run itself
    cab = line["multi_gpu"]["cabi"]
    assert "error" not in cab, cab
    assert cab["entry"] == "rt_hip_multi_image_loop" and cab["ndev"] == 2 and "tiles" in cab["mode"]
    assert cab["image_matches_single_device_1e-12"] and cab["ray_steps"] == 75601675 and cab["failure_code"] == 0
    assert cab["ms_per_image"] > 0 and cab["kernel_ms_max_over_devices"] > 0
    # no fraction above 1 anywhere in the line (contract_frac is the formula's figure and may exceed it)
    from test_host_logic import _fracs
    for path, v in _fracs(line):
        assert v is None or v <= 1.0, (path, v)
    assert line["warmup"] == 1 and line["warmup_requested"] == 1

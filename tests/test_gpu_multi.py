"""The multi-device entry of the C ABI (rt_hip_multi_image_loop: RCCL communicator, pixel-column tiles +
gather for ASE, ray chunks + sum-reduce otherwise) and the ray-grid recognition of the host-pointer entry.
The GPU box has ONE device: the communicator is degenerate (self send / self reduce), which still runs
every line of the multi-device code path; N > 1 is unmeasured on hardware here."""
import importlib
import os

import numpy as np
import pytest

from conftest import rel_l2

rt = importlib.import_module("raytrace-miniapp_amd")
pytestmark = pytest.mark.gpu


def test_multi_loop_ase_tiles_equal_the_single_device_image(hip, oracle, ase_small):
    one = hip.image_loop(ase_small)
    out = hip.multi_image_loop(ase_small, n_devices=1)
    assert out["mode"] == 1, "ASE with the beam's own ray grid must take the pixel-tile path"
    assert out["failure_code"] == 0
    assert out["stats"]["cell_steps"] == one["stats"]["cell_steps"] == 4768067
    ref = oracle.image_loop(ase_small, n_threads=8)
    assert rel_l2(out["image"], ref["image"]) < 1e-6 and rel_l2(out["I_ang"], ref["I_ang"]) < 1e-6
    assert rel_l2(out["image"], one["image"]) < 1e-12 and rel_l2(out["I_ang"], one["I_ang"]) < 1e-12


def test_multi_loop_seeded_takes_the_sum_reduce_path(hip, oracle, seed_small):
    p = rt.scale_problem(seed_small, 0.05)
    out = hip.multi_image_loop(p, n_devices=1)
    assert out["mode"] == 2
    ref = oracle.image_loop(p, n_threads=8)
    assert out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
    assert rel_l2(out["image"], ref["image"]) < 1e-9 and rel_l2(out["I_ang"], ref["I_ang"]) < 1e-9


def test_multi_loop_arbitrary_list_falls_back_to_chunks(hip, oracle, ase_small):
    rays = ase_small.build_rays()[5:-11:3].copy()          # not a tensor grid
    out = hip.multi_image_loop(ase_small, rays, n_devices=1)
    assert out["mode"] == 2
    ref = oracle.image_loop(ase_small, rays, n_threads=8)
    assert out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
    assert rel_l2(out["image"], ref["image"]) < 1e-6 and rel_l2(out["I_ang"], ref["I_ang"]) < 1e-6


def test_image_loop_recognises_the_grid_and_survives_a_list_that_only_looks_like_one(hip, oracle, ase_small, monkeypatch):
    rays = ase_small.build_rays()
    assert hip.ray_list_grid_dims(rays) == (ase_small.beam.nx, ase_small.beam.ny, ase_small.beam.na, ase_small.beam.nb)
    a = hip.image_loop(ase_small, rays)                    # rays generated on the device, list verified beside the run
    monkeypatch.setenv("RT_HIP_NO_GRID_DETECT", "1")
    b = hip.image_loop(ase_small, rays)                    # list uploaded
    monkeypatch.delenv("RT_HIP_NO_GRID_DETECT")
    assert a["stats"]["cell_steps"] == b["stats"]["cell_steps"]
    assert rel_l2(a["image"], b["image"]) < 1e-12 and rel_l2(a["I_ang"], b["I_ang"]) < 1e-12
    # one ray moved: the periods still say "grid", the ray-by-ray check says no, the list itself is traced
    odd = rays.copy()
    odd["x"][123457] = odd["x"][0]
    assert hip.ray_list_grid_dims(odd) is None
    c = hip.image_loop(ase_small, odd)
    ref = oracle.image_loop(ase_small, odd, n_threads=8)
    assert c["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
    assert rel_l2(c["image"], ref["image"]) < 1e-6 and rel_l2(c["I_ang"], ref["I_ang"]) < 1e-6
    assert rel_l2(c["image"], a["image"]) > 0
    # the all-devices entry speculates likewise (devices start tracing the guessed grid while the list is verified, the
    # verdict arrives before the collective): the look-alike list ends the first attempt and is traced as ray chunks,
    # on the degenerate communicator and on the three-worker rehearsal
    for loop in ("", "3"):
        if loop:
            monkeypatch.setenv("RT_HIP_MULTI_LOOPBACK", loop)
        d = hip.multi_image_loop(ase_small, odd)
        assert d["mode"] == 2 and d["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
        assert rel_l2(d["image"], ref["image"]) < 1e-6 and rel_l2(d["I_ang"], ref["I_ang"]) < 1e-6
        e = hip.multi_image_loop(ase_small, rays)
        assert e["mode"] == 1 and rel_l2(e["image"], a["image"]) < 1e-12
    monkeypatch.delenv("RT_HIP_MULTI_LOOPBACK")


def test_pool_trim_and_concurrent_image_loops(hip, ase_small):
    """Two host threads call the host-pointer entry on one device at once (create_image is documented
    thread-safe, RayTrace.h:90-91): each call leases its own queue; results are the single-call results."""
    import threading
    want = hip.image_loop(ase_small)
    got = [None, None]

    def run(i):
        got[i] = hip.image_loop(ase_small)

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g in got:
        assert g["stats"]["cell_steps"] == want["stats"]["cell_steps"]
        assert rel_l2(g["image"], want["image"]) < 1e-12
    hip.HipLibrary.get().lib.rt_hip_pool_trim()
    again = hip.image_loop(ase_small)
    assert rel_l2(again["image"], want["image"]) < 1e-12


def test_timing_ring_keeps_the_last_runs(hip, ase_small):
    """rt_hip_plan_set_timing_ring: back-to-back runs are timed without waiting for each (bench.py)."""
    with hip.Plan(ase_small) as plan:
        plan.set_ray_grid().set_timing_ring(3)
        assert plan.ring_times() == []
        for _ in range(5):
            plan.run()
        t = plan.ring_times()
        assert len(t) == 3 and all(0.0 < m < 50.0 and 0.0 < f < 50.0 for m, f in t)
        last = plan.kernel_times()
        assert abs(last[0] - t[-1][0]) < 1e-6 and abs(last[1] - t[-1][1]) < 1e-6
        out = plan.fetch()
        assert out["failure_code"] == 0 and out["stats"]["cell_steps"] == 4768067


def test_host_libm_probe_and_host_tangent_fallback(hip, oracle, ase_small):
    """The device restatement of tanf is probed against THIS host's tanf once per process
    (rt_hip_host_libm_mode); on the reference platform (glibc 2.35) they agree.  The fallback -- list-mode
    tangents from the host's tanf -- must give the same march records (run in a fresh process with
    RT_HIP_TAN_ON_HOST=1, since the mode is decided once)."""
    import subprocess
    import sys
    from pathlib import Path
    lib = hip.HipLibrary.get().lib
    assert lib.rt_hip_host_libm_mode(0) == 1
    root = Path(__file__).resolve().parents[1]
    code = (
        "import importlib, sys, numpy as np\n"
        f"sys.path.insert(0, {str(root)!r})\n"
        "rt = importlib.import_module('raytrace-miniapp_amd'); be = importlib.import_module('raytrace-miniapp_amd.backend')\n"
        "from oracle.binding import Oracle\n"
        f"p = rt.datfile.load({str(root / 'tests' / 'golden' / 'ASE_small.dat.xz')!r})\n"
        "rays = p.build_rays(np.arange(0, p.n_rays_total, 97, dtype=np.int64))\n"
        "rays['a'] *= np.float32(7.3)\n"
        "assert be.HipLibrary.get().lib.rt_hip_host_libm_mode(0) == 2\n"
        "with be.Plan(p) as plan:\n"
        "    plan.set_rays(rays).enable_probe().run(); out = plan.fetch(); pr = plan.fetch_probe()\n"
        "o = Oracle().probe(p, rays, want_Iv=False)\n"
        "for k in ('gvl', 'evl'):\n"
        "    assert np.array_equal(pr[k].view(np.uint32), o[k].view(np.uint32))\n"
        "assert np.array_equal(pr['ivl'], o['ivl']) and np.array_equal(pr['steps'], o['steps'])\n"
        "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RT_HIP_TAN_ON_HOST="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("ndev", [2, 3, 5, 8])
def test_multi_loop_partition_and_assembly_for_several_devices(hip, oracle, ase_small, seed_small, monkeypatch, ndev):
    """The N > 1 logic of rt_hip_multi_image_loop -- pixel-column tiles (equal and unequal widths), tile
    buffers, the interleave kernel and the I_ang sum; ray chunks and the summed images -- rehearsed on the one
    GPU of this box: RT_HIP_MULTI_LOOPBACK runs n workers with n plans on device 0 and replaces only the RCCL
    collective by device-to-device copies into the same receive layout."""
    monkeypatch.setenv("RT_HIP_MULTI_LOOPBACK", str(ndev))
    p = rt.scale_problem(ase_small, 0.3)                       # nx = 44: unequal tiles for 3, 5, 8 devices
    assert p.beam.nx % 3 != 0
    one = hip.image_loop(p)
    out = hip.multi_image_loop(p)
    assert out["mode"] == 1 and out["failure_code"] == 0
    assert out["stats"]["n_rays"] == p.n_rays_total and out["stats"]["cell_steps"] == one["stats"]["cell_steps"]
    ref = oracle.image_loop(p, n_threads=8)
    assert rel_l2(out["image"], ref["image"]) < 1e-6 and rel_l2(out["I_ang"], ref["I_ang"]) < 1e-6
    assert rel_l2(out["image"], one["image"]) < 1e-12 and rel_l2(out["I_ang"], one["I_ang"]) < 1e-12
    # seeded: ray chunks of the seed-beam grid, full images summed
    q = rt.scale_problem(seed_small, 0.02)
    one = hip.image_loop(q)
    out = hip.multi_image_loop(q)
    assert out["mode"] == 2
    assert out["stats"]["n_rays"] == q.n_rays_total and out["stats"]["cell_steps"] == one["stats"]["cell_steps"]
    assert rel_l2(out["image"], one["image"]) < 1e-12 and rel_l2(out["I_ang"], one["I_ang"]) < 1e-12
    # an arbitrary list: chunks of the list itself
    rays = ase_small.build_rays()[7:-3:5].copy()
    ref = oracle.image_loop(ase_small, rays, n_threads=8)
    out = hip.multi_image_loop(ase_small, rays)
    assert out["mode"] == 2 and out["stats"]["cell_steps"] == ref["counters"]["cell_steps"]
    assert rel_l2(out["image"], ref["image"]) < 1e-6 and rel_l2(out["I_ang"], ref["I_ang"]) < 1e-6


_CHILD = r"""
import importlib, os, sys
sys.path.insert(0, {root!r})
rt = importlib.import_module("raytrace-miniapp_amd")
be = importlib.import_module("raytrace-miniapp_amd.backend")
p = rt.datfile.load({dat!r})
try:
    be.multi_image_loop(p, n_devices=1)
    print("RESULT ok")
except be.RayTraceError as exc:
    print("RESULT error:", exc)
os.environ.pop("RT_HIP_MULTI_INJECT_FAIL", None)
out = be.multi_image_loop(p, n_devices=1)          # the next call must work again (new communicator)
print("AFTER", out["failure_code"], out["stats"]["cell_steps"])
"""


@pytest.mark.parametrize("loopback", [0, 3])
def test_a_worker_failing_after_the_rendezvous_ends_the_call_with_an_error_not_a_hang(loopback):
    """rt_hip_multi_image_loop: a device that fails between the rendezvous and the collective (RCCL error, faulted
    queue) aborts every communicator (ncclCommAbort) so that no peer waits for it; the call returns RT_ERR_HIP,
    and the next call builds new communicators.  Run in a child process under a time limit: a hang is the failure
    this guards against.  loopback = 3: three workers on the one device (the collective replaced by copies),
    worker 1 fails; loopback = 0: the degenerate one-device RCCL communicator, its only worker fails and aborts it."""
    import subprocess
    import sys
    from conftest import GOLDEN, ROOT
    env = dict(os.environ)
    env["RT_HIP_MULTI_INJECT_FAIL"] = "1" if loopback else "0"
    if loopback:
        env["RT_HIP_MULTI_LOOPBACK"] = str(loopback)
    env["RT_HIP_MULTI_TIMEOUT_MS"] = "20000"
    code = _CHILD.format(root=str(ROOT), dat=str(GOLDEN / "ASE_small.dat.xz"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RESULT error:" in r.stdout and "injected failure after the rendezvous" in r.stdout, r.stdout
    assert "AFTER 0 4768067" in r.stdout, r.stdout

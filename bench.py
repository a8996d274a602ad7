#!/usr/bin/env python3
"""bench.py -- throughput of the ray-trace imaging hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak]
                  [--workload standin|seed_medium|config5] [--no-cpu-baseline] [--no-config5] [--no-seed-medium]

N > 1 may be started either way:
  * `python bench.py --gpus N ...`  -- this process spawns the N ranks itself (a
    `torch.distributed.run` child, started before anything here touches the GPU) and exits
    with the child's code;
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` -- one rank
    per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

A *step* is one pass of the hot path (zero outputs + march + frequency pass -- one launch, rt_fused.hip, where
it applies, else two -- [+ the RCCL assembly of the image tiles when N > 1]) over the workload, every input
resident in HBM.

Workloads (BASELINE.json `configs`):
  standin      configs[2]/[3], the metric's workload: ASE_medium.dat is absent from the reference
               checkout (.MISSING_LARGE_BLOBS), so this is the stand-in SURVEY.md 8(d) prescribes --
               ASE_small's plasma tables on the reference's own enlargement rule scale_problem(16):
               euv grid 120 x 50 x 38 x 28 = 6,384,000 rays, nv = 52, N = 3.
  seed_medium  the seeded analogue, seed_small x scale_problem(16) = 124,848,000 rays.
  config5      configs[4]: synthetic 4096 x 4096 pixels x 512 frequencies, na = nb = 1 (68.7 GB image).
N > 1: pixel columns (seeded: source columns) are dealt round-robin to the ranks.  `--scaling
strong` (default for N > 1, BASELINE config 4) splits the fixed workload; `--scaling weak` gives
every rank the whole single-GPU workload (N times as many columns).  Tiles are assembled on rank 0
with ONE collective per step: gather of (tile | I_ang) buffers (ASE) or sum-reduce (seeded).

Sub-records of the line (all beside `value`, never part of it): `roofline_small` = the two input files the
reference ships (BASELINE config 2), `roofline_config5`, `roofline_seed_medium`, `cpu_baseline` (the unmodified
reference CPU loop on every hardware thread of the box), `multi_gpu_cabi` = the product's own multi-GPU arm
(rt_hip_multi_image_loop with ndev = N, in a child process).

Rank 0 prints ONE JSON line; DESIGN.md section 5 describes every field.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 1024          # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
KERNEL_SOURCES = ("rt_device.h", "rt_freq.hip", "rt_fused.hip", "rt_march.hip", "rt_math.h", "rt_path.hip")


# --------------------------------------------------------------------------- self-launch
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the ranks as a torch.distributed.run
    child.  Nothing in this process has touched HIP (torch is not even imported yet), so the
    children are fresh processes, not re-execs of a GPU-initialised one."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())]
    cmd += sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# --------------------------------------------------------------------------- accounting
def algorithmic_bytes(n_rays: int, cell_steps: int, L: int, K: int, seeded: bool, n_live: int, n_pix: int,
                      n_ang: int) -> dict:
    """B_read = 16 R + C_step S + 4 K 3 L R [+ (256 + 8K) R_live], split between the
    two kernels of the path: the march gathers the plasma grid (16 R + C_step S), the frequency pass
    reads the lineshape rows (4 K 3 L R) and, seeded, the seed tables.  B_write = 8 (nx ny K + na nb)."""
    c_step = 80 if seeded else 96
    march = 16 * n_rays + c_step * cell_steps
    freq = 4 * K * 3 * L * n_rays + ((256 + 8 * K) * n_live if seeded else 0)
    return {"march": march, "freq": freq, "path": march + freq, "write": 8 * (n_pix * K + n_ang)}


def bounded_roof(alg_bytes: float, seconds: float, traffic: float | None = None, sq: dict | None = None,
                 clock_hz: float = 2.4e9, prefix: str = "") -> dict:
    """Roofline fields of a sub-record that can never show a fraction above 1.  `contract_frac` is the yardstick of
    SURVEY.md 8(d) -- ALGORITHMIC bytes of the contract formula over the time against the HBM peak -- which prices
    lineshape rows and grid nodes that the caches serve and therefore exceeds 1 on several workloads: it is a figure of
    merit, not a bound.  `frac` is a bound that can bind: HBM bytes measured by the counter passes over the time against the
    peak (null while no counter pass of these kernel sources is committed), and `issue_frac` the VALU + SALU
    wave-instructions per second against one per two cycles per SIMD, from the SQ pass of the same workload."""
    contract = alg_bytes / seconds / 1e9
    d = {prefix + "contract_achieved": contract, prefix + "contract_frac": contract / HBM_PEAK_GBS,
         prefix + "bound": "hbm", prefix + "peak": HBM_PEAK_GBS, prefix + "unit": "GB/s"}
    if traffic:
        d[prefix + "achieved"] = traffic / seconds / 1e9
        d[prefix + "frac"] = traffic / seconds / 1e9 / HBM_PEAK_GBS
        d[prefix + "achieved_basis"] = "measured HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE passes) over this run's time"
    elif contract / HBM_PEAK_GBS <= 1.0:
        d[prefix + "achieved"] = contract
        d[prefix + "frac"] = contract / HBM_PEAK_GBS
        d[prefix + "achieved_basis"] = "algorithmic bytes of the contract formula (no counter pass of these kernel sources)"
    else:
        d[prefix + "achieved"] = None
        d[prefix + "frac"] = None
        d[prefix + "achieved_basis"] = ("no counter pass of these kernel sources is committed, and the contract formula "
                                        "exceeds what HBM can deliver (cache-served rows): see contract_frac")
    c = (sq or {}).get("avg", {}) if sq else {}
    if c.get("SQ_INSTS_VALU"):
        instr = c["SQ_INSTS_VALU"] + c.get("SQ_INSTS_SALU", 0.0)
        peak = N_SIMD * clock_hz / 2.0
        d[prefix + "issue_frac"] = instr / seconds / peak
        d[prefix + "issue"] = {"bound": "valu+salu issue", "achieved": instr / seconds / 1e9, "peak": peak / 1e9, "unit": "G wave-instr/s",
                               "valu_instr_per_launch": c["SQ_INSTS_VALU"], "salu_instr_per_launch": c.get("SQ_INSTS_SALU")}
    return d


def kernel_source_hash() -> str:
    """sha1 over the kernel sources (the device code: KERNEL_SOURCES below, not the host translation units):
    profiles/summarize.py stamps the committed counter summary with it, and the counters are quoted only while the
    kernels are the ones they were measured on."""
    import hashlib

    h = hashlib.sha1()
    csrc = ROOT / "raytrace-miniapp_amd" / "csrc"
    for name in KERNEL_SOURCES:
        h.update(name.encode())
        h.update((csrc / name).read_bytes())
    return h.hexdigest()[:16]


def committed_counters() -> dict:
    """Counter passes are separate rocprofv3 runs (MI355X_MICROARCH.md); their summary is committed
    under profiles/ and quoted here with its source, the commit it was taken at and the hash of the kernel
    sources it was taken on.  Counters of other kernel sources are NOT quoted: {"_stale": ...}."""
    for name in ("r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc.json"):
        f = ROOT / "profiles" / name
        if f.exists():
            try:
                d = json.loads(f.read_text())
            except Exception:  # noqa: BLE001
                continue
            src = f"profiles/{name}"
            if d.get("git_head"):
                src += f" (measured at commit {d['git_head']})"
            if d.get("kernel_source_hash") != kernel_source_hash():
                return {"_stale": f"{src}: taken on other kernel sources (hash {d.get('kernel_source_hash')}, now "
                                  f"{kernel_source_hash()}); not quoted"}
            d["_source"] = src
            return d
    return {}


# --------------------------------------------------------------------------- workloads
def build_workload(rt, problem_mod, name: str, world: int, scaling: str):
    base = rt.datfile.load(ROOT / "tests" / "golden" / "ASE_small.dat.xz")
    if name == "standin":
        p = rt.scale_problem(base, 16.0)
        label = "ASE_medium stand-in: ASE_small tables x scale_problem(16)"
        data = ("ASE_small.dat plasma tables (reference input) on the synthetic scale_problem(16) ray grid; "
                "ASE_medium.dat itself is absent from the reference checkout")
    elif name == "seed_medium":
        p = rt.scale_problem(rt.datfile.load(ROOT / "tests" / "golden" / "seed_small.dat.xz"), 16.0)
        label = "seed_medium stand-in: seed_small tables x scale_problem(16)"
        data = "seed_small.dat tables (reference input) on the synthetic scale_problem(16) seed-beam grid"
    elif name == "config5":
        p = config5_problem(rt, problem_mod, 4096)
        label = "synthetic 4096x4096 pixels x 512 frequencies, na = nb = 1 (BASELINE config 5)"
        data = "ASE_small.dat plasma tables with the frequency axis resampled 52 -> 512, one ray per pixel"
    else:
        raise SystemExit(f"unknown workload {name}")
    if world > 1 and scaling == "weak":
        if p.seed is None:
            p = problem_mod.regrid_beam(p, nx=p.beam.nx * world)
        else:
            p = problem_mod.regrid_seed_beam(p, nx=p.seed_beam.nx * world)
        label += f", nx x{world} (weak scaling)"
    p.label = label
    return p, data


def config5_problem(rt, problem_mod, n: int):
    base = rt.datfile.load(ROOT / "tests" / "golden" / "ASE_small.dat.xz")
    return problem_mod.regrid_beam(problem_mod.resample_frequency(base, 512), nx=n, ny=n, a_centre=-1.0,
                                   b_centre=-4.5)


# --------------------------------------------------------------------------- probes on the box
def measure_hbm_peak(torch, dev) -> dict:
    """Device-to-device copy of 2 GiB (read + write counted) and a read-only pass: what this box's HBM
    delivers to a trivially coalesced kernel, quoted beside the 8 TB/s spec."""
    n = (2 << 30) // 8
    a = torch.empty(n, dtype=torch.float64, device=dev).normal_()
    b = torch.empty_like(a)
    best_copy = best_read = 1e9
    for _ in range(6):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        b.copy_(a)
        e1.record()
        a.sum()
        e2.record()
        torch.cuda.synchronize()
        best_copy = min(best_copy, e0.elapsed_time(e1))
        best_read = min(best_read, e1.elapsed_time(e2))
    del a, b
    torch.cuda.empty_cache()
    return {"copy_GBs": 2 * n * 8 / best_copy / 1e6, "read_GBs": n * 8 / best_read / 1e6,
            "how": "torch D2D copy of 2 GiB (read + write bytes) / torch.sum of 2 GiB (read bytes), best of 6, HIP events"}


def cpu_info() -> dict:
    """CPU model, the hardware threads this process may run on (its affinity mask -- what
    std::thread::hardware_concurrency() reports to the reference's `threads` method, RayTraceImage.cpp:409-413),
    one thread per physical core among them (sysfs topology), and the cgroup CPU quota if there is one."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(os.cpu_count() or 1))
    first_of_core, seen = [], set()
    for c in avail:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            first_of_core.append(c)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    return {"cpu_model": model, "nproc": os.cpu_count() or 1, "available": len(avail), "cpus": avail,
            "physical_cpus": first_of_core, "cgroup_cpu_quota": quota}


def cpu_baseline(problem, cell_steps: int) -> dict:
    """Rank 0, N = 1 only.  Times the UNMODIFIED reference RayTraceImageCPULoop (oracle/_ref/librt_ref.so,
    kind "reference") if it travelled to this box, else our bit-identical C restatement (kind "port"), on the
    WHOLE workload:
      * `value`: one host thread per hardware thread this process may use (the count the reference's `threads`
        method takes from std::thread::hardware_concurrency(), RayTraceImage.cpp:409-413), each pinned to its own
        CPU, each with a private image as RayTraceImageThreadLoop gives them (RayTraceImage.cpp:89-134).  The
        reference deals contiguous ray chunks; here the rays are dealt pixel by pixel, round-robin, so that the
        long rays (low pixel columns) spread over all threads -- the contiguous split leaves most threads idle
        while the first few finish (measured in round 3: 8.9x over one core on 16 threads);
      * `physical_cores`: the same with one thread per physical core;
      * `one_core`: every 16th pixel's rays on one thread, scaled by the sample's own ray-step count.
    3 runs each, min and mean; the threads are started before the clock and released together."""
    import numpy as np
    from oracle.binding import Oracle, Reference

    info = cpu_info()
    rays = problem.build_rays()
    n = len(rays)
    per_pixel = problem.beam.na * problem.beam.nb if problem.seed is None else problem.seed_beam.na * problem.seed_beam.nb
    n_pix = n // per_pixel
    if Reference.available():
        kind, eng = "reference", Reference()
        run = lambda r: eng.cpu_loop(problem, r)  # noqa: E731
    else:
        kind, eng = "port", Oracle()
        run = lambda r: eng.image_loop(problem, r, n_threads=1)  # noqa: E731
    by_pixel = rays[: n_pix * per_pixel].reshape(n_pix, per_pixel)

    def timed(cpus: list, runs: int = 3) -> dict:
        T = max(1, min(len(cpus), n_pix))
        parts = [np.ascontiguousarray(by_pixel[t::T].reshape(-1)) for t in range(T)]
        if n > n_pix * per_pixel:  # (a ragged tail, never the case for a grid)
            parts[0] = np.concatenate([parts[0], rays[n_pix * per_pixel:]])
        times = []
        for _ in range(runs):
            gate = threading.Barrier(T + 1)

            def work(t):
                try:
                    os.sched_setaffinity(0, {cpus[t]})
                except (AttributeError, OSError):
                    pass
                gate.wait()
                run(parts[t])

            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            for t in th:
                t.start()
            gate.wait()
            t0 = time.perf_counter()
            for t in th:
                t.join()
            times.append(time.perf_counter() - t0)
        return {"value": cell_steps / min(times), "unit": "ray-steps/s", "threads": T, "seconds_min": min(times),
                "seconds_mean": sum(times) / len(times), "runs": len(times)}

    me = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    try:
        allt = timed(info["cpus"])
        phys = timed(info["physical_cpus"]) if len(info["physical_cpus"]) < len(info["cpus"]) else None
        # 1 core on a strided pixel sample
        pix = np.arange(0, n_pix, 16, dtype=np.int64)
        ids = (pix[:, None] * per_pixel + np.arange(per_pixel)[None, :]).reshape(-1)
        sample = problem.build_rays(ids)
        steps_sample = int(Oracle().image_loop(problem, sample, n_threads=min(16, info["available"]))["counters"]["cell_steps"])
        t1 = []
        for _ in range(3):
            t0 = time.perf_counter()
            run(sample)
            t1.append(time.perf_counter() - t0)
    finally:
        if me is not None:
            os.sched_setaffinity(0, me)
    best = allt if (phys is None or allt["value"] >= phys["value"]) else phys
    rec = {"value": best["value"], "unit": "ray-steps/s", "cores": best["threads"], "kind": kind,
           "seconds_min": best["seconds_min"], "seconds_mean": best["seconds_mean"], "runs": best["runs"],
           "ms_per_image": best["seconds_min"] * 1e3,
           "all_hardware_threads": allt, "physical_cores": phys,
           "one_core": {"value": steps_sample / min(t1), "unit": "ray-steps/s", "seconds_min": min(t1),
                        "seconds_mean": sum(t1) / len(t1), "runs": len(t1),
                        "sample": f"every 16th pixel: {len(sample)} rays, {steps_sample} ray-steps, one thread"},
           "cpu_model": info["cpu_model"], "nproc": info["nproc"], "available_hardware_threads": info["available"],
           "cgroup_cpu_quota": info["cgroup_cpu_quota"],
           "pinning": "one thread per CPU of the affinity mask (sched_setaffinity)",
           "sample": f"whole workload ({n} rays, {cell_steps} ray-steps), pixels dealt round-robin to "
                     f"{best['threads']} pinned host threads, a private image per thread; RayTraceImageCPULoop per thread"}
    return rec


def measure_image_loop(backend, problem) -> dict:
    """ms/image as the reference harness times create_image's back-end call (CreateImage.cpp:147-152):
    rt_hip_image_loop with host pointers -- table pack + upload, ray list hand-over, kernels, download.
    PCIe-inclusive; never `value`."""
    rays = problem.build_rays()
    backend.image_loop(problem, rays)
    t, c = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        out = backend.image_loop(problem, rays)
        t.append((time.perf_counter() - t0) * 1e3)
        c.append(out["call_ms"])
    return {"ms_min": min(c), "ms_mean": sum(c) / len(c), "runs": len(c), "kernel_ms": out["stats"]["kernel_ms"],
            "python_wall_ms_min": min(t),
            "what": "rt_hip_image_loop (host-pointer C ABI behind RayTraceImageHipLoop): pack + upload tables, "
                    f"{len(rays)}-ray list handed over as host memory, kernels, download; wall clock of the C call (the "
                    "Python caller's marshalling and output allocation beside it in python_wall_ms_min)"}


def measure_config5(torch, backend, rt, problem_mod, dev, counters: dict) -> dict:
    """BASELINE config 5 on one GPU: 16,777,216 rays, nv = 512, the image (68.7 GB) is written once."""
    p = config5_problem(rt, problem_mod, 4096)
    b = p.beam
    image = torch.empty(b.nx * b.ny * b.nv, dtype=torch.float64, device=dev)
    iang = torch.zeros(b.na * b.nb, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    with backend.Plan(p, device=dev.index or 0) as plan:
        plan.set_ray_grid()
        ms = []
        for i in range(4):
            plan.run(stream, image.data_ptr(), iang.data_ptr())
            m, f = plan.kernel_times()
            if i:
                ms.append((m, f))
        st = plan.fetch(want_image=False)["stats"]
        # size-independent property at full size, on the device: sum(I_ang) = sum_pixels sum_k 2 dv_k image
        dv2 = torch.tensor(2.0 * p.beam.dv, dtype=torch.float64, device=dev)
        lhs = float(iang.sum().item())
        rhs = float((image.view(-1, b.nv) * dv2[None, :]).sum().item())
    del image
    torch.cuda.empty_cache()
    march = sum(m for m, _ in ms) / len(ms)  # mean of the 3 runs after the first (the fractions below are not best-of-N)
    freq = sum(f for _, f in ms) / len(ms)
    alg = algorithmic_bytes(st["n_rays"], st["cell_steps"], p.N - 1, b.nv, False, 0, b.nx * b.ny, b.na * b.nb)
    t = (march + freq) * 1e-3
    rec = {"workload": "synthetic 4096x4096x512, na = nb = 1", "rays": st["n_rays"], "ray_steps": st["cell_steps"],
           "march_ms": march, "freq_ms": freq, "kernel_ms": march + freq, "kernel_ms_min": min(m + f for m, f in ms), "runs": len(ms),
           "algorithmic_read_bytes": alg["path"], "algorithmic_write_bytes": alg["write"],
           "read_plus_write_GBs": (alg["path"] + alg["write"]) / t / 1e9, "store_GBs": alg["write"] / t / 1e9,
           "store_floor_ms": alg["write"] / (HBM_PEAK_GBS * 1e9) * 1e3,
           "ray_steps_per_sec": st["cell_steps"] / t,
           "iang_identity_rel_err": abs(lhs - rhs) / abs(rhs) if rhs else None,
           "limiter": f"{st['n_rays'] * 6 * b.nv / 1e9:.1f} G float64 exponential updates (f64 VALU issue) beside "
                      f"{alg['write'] / 1e9:.1f} GB of stores; not HBM reads"}
    c5 = counters.get("config5", {})
    rec["traffic"] = c5.get("hbm_bytes_per_step")
    rec["traffic_source"] = (counters.get("_source") + " (config5 counter pass)") if c5 else counters.get("_stale")
    fk = c5.get("pmc_sq", {}).get("rt_freq_kernel")
    rec.update(bounded_roof(alg["path"], t, rec["traffic"], None, counters.get("shader_clock_hz", 2.4e9)))
    if fk:
        rec["freq_kernel_issue"] = bounded_roof(alg["freq"], freq * 1e-3, None, fk, counters.get("shader_clock_hz", 2.4e9)).get("issue")
        rec["freq_kernel_issue_frac"] = bounded_roof(alg["freq"], freq * 1e-3, None, fk, counters.get("shader_clock_hz", 2.4e9)).get("issue_frac")
    return rec


def cpu_config5_sample(rt, problem_mod, full_rays: int, full_steps: int) -> dict:
    """CPU figure for BASELINE config 5: the reference cannot run it (nv = 512 is beyond its K_MAX), so
    the bit-identical C restatement (kind "port") is timed on a SAMPLE -- every 8th pixel in x and in y of the 4096 x 4096
    grid, 262 144 rays spread over the whole image, the cell sizes and tables of the full problem -- on the host threads of
    the box, and on one thread for every 64th pixel; the whole image is that time scaled by the ray count."""
    import copy

    import numpy as np
    from oracle.binding import Oracle

    info = cpu_info()
    eng = Oracle()
    p = config5_problem(rt, problem_mod, 4096)

    def sample(stride, part=0, parts=1):
        q = copy.copy(p)
        b = copy.copy(p.beam)
        b.x = np.ascontiguousarray(p.beam.x[stride // 2::stride][part::parts])
        b.y = np.ascontiguousarray(p.beam.y[stride // 2::stride])
        q.beam = b
        return q

    def timed(stride, cpus, runs=3):
        # one sub-problem per thread (its own pixel columns, its own small image: a private copy of the sample's
        # 268 MB image per thread, as the thread loop of the restatement keeps them, would be the thing measured)
        T = max(1, min(len(cpus), len(p.beam.x[stride // 2::stride])))
        subs = [sample(stride, t, T) for t in range(T)]
        t, steps = [], 0
        me = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
        try:
            for _ in range(runs):
                gate = threading.Barrier(T + 1)
                got = [0] * T

                def work(i):
                    try:
                        os.sched_setaffinity(0, {cpus[i]})
                    except (AttributeError, OSError):
                        pass
                    gate.wait()
                    got[i] = int(eng.image_loop(subs[i], n_threads=1)["counters"]["cell_steps"])

                th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
                for x in th:
                    x.start()
                gate.wait()
                t0 = time.perf_counter()
                for x in th:
                    x.join()
                t.append(time.perf_counter() - t0)
                steps = sum(got)
        finally:
            if me is not None:
                os.sched_setaffinity(0, me)
        n = sum(q.n_rays_total for q in subs)
        return {"rays": n, "ray_steps": steps, "threads": T, "seconds_min": min(t), "seconds_mean": sum(t) / len(t),
                "runs": runs, "value": steps / min(t), "unit": "ray-steps/s",
                "ms_per_image_extrapolated": min(t) * 1e3 * full_rays / n}

    many = timed(8, info["cpus"])
    one = timed(64, info["cpus"][:1])
    return {"value": many["value"], "unit": "ray-steps/s", "cores": many["threads"], "kind": "port",
            "ms_per_image_extrapolated": many["ms_per_image_extrapolated"], "all_threads": many, "one_core": one,
            "full_problem": {"rays": full_rays, "ray_steps": full_steps},
            "sample": "every 8th pixel in x and y (262 144 rays over the whole image, full-size cells and tables, nv = 512), "
                      "pixel columns dealt round-robin to pinned host threads, one small image each; one core: every 64th pixel; "
                      "the reference itself stops at nv < K_MAX = 100"}


def measure_seed_medium(torch, backend, rt, dev, counters: dict) -> dict:
    """The seeded half of BASELINE config 3 on one GPU: seed_small's tables on scale_problem(16), 124,848,000 rays,
    nv = 82 (seed_medium.dat itself is absent from the reference checkout).  March + gain-only frequency pass
    (Helper.h:569-581, RayTraceImageCPU.cpp:37-68); kernel times from HIP events on the launch stream."""
    p = rt.scale_problem(rt.datfile.load(ROOT / "tests" / "golden" / "seed_small.dat.xz"), 16.0)
    b = p.beam
    image = torch.zeros(b.nx * b.ny * b.nv, dtype=torch.float64, device=dev)
    iang = torch.zeros(b.na * b.nb, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    with backend.Plan(p, device=dev.index or 0) as plan:
        plan.set_ray_grid()
        ms = []
        for i in range(4):
            plan.run(stream, image.data_ptr(), iang.data_ptr())
            m, f = plan.kernel_times()
            if i:
                ms.append((m, f))
        st = plan.fetch(want_image=False)["stats"]
    torch.cuda.empty_cache()
    march = sum(m for m, _ in ms) / len(ms)  # mean of the 3 runs after the first
    freq = sum(f for _, f in ms) / len(ms)
    n_live = st["n_rays"] - st["n_escaped"]
    alg = algorithmic_bytes(st["n_rays"], st["cell_steps"], p.N - 1, b.nv, True, n_live, b.nx * b.ny, b.na * b.nb)
    t = (march + freq) * 1e-3
    rec = {"workload": "seed_medium stand-in: seed_small tables x scale_problem(16)", "rays": st["n_rays"],
           "rays_live": n_live, "ray_steps": st["cell_steps"], "nv": b.nv, "march_ms": march, "freq_ms": freq,
           "kernel_ms": march + freq, "kernel_ms_min": min(m + f for m, f in ms), "runs": len(ms),
           "ray_steps_per_sec": st["cell_steps"] / t,
           "algorithmic_bytes": alg["path"], "bytes_per_ray_step": alg["path"] / max(1, st["cell_steps"]),
           "failure_code": st.get("failure_code")}
    sm = counters.get("seed_medium", {})
    rec["traffic"] = sm.get("hbm_bytes_per_step")
    rec["traffic_source"] = (counters.get("_source") + " (seed_medium counter pass)") if sm else counters.get("_stale")
    clk = counters.get("shader_clock_hz", 2.4e9)
    ker, sq = sm.get("kernels", {}), sm.get("pmc_sq", {})
    rec.update(bounded_roof(alg["path"], t, rec["traffic"], None, clk))
    rec["dominant_kernel"] = {"kernel": "rt_march_kernel", "kernel_ms_avg": march, "algorithmic_bytes_per_launch": alg["march"],
                              **bounded_roof(alg["march"], march * 1e-3, ker.get("rt_march_kernel", {}).get("hbm_bytes_per_launch"),
                                             sq.get("rt_march_kernel"), clk)}
    rec["freq_kernel"] = {"kernel": "rt_freq_kernel", "kernel_ms_avg": freq, "algorithmic_bytes_per_launch": alg["freq"],
                          **bounded_roof(alg["freq"], freq * 1e-3, ker.get("rt_freq_kernel", {}).get("hbm_bytes_per_launch"),
                                         sq.get("rt_freq_kernel"), clk),
                          "note": "the contract's (256 + 8 K) R_live term prices a per-ray seed evaluation; the kernel reads "
                                  "tabulated per-axis seed factors instead, so contract_frac of this split exceeds 1"}
    return rec


def measure_small_files(torch, backend, rt, dev) -> dict:
    """BASELINE config 2: the two input files the reference ships, unchanged (tests/golden/*.dat.xz are the
    reference's ASE_small.dat / seed_small.dat), on one GPU: kernel times of a resident plan (mean of 5 runs after 2),
    ray-steps/s, the contract fraction, ms/image through rt_hip_image_loop as the harness times a back-end call
    (CreateImage.cpp:147-173), and the image against the committed output of the reference's own CPU loop
    (tests/golden/*_ref_cpu.npz, generated from oracle/_ref by tests/golden/make_golden.py)."""
    import numpy as np

    out = {}
    stream = torch.cuda.current_stream().cuda_stream
    for name in ("ASE_small", "seed_small"):
        p = rt.datfile.load(ROOT / "tests" / "golden" / f"{name}.dat.xz")
        b = p.beam
        seeded = p.seed is not None
        image = torch.zeros(b.nx * b.ny * b.nv, dtype=torch.float64, device=dev)
        iang = torch.zeros(b.na * b.nb, dtype=torch.float64, device=dev)
        with backend.Plan(p, device=dev.index or 0) as plan:
            plan.set_ray_grid()
            ms = []
            for i in range(7):
                plan.run(stream, image.data_ptr(), iang.data_ptr())
                m, f = plan.kernel_times()
                if i >= 2:
                    ms.append((m, f))
            st = plan.fetch(want_image=False)["stats"]
        march = float(np.mean([m for m, _ in ms]))
        freq = float(np.mean([f for _, f in ms]))
        n_live = st["n_rays"] - st["n_escaped"]
        alg = algorithmic_bytes(st["n_rays"], st["cell_steps"], p.N - 1, b.nv, seeded, n_live, b.nx * b.ny, b.na * b.nb)
        t = (march + freq) * 1e-3
        rays = p.build_rays()
        backend.image_loop(p, rays)
        wall, pywall = [], []  # the C call's own wall clock; with the Python caller's marshalling beside it
        for _ in range(5):
            t0 = time.perf_counter()
            res = backend.image_loop(p, rays)
            pywall.append((time.perf_counter() - t0) * 1e3)
            wall.append(res["call_ms"])
        rec = {"file": f"{name}.dat (reference input, unchanged)", "rays": st["n_rays"], "ray_steps": st["cell_steps"], "nv": b.nv,
               "march_ms": march, "freq_ms": freq, "kernel_ms_avg": march + freq, "runs": len(ms),
               "ray_steps_per_sec": st["cell_steps"] / t,
               "algorithmic_bytes": alg["path"], **bounded_roof(alg["path"], t),
               "march_contract_frac": alg["march"] / (march * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "ms_per_image": min(wall), "ms_per_image_mean": sum(wall) / len(wall), "python_wall_ms_min": min(pywall),
               "image_loop_ray_steps_per_sec": res["stats"]["cell_steps"] / (min(wall) * 1e-3),
               "failure_code": res["failure_code"]}
        ref = ROOT / "tests" / "golden" / f"{name}_ref_cpu.npz"
        if ref.exists():
            g = np.load(ref)
            rec["rel_l2_image_vs_reference_cpu"] = float(np.linalg.norm(res["image"] - g["image"]) / np.linalg.norm(g["image"]))
            rec["rel_l2_I_ang_vs_reference_cpu"] = float(np.linalg.norm(res["I_ang"] - g["I_ang"]) / np.linalg.norm(g["I_ang"]))
        out[name] = rec
    return out


# --------------------------------------------------------------------------- the product's multi-GPU arm
def cabi_multi_child(n_dev: int, workload: str) -> int:
    """Child process of measure_cabi_multi (fresh: it owns every device it uses): the workload through
    rt_hip_multi_image_loop(ndev = n_dev) -- what create_image(..., "hip-multigpu") and RayTraceImageHipMultiGPULoop
    call (replaces src/RayTraceImage.cpp:389-405) -- timed as the harness times a back-end call, and its image
    checked against the single-device plan in this very run."""
    import importlib

    import numpy as np

    sys.path.insert(0, str(ROOT))
    rt = importlib.import_module("raytrace-miniapp_amd")
    backend = importlib.import_module("raytrace-miniapp_amd.backend")
    problem_mod = importlib.import_module("raytrace-miniapp_amd.problem")
    p, _ = build_workload(rt, problem_mod, workload, 1, "strong")
    rays = p.build_rays()
    with backend.Plan(p, device=0) as plan:
        plan.set_ray_grid()
        one = plan.run().fetch()
    t0 = time.perf_counter()
    warm = backend.multi_image_loop(p, rays, n_devices=n_dev)  # loads librccl, builds the communicator
    first_ms = (time.perf_counter() - t0) * 1e3
    wall, kern = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        out = backend.multi_image_loop(p, rays, n_devices=n_dev)
        wall.append(out["call_ms"])  # wall clock of the C call (the Python caller's marshalling is not the product's)
        kern.append(out["stats"]["kernel_ms"])
    scale = float(np.abs(one["image"]).max())
    err_img = float(np.abs(out["image"] - one["image"]).max() / scale)
    err_ang = float(np.abs(out["I_ang"] - one["I_ang"]).max() / np.abs(one["I_ang"]).max())
    loop = os.environ.get("RT_HIP_MULTI_LOOPBACK")
    rec = {"entry": "rt_hip_multi_image_loop", "ndev": n_dev, "devices_visible": backend.HipLibrary.get().device_count(),
           "mode": {1: "pixel-column tiles + one grouped ncclSend/ncclRecv gather", 2: "ray chunks + one ncclReduce(sum)"}.get(out["mode"], out["mode"]),
           "collective": ("loop-back rehearsal on device 0 (RT_HIP_MULTI_LOOPBACK=" + loop + "): the RCCL call replaced by "
                          "device-to-device copies") if loop else f"RCCL, {n_dev} rank(s) in one process",
           "ms_per_image": min(wall), "ms_per_image_mean": sum(wall) / len(wall), "calls": len(wall),
           "kernel_ms_max_over_devices": float(np.mean(kern)),
           "ray_steps": int(out["stats"]["cell_steps"]), "ray_steps_per_sec": out["stats"]["cell_steps"] / (min(wall) * 1e-3),
           "first_call_ms_incl_librccl_and_communicator": first_ms,
           "max_abs_err_image_vs_single_device": err_img, "max_abs_err_I_ang_vs_single_device": err_ang,
           "image_matches_single_device_1e-12": bool(err_img <= 1e-12 and err_ang <= 1e-12),
           "failure_code": out["failure_code"] | warm["failure_code"],
           "what": f"host-pointer call: {len(rays)}-ray list as host memory, tables packed and uploaded per device, kernels, "
                   "one collective, one download -- PCIe-inclusive, never `value`"}
    print("CABI_MULTI " + json.dumps(rec), flush=True)
    return 0


def measure_cabi_multi(n_dev: int, workload: str, share_gpu: bool) -> dict:
    """Runs cabi_multi_child in a fresh process (never a re-exec of this GPU-initialised one) and returns its record."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                        "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # a collective that never completes ends the child's call after a minute (rt_multi.hip), and the child itself is
    # given five: the bench line must not wait a quarter of an hour for a sub-record
    env.setdefault("RT_HIP_MULTI_TIMEOUT_MS", "60000")
    if share_gpu and n_dev > 1:  # rehearsal on a box with fewer GPUs than ranks
        env["RT_HIP_MULTI_LOOPBACK"] = str(n_dev)
    r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--cabi-multi-child", str(n_dev), "--workload", workload],
                       env=env, capture_output=True, text=True, timeout=300)
    for ln in r.stdout.splitlines():
        if ln.startswith("CABI_MULTI "):
            return json.loads(ln[len("CABI_MULTI "):])
    return {"error": f"child exited with {r.returncode}", "stderr_tail": r.stderr[-800:]}


# --------------------------------------------------------------------------- main
def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("strong", "weak"), default=None)
    ap.add_argument("--workload", choices=("standin", "seed_medium", "config5"), default="standin")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 sub-record of the default N = 1 run")
    ap.add_argument("--no-seed-medium", action="store_true", help="skip the seeded sub-record of the default N = 1 run")
    ap.add_argument("--no-extras", action="store_true", help="only the timed loop (profiling runs)")
    ap.add_argument("--no-small", action="store_true", help="skip the ASE_small / seed_small sub-record of the default N = 1 run")
    ap.add_argument("--no-cabi-multi", action="store_true", help="skip the rt_hip_multi_image_loop sub-record (a child process)")
    ap.add_argument("--clock-ramp-steps", type=int, default=None,
                    help="further untimed steps after --warmup, before the timed region (reported in `warmup`); default 0: "
                         "the driver's --warmup is the protocol (round 4 ran 64 by default; the steady_state loop beside "
                         "the timed region shows the steady clock)")
    ap.add_argument("--cabi-multi-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-assemble", action="store_true",
                    help="N > 1: leave the tiles on their GPUs (config 5: gathering the 68.7 GB image takes longer than "
                         "tracing it, SURVEY.md 7.3-6); the collective is then reported as absent")
    args = ap.parse_args()

    if args.cabi_multi_child:
        return cabi_multi_child(args.cabi_multi_child, args.workload)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)

    import importlib

    import numpy as np
    import torch
    import torch.distributed as dist

    sys.path.insert(0, str(ROOT))
    rt = importlib.import_module("raytrace-miniapp_amd")
    backend = importlib.import_module("raytrace-miniapp_amd.backend")
    multigpu = importlib.import_module("raytrace-miniapp_amd.multigpu")
    problem_mod = importlib.import_module("raytrace-miniapp_amd.problem")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    scaling = args.scaling or "strong"  # the workload is fixed as N grows (BASELINE config 4)
    # Rehearsal on a box with fewer GPUs than ranks (never a measurement): RT_BENCH_BACKEND=gloo stages the
    # collectives through host memory, RT_BENCH_SHARE_GPU=1 puts every rank on device 0.
    dist_backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if os.environ.get("RT_BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if dist_backend == "nccl" else torch.device("cpu")  # where the bookkeeping all-reduces live
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(dist_backend)

    full, data = build_workload(rt, problem_mod, args.workload, world, scaling)
    mine = multigpu.shard(full, rank, world)
    b = mine.beam
    seeded = mine.seed is not None
    plan = backend.Plan(mine, device=local)
    plan.set_ray_grid()
    n_ang = b.na * b.nb
    # one buffer per rank, (tile | I_ang): the assembly is ONE collective
    asm = multigpu.Assembler(full, rank, world, dev, via_host=(dist_backend != "nccl"))
    image, iang = asm.image, asm.iang
    stream = torch.cuda.current_stream().cuda_stream

    plan.set_timing_ring(max(1, args.steps))  # HIP events around each kernel of every timed step, read after the loop

    def step():
        plan.run(stream, image.data_ptr(), iang.data_ptr())
        if world > 1 and not args.no_assemble:
            asm.assemble()

    for _ in range(args.warmup):
        step()
    # The W warm-up steps of a 3 ms step last ~10 ms: the shader clock is still ramping when the timed region
    # starts (the first timed steps can run a few per cent longer than the steady state).  --clock-ramp-steps further
    # UNTIMED steps (default 0 since round 5: the driver's --warmup is the protocol; `steady_state` below is the long loop)
    # may be asked for; the line's `warmup` is the total number of untimed steps that ran, `warmup_requested` the
    # --warmup part of it.
    extra_warm = max(0, args.clock_ramp_steps) if args.clock_ramp_steps is not None else 0
    for _ in range(extra_warm):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # nothing here waits for the device: the steps queue up back to back
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    kms = plan.ring_times()  # (march_ms, freq_ms) of the timed steps, recorded on the launch stream
    fused = plan.last_fused()  # one launch for the whole path: the first time is the launch's, the second ~0

    st = plan.fetch(want_image=False)
    stats = st["stats"]
    march_ms = float(np.mean([k[0] for k in kms]))
    freq_ms = float(np.mean([k[1] for k in kms]))
    t = torch.tensor([dt, march_ms + freq_ms], dtype=torch.float64, device=cdev)
    cnt = torch.tensor([float(stats["cell_steps"]), float(stats["n_rays"]), float(stats["n_rays"] - stats["n_escaped"])],
                       dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    dt_max, kernel_ms_max = float(t[0].item()), float(t[1].item())
    steps_all, rays_all = int(cnt[0].item()), int(cnt[1].item())
    # A second, longer loop of the same step (>= 6 s of back-to-back work; never `value`): the timed region of a
    # 3 ms step is a fraction of a second, and an outside GPU-busy sampler that looks every ~5 s (the driver's
    # showed 0 % in every sample of round 3, whose loop here lasted 0.6 s) needs a busy stretch longer than its
    # period.  The number of steps comes from the all-reduced time, so that every rank issues the same number
    # of collectives.
    steady = None
    if not args.no_extras:
        n_ss = max(args.steps, int(6.0 / max(dt_max / args.steps, 1e-4)) + 1)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        s0 = time.perf_counter()
        for _ in range(n_ss):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        s_dt = time.perf_counter() - s0
        steady = {"steps": n_ss, "seconds": s_dt, "ms_per_step": s_dt / n_ss * 1e3}

    # N > 1, strong scaling: the per-rank shard of the 6.4 M-ray stand-in is under a millisecond of kernels at
    # N = 8, where the tail of the persistent march (one long ray is ~0.2 ms) and the collective weigh most.
    # The weak-scaling form of the same workload (every rank the whole single-GPU workload) is measured after
    # the timed run, same protocol, and reported beside it -- extra information, never `value`.
    weak = None
    if world > 1 and scaling == "strong" and not args.no_extras and args.workload == "standin":
        wfull, _ = build_workload(rt, problem_mod, args.workload, world, "weak")
        wmine = multigpu.shard(wfull, rank, world)
        wplan = backend.Plan(wmine, device=local)
        wplan.set_ray_grid()
        wasm = multigpu.Assembler(wfull, rank, world, dev, via_host=(dist_backend != "nccl"))

        def wstep():
            wplan.run(stream, wasm.image.data_ptr(), wasm.iang.data_ptr())
            wasm.assemble()

        for _ in range(2):
            wstep()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        wsteps = max(3, args.steps // 4)
        for _ in range(wsteps):
            wstep()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        wdt = torch.tensor([time.perf_counter() - w0], dtype=torch.float64, device=cdev)
        wst = wplan.fetch(want_image=False)["stats"]
        wcnt = torch.tensor([float(wst["cell_steps"])], dtype=torch.float64, device=cdev)
        dist.all_reduce(wdt, op=dist.ReduceOp.MAX)
        dist.all_reduce(wcnt, op=dist.ReduceOp.SUM)
        weak = {"value": float(wcnt.item()) / (float(wdt.item()) / wsteps), "unit": "ray-steps/s",
                "ms_per_step": float(wdt.item()) / wsteps * 1e3, "steps": wsteps,
                "rays_per_gpu": wst["n_rays"], "workload": wfull.label}
        wplan.close()

    rc = 0
    if rank == 0:
        ms_step = dt_max / args.steps * 1e3
        kernel_ms = march_ms + freq_ms
        L, K = mine.N - 1, b.nv
        n_live = stats["n_rays"] - stats["n_escaped"]
        alg = algorithmic_bytes(stats["n_rays"], stats["cell_steps"], L, K, seeded, n_live, b.nx * b.ny, n_ang)
        counters = committed_counters() if (world == 1 and args.workload == "standin") else {}
        hbm = counters.get("hbm", {})
        sq = counters.get("pmc_sq", {})

        def roof(name, kname, ms):
            # The contract's yardstick: ALGORITHMIC bytes per launch over the kernel's mean time
            # against the HBM peak.  It is not HBM utilisation -- the tables are LDS / cache resident -- so the bytes
            # that really crossed the HBM interface (counter passes) and the resource that binds the kernel (issue
            # rate, from the same passes) are given beside it, each labelled.
            ach = alg[name] / (ms * 1e-3) / 1e9
            traffic = hbm.get(kname, {}).get("hbm_bytes_per_launch")
            cf = ach / HBM_PEAK_GBS
            r = {"bound": "hbm", "achieved": ach if cf <= 1.0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": cf if cf <= 1.0 else None, "contract_achieved": ach, "contract_frac": cf,
                 "achieved_basis": "algorithmic bytes of the contract formula (16 R + C_step S | 4 K 3 L R), not HBM traffic"
                                   + ("" if cf <= 1.0 else "; above 1 (cache-served rows): quoted as contract_frac only, frac is null"),
                 "traffic": traffic,
                 "traffic_source": (counters["_source"] + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                    "command, gfx950 corrections applied; not measured in this run") if kname in hbm
                 else counters.get("_stale"),
                 "kernel": kname, "kernel_ms_avg": ms, "algorithmic_bytes_per_launch": alg[name]}
            if traffic:
                r["hbm_measured"] = {"GBs": traffic / (ms * 1e-3) / 1e9, "frac_of_peak": traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "what": "counter bytes per launch / this run's kernel time"}
            c = sq.get(kname, {}).get("avg", {})
            if c.get("SQ_INSTS_VALU"):
                # what actually binds the kernel: wave-instructions issued per SIMD (every class takes an
                # issue slot, tools/ubench/issue_cost.hip); peak = one instruction per 2 cycles per SIMD
                instr = c["SQ_INSTS_VALU"] + c.get("SQ_INSTS_SALU", 0.0)
                clock = counters.get("shader_clock_hz", 2.4e9)
                rate = instr / (ms * 1e-3)
                peak = N_SIMD * clock / 2.0
                r["binding"] = {"bound": "valu+salu issue", "achieved": rate / 1e9, "peak": peak / 1e9,
                                "unit": "G wave-instr/s", "frac": min(rate / peak, 1.0), "raw_frac": rate / peak,
                                "valu_instr_per_launch": c["SQ_INSTS_VALU"], "salu_instr_per_launch": c.get("SQ_INSTS_SALU"),
                                "source": counters["_source"] + " (committed SQ counter pass, instruction counts per launch)"}
                r["secondary"] = r["binding"]
            return r

        if fused:
            # ONE launch runs the whole path (rt_fused.hip: march and frequency pass as two phases of the same persistent
            # waves): its algorithmic bytes are the path's, 16 R + C_step S + 4 K 3 L R
            kernels = [roof("path", "rt_fused_kernel", march_ms + freq_ms)]
            kernels[0]["achieved_basis"] = ("algorithmic bytes of the contract formula for the whole path "
                                            "(16 R + C_step S + 4 K 3 L R): one launch marches and integrates; not HBM traffic")
        else:
            kernels = [roof("march", "rt_march_kernel", march_ms), roof("freq", "rt_freq_kernel", freq_ms)]
        dominant = max(kernels, key=lambda r: r["kernel_ms_avg"])
        path_ach = alg["path"] / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "ray_steps_per_sec", "value": steps_all / (dt_max / args.steps), "unit": "ray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup + extra_warm, "warmup_requested": args.warmup,
            "warmup_extra_steps": extra_warm,
            "ms_per_step": ms_step,
            "kernel_ms": kernel_ms_max, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32+f64",  # float32 march (bit-exact), float64 frequency integration
            "data": data,
            "config": {"workload": full.label, "rays_per_gpu": stats["n_rays"], "rays_total": rays_all,
                       "ray_steps_total": steps_all, "nv": K, "N": mine.N,
                       "image": [full.beam.ny, full.beam.nx, K],
                       "parallelism": ("pixel" if not seeded else "source") + f"-columns x{world}"},
            # the dominant kernel, as the contract asks; every kernel of the path and the path as a
            # whole are listed next to it
            "roofline": dominant,
            "roofline_kernels": kernels,
            "roofline_path": {**bounded_roof(alg["path"], kernel_ms * 1e-3), "kernel_ms_sum": kernel_ms,
                              "algorithmic_bytes": alg["path"],
                              "bytes_per_ray_step": alg["path"] / max(1, stats["cell_steps"])},
        }
        line["launches_per_step"] = 1 if fused else 2
        if fused and world == 1 and not args.no_extras:
            # the same workload as two kernels (RT_HIP_FUSED=2), for the per-kernel figures of earlier rounds: mean of
            # 8 runs after 2, HIP events on the launch stream
            try:
                os.environ["RT_HIP_FUSED"] = "2"
                with backend.Plan(mine, device=local) as plan2:
                    plan2.set_ray_grid()
                    tk = []
                    for i in range(10):
                        plan2.run(stream, image.data_ptr(), iang.data_ptr())
                        if i >= 2:
                            tk.append(plan2.kernel_times())
                m2 = float(np.mean([t[0] for t in tk]))
                f2 = float(np.mean([t[1] for t in tk]))
                line["roofline_two_kernel"] = {
                    "what": "the same step as two launches (RT_HIP_FUSED=2): march kernel, records in HBM / L2, frequency kernel",
                    "runs": len(tk), "kernel_ms_sum": m2 + f2, "one_launch_speedup": (m2 + f2) / kernel_ms,
                    "march": {"kernel": "rt_march_kernel", "kernel_ms_avg": m2, "algorithmic_bytes_per_launch": alg["march"],
                              **bounded_roof(alg["march"], m2 * 1e-3, None, counters.get("two_kernel", {}).get("pmc_sq", {}).get("rt_march_kernel"),
                                             counters.get("shader_clock_hz", 2.4e9))},
                    "freq": {"kernel": "rt_freq_kernel", "kernel_ms_avg": f2, "algorithmic_bytes_per_launch": alg["freq"],
                             **bounded_roof(alg["freq"], f2 * 1e-3, None, counters.get("two_kernel", {}).get("pmc_sq", {}).get("rt_freq_kernel"),
                                            counters.get("shader_clock_hz", 2.4e9))},
                    "path_contract_frac": alg["path"] / ((m2 + f2) * 1e-3) / 1e9 / HBM_PEAK_GBS}
            except Exception as exc:  # noqa: BLE001
                line["roofline_two_kernel"] = {"error": repr(exc)}
            finally:
                os.environ.pop("RT_HIP_FUSED", None)
        if steady is not None:
            steady["value"] = stats["cell_steps"] / (steady["ms_per_step"] * 1e-3) * (world if scaling == "weak" else 1)
            if world > 1 and scaling == "strong":
                steady["value"] = steps_all / (steady["ms_per_step"] * 1e-3)
            line["steady_state"] = steady
        if world > 1:
            line["multi_gpu"] = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
                                 "kernel_ms_max_over_ranks": kernel_ms_max,
                                 "assembly_ms": max(0.0, ms_step - kernel_ms_max),
                                 "collective": "none (--no-assemble: tiles stay on their GPUs)" if args.no_assemble else asm.describe()}
            if weak is not None:
                line["multi_gpu"]["weak_scaling"] = weak
        if world == 1 and not args.no_extras:
            try:
                line["peak_measured"] = measure_hbm_peak(torch, dev)
            except Exception as exc:  # noqa: BLE001
                line["peak_measured"] = {"error": repr(exc)}
            try:
                il = measure_image_loop(backend, mine)
                line["ms_per_image"] = il["ms_min"]
                line["image_loop"] = il
            except Exception as exc:  # noqa: BLE001
                line["image_loop"] = {"error": repr(exc)}
            if args.workload == "standin" and not args.no_config5:
                plan.close()
                try:
                    line["roofline_config5"] = measure_config5(torch, backend, rt, problem_mod, dev, counters)
                    if not args.no_cpu_baseline:
                        c5 = line["roofline_config5"]
                        try:
                            c5["cpu_baseline"] = cpu_config5_sample(rt, problem_mod, c5["rays"], c5["ray_steps"])
                        except Exception as exc:  # noqa: BLE001
                            c5["cpu_baseline"] = {"error": repr(exc)}
                except Exception as exc:  # noqa: BLE001
                    line["roofline_config5"] = {"error": repr(exc)}
            if args.workload == "standin" and not args.no_seed_medium:
                plan.close()
                try:
                    line["roofline_seed_medium"] = measure_seed_medium(torch, backend, rt, dev, counters)
                except Exception as exc:  # noqa: BLE001
                    line["roofline_seed_medium"] = {"error": repr(exc)}
            if args.workload == "standin" and not args.no_small:
                plan.close()
                try:
                    line["roofline_small"] = measure_small_files(torch, backend, rt, dev)
                except Exception as exc:  # noqa: BLE001
                    line["roofline_small"] = {"error": repr(exc)}
            if not args.no_cpu_baseline:
                try:
                    line["cpu_baseline"] = cpu_baseline(mine, stats["cell_steps"])
                except Exception as exc:  # noqa: BLE001
                    line["cpu_baseline"] = {"error": repr(exc)}
        # The product's own multi-GPU arm (rt_hip_multi_image_loop, what create_image(..., "hip-multigpu") calls) on
        # the same workload with ndev = N, in a fresh child process while this job's ranks sit idle on the host
        # (ranks 1 .. N-1 wait on the rendezvous store, not in a GPU-side barrier).  Beside it, `value` above is the
        # one-process-per-GPU form (multigpu.Assembler over torch.distributed).
        if not args.no_extras and not args.no_cabi_multi and args.workload in ("standin", "seed_medium"):
            plan.close()
            torch.cuda.synchronize()
            try:
                cab = measure_cabi_multi(world, args.workload, bool(os.environ.get("RT_BENCH_SHARE_GPU")))
            except Exception as exc:  # noqa: BLE001
                cab = {"error": repr(exc)}
            line["multi_gpu_cabi"] = cab
            if world > 1:
                line["multi_gpu"]["cabi"] = cab
        print(json.dumps(line), flush=True)
    if world > 1:
        # host-side wait for rank 0's child process (a GPU-side barrier would spin on the devices the child measures)
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("rt_bench_cabi_done", "1")
        else:
            import datetime

            store.wait(["rt_bench_cabi_done"], datetime.timedelta(seconds=1200))
    plan.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())

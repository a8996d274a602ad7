#!/usr/bin/env python3
"""bench.py -- throughput of the ray-trace imaging hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ...`)

A *step* is one pass of the hot path (zero outputs + trace kernel [+ the RCCL
assembly of image tiles when N > 1]) over the workload, with every input
already resident in HBM.

Workload (BASELINE.json metric: "ray-steps/sec for ASE_medium"): ASE_medium.dat
is absent from the reference checkout (.MISSING_LARGE_BLOBS), so the workload
is the stand-in SURVEY.md 8(d) prescribes: ASE_small's plasma tables on the
reference's own enlargement rule scale_problem(16) -> euv grid 120x50x38x28 =
6,384,000 rays, nv = 52, N = 3.  At N > 1 the image gets N times as many pixel
columns (weak scaling: 6,384,000 rays per GPU); columns are dealt round-robin
to the ranks, each rank traces its tile, tiles are gathered to rank 0 over RCCL
and I_ang is sum-reduced.

One JSON line is printed by rank 0; see DESIGN.md "Measurement" for the fields.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
rt = importlib.import_module("raytrace-miniapp_amd")
backend = importlib.import_module("raytrace-miniapp_amd.backend")
multigpu = importlib.import_module("raytrace-miniapp_amd.multigpu")
problem_mod = importlib.import_module("raytrace-miniapp_amd.problem")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_read_bytes(n_rays: int, cell_steps: int, L: int, K: int, seeded: bool, n_live: int = 0) -> dict:
    """B_read = 16 R + C_step S + 4 K 3 L R [+ (256 + 8K) R_live],
    split between the two kernels of the path: the march gathers the plasma grid
    (16 R + C_step S), the frequency pass reads the lineshape rows (4 K 3 L R)."""
    c_step = 80 if seeded else 96
    march = 16 * n_rays + c_step * cell_steps
    freq = 4 * K * 3 * L * n_rays + ((256 + 8 * K) * n_live if seeded else 0)
    return {"march": march, "freq": freq, "path": march + freq}


def build_workload(world: int):
    base = rt.datfile.load(ROOT / "tests" / "golden" / "ASE_small.dat.xz")
    med = rt.scale_problem(base, 16.0)
    if world > 1:
        med = problem_mod.regrid_beam(med, nx=med.beam.nx * world)
    med.label = "ASE_medium stand-in: ASE_small tables x scale_problem(16)" + (
        f", nx x{world} (weak scaling)" if world > 1 else "")
    return med


def cpu_baseline(problem, cell_steps: int) -> dict:
    """Rank 0, N = 1 only.  Times the UNMODIFIED reference RayTraceImageCPULoop
    (oracle/_ref/librt_ref.so, kind "reference") if it travelled to this box,
    else our bit-identical C restatement (kind "port"), on the whole workload,
    split in contiguous ray chunks over host threads exactly as the reference's
    `threads` method does (RayTraceImage.cpp:89-134)."""
    from oracle.binding import Oracle, Reference

    rays = problem.build_rays()
    n = len(rays)
    cores = max(1, min(16, os.cpu_count() or 1))
    chunk = n // cores + 1
    if Reference.available():
        kind, eng = "reference", Reference()
        run = lambda r: eng.cpu_loop(problem, r)  # noqa: E731
    else:
        kind, eng = "port", Oracle()
        run = lambda r: eng.image_loop(problem, r, n_threads=1)  # noqa: E731
    parts = [rays[i * chunk:(i + 1) * chunk] for i in range(cores)]
    parts = [p for p in parts if len(p)]
    th = [threading.Thread(target=run, args=(p,)) for p in parts]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": cell_steps / dt, "unit": "ray-steps/s", "cores": len(parts), "kind": kind,
            "seconds": dt, "ms_per_image": dt * 1e3,
            "sample": f"whole workload ({n} rays, {cell_steps} ray-steps), contiguous chunks on "
                      f"{len(parts)} host threads; RayTraceImageCPULoop per chunk"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    full = build_workload(world)
    mine = multigpu.shard(full, rank, world)
    b = mine.beam
    plan = backend.Plan(mine, device=local)
    plan.set_ray_grid()
    image = torch.zeros(b.nx * b.ny * b.nv, dtype=torch.float64, device=dev)
    iang = torch.zeros(b.na * b.nb, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        plan.run(stream, image.data_ptr(), iang.data_ptr())
        if world > 1:
            multigpu.assemble(full, image, iang, rank, world)
        return plan.kernel_times()  # HIP events on the launch stream, recorded around each kernel

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = []
    for _ in range(args.steps):
        kms.append(step())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    st = plan.fetch(want_image=False)
    stats = st["stats"]
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    cnt = torch.tensor([float(stats["cell_steps"]), float(stats["n_rays"])], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    dt_max = float(t.item())
    steps_all, rays_all = int(cnt[0].item()), int(cnt[1].item())

    if rank == 0:
        ms_step = dt_max / args.steps * 1e3
        march_ms = float(np.mean([k[0] for k in kms]))
        freq_ms = float(np.mean([k[1] for k in kms]))
        kernel_ms = march_ms + freq_ms
        L, K = mine.N - 1, b.nv
        alg = algorithmic_read_bytes(stats["n_rays"], stats["cell_steps"], L, K, mine.seed is not None)
        traffic = {}
        tf = ROOT / "profiles" / "traffic_latest.json"
        if tf.exists() and world == 1:
            try:
                traffic = json.loads(tf.read_text()).get("kernels", {})
            except Exception:  # noqa: BLE001
                traffic = {}

        def roof(name, kname, ms):
            ach = alg[name] / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic.get(kname), "kernel": kname, "kernel_ms_avg": ms,
                    "algorithmic_bytes_per_launch": alg[name]}

        kernels = [roof("march", "rt_march_kernel", march_ms), roof("freq", "rt_freq_kernel", freq_ms)]
        dominant = max(kernels, key=lambda r: r["kernel_ms_avg"])
        path_ach = alg["path"] / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "ray_steps_per_sec", "value": steps_all / (dt_max / args.steps), "unit": "ray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "ms_per_image": ms_step, "kernel_ms": kernel_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32+f64",  # float32 march (bit-exact), float64 frequency integration
            "data": "ASE_small.dat plasma tables (reference input) on the synthetic scale_problem(16) ray grid; "
                    "ASE_medium.dat itself is absent from the reference checkout",
            "config": {"workload": full.label, "rays_per_gpu": stats["n_rays"], "rays_total": rays_all,
                       "ray_steps_total": steps_all, "nv": K, "N": mine.N,
                       "image": [full.beam.ny, full.beam.nx, K], "parallelism": f"pixel-columns x{world}"},
            # the dominant kernel, as the contract asks; every kernel of the path and the
            # path as a whole are listed next to it
            "roofline": dominant,
            "roofline_kernels": kernels,
            "roofline_path": {"bound": "hbm", "achieved": path_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": path_ach / HBM_PEAK_GBS, "kernel_ms_sum": kernel_ms,
                              "algorithmic_bytes": alg["path"],
                              "bytes_per_ray_step": alg["path"] / max(1, stats["cell_steps"])},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(mine, stats["cell_steps"])
            except Exception as exc:  # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)
    plan.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

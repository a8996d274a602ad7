"""Multi-GPU forms of the path.

Two drivers over the same per-device plan:

thread_loop(problem)
    One process, one host thread per device -- the analogue of the reference's
    "cuda-multigpu" arm (RayTraceImageThreadLoop, src/RayTraceImage.cpp:89-134,
    dispatched at :389-405): contiguous ray chunks, private images, summed in
    device order.  Unlike the reference, the device is bound INSIDE the worker
    (every C-ABI entry calls hipSetDevice itself), so the parent-thread setGPU
    defect (RayTraceImage.cpp:116) cannot occur.

shard / assemble (one process per GPU, torch.distributed; backend "nccl" is
    RCCL over xGMI on ROCm, "gloo" in the CPU tests)
    ASE    : pixel columns are dealt round-robin to ranks (problem.shard_columns);
             a rank's image is a compact tile [ny][nx_local][nv]; assembly is an
             RCCL *gather* of tiles to rank 0 plus an interleave, and a
             sum-reduce of the na*nb doubles of I_ang.  No other exchange.
    seeded : source columns are dealt the same way, every rank holds a full
             private image, assembly is a sum-reduce of image and I_ang (the
             analogue of intensity_step_struct::sum_reduce,
             src/RayTraceStructures.cpp:1603-1646).
"""
from __future__ import annotations

import threading

import numpy as np

from .problem import Problem, shard_columns


# --------------------------------------------------------------------------- one process, N devices
def thread_loop(problem: Problem, n_devices: int | None = None) -> dict:
    from .backend import HipLibrary, Plan

    hl = HipLibrary.get()
    ndev = hl.device_count() if n_devices is None else n_devices
    if ndev < 1:
        from .backend import RayTraceError
        raise RayTraceError("Hip-MultiGPU is not availible: no device")
    first, stride = problem.N_start, problem.N_parallel
    nt = problem.n_rays_total
    n_own = 0 if first >= nt else (nt - first + stride - 1) // stride
    chunk = n_own // ndev + 1  # RayTraceImage.cpp:107: rays.size()/N_threads + 1
    results: list = [None] * ndev
    errors: list = [None] * ndev

    def work(d: int) -> None:
        try:
            begin = min(d * chunk, n_own)
            cnt = min(chunk, n_own - begin)
            with Plan(problem, device=d, lib=hl) as plan:
                plan.set_ray_grid(first + begin * stride, stride, cnt)
                results[d] = plan.run().fetch()
        except Exception as exc:  # noqa: BLE001
            errors[d] = exc

    threads = [threading.Thread(target=work, args=(d,)) for d in range(ndev)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    b = problem.beam
    image = np.zeros(b.nx * b.ny * b.nv)
    iang = np.zeros(b.na * b.nb)
    code = 0
    failed = []
    stats = dict(n_rays=0, cell_steps=0, n_escaped=0, n_skipped=0, kernel_ms=0.0, total_ms=0.0)
    for r in results:  # join order = device order, as the reference sums
        image += r["image"]
        iang += r["I_ang"]
        code |= r["failure_code"]
        failed.extend(list(r["failed_rays"]))
        for k in ("n_rays", "cell_steps", "n_escaped", "n_skipped"):
            stats[k] += r["stats"][k]
        stats["kernel_ms"] = max(stats["kernel_ms"], r["stats"]["kernel_ms"])
        stats["total_ms"] = max(stats["total_ms"], r["stats"]["total_ms"])
    return dict(image=image, I_ang=iang, failure_code=code, failed_rays=np.array(failed), stats=stats)


# --------------------------------------------------------------------------- one process per GPU
def shard(problem: Problem, rank: int, world: int) -> Problem:
    return shard_columns(problem, rank, world)


def tile_columns(nx: int, rank: int, world: int) -> int:
    return len(range(rank, nx, world))


def assemble(problem: Problem, tile_image, tile_iang, rank: int, world: int, group=None, dst: int = 0):
    """Assemble the final image on rank `dst` from per-rank results.

    tile_image / tile_iang are torch tensors (device tensors under RCCL, CPU
    tensors under gloo) holding this rank's result of shard(problem, rank, world).
    Returns (image, I_ang) as flat tensors on rank dst, (None, None) elsewhere.
    """
    import torch
    import torch.distributed as dist

    b = problem.beam
    if world == 1:
        return tile_image.reshape(-1), tile_iang.reshape(-1)
    iang = tile_iang.reshape(-1).clone()
    dist.reduce(iang, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if problem.seed is not None:
        img = tile_image.reshape(-1).clone()
        dist.reduce(img, dst=dst, op=dist.ReduceOp.SUM, group=group)
        return (img, iang) if rank == dst else (None, None)
    # ASE: gather of equal-sized tiles (padded to the widest), then interleave
    ncol_max = tile_columns(b.nx, 0, world)
    ncol = tile_columns(b.nx, rank, world)
    tile = tile_image.reshape(b.ny, ncol, b.nv)
    if ncol != ncol_max:
        pad = torch.zeros((b.ny, ncol_max - ncol, b.nv), dtype=tile.dtype, device=tile.device)
        tile = torch.cat([tile, pad], dim=1)
    tile = tile.contiguous()
    parts = _gather_tiles(tile, rank, world, dst, group)
    if rank != dst:
        return None, None
    if b.nx % world == 0:
        # equal tiles: column i = r + world * c  <=>  stack the tiles behind the column axis (one kernel)
        full = torch.stack(parts, dim=2).reshape(b.ny, b.nx, b.nv)
    else:
        full = torch.empty((b.ny, b.nx, b.nv), dtype=tile.dtype, device=tile.device)
        for r in range(world):
            full[:, r::world, :] = parts[r][:, :tile_columns(b.nx, r, world), :]
    return full.reshape(-1), iang


_GATHER_OK = True


def _gather_tiles(tile, rank: int, world: int, dst: int, group):
    """Tiles of all ranks on rank dst (list), None elsewhere.  A gather to one root is all the path
    needs (each peer has its own xGMI link to the root); should the backend refuse `gather`, every
    rank falls back to `all_gather` -- the tiles are small -- and keeps doing so."""
    global _GATHER_OK
    import torch
    import torch.distributed as dist

    if _GATHER_OK:
        try:
            parts = [torch.empty_like(tile) for _ in range(world)] if rank == dst else None
            dist.gather(tile, gather_list=parts, dst=dst, group=group)
            return parts
        except (RuntimeError, NotImplementedError):
            _GATHER_OK = False
    parts = [torch.empty_like(tile) for _ in range(world)]
    dist.all_gather(parts, tile, group=group)
    return parts if rank == dst else None

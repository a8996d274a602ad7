"""Multi-GPU forms of the path.

One process, all devices: behind the C ABI (rt_hip_multi_image_loop, include/rt_hip.h; Python:
backend.multi_image_loop / create_image(p, "hip-multigpu")).  This module is the other form:

shard / Assembler (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
    "gloo" in the CPU tests) -- what bench.py runs at N > 1
    ASE    : pixel columns are dealt round-robin to ranks (problem.shard_columns); a rank's image is a
             compact tile [ny][nx_local][nv]; assembly is ONE RCCL gather of (tile | I_ang) buffers to
             rank 0, one interleave copy and one sum over the I_ang parts.  No other exchange.
    seeded : source columns are dealt the same way, every rank holds a full private image, assembly
             is ONE sum-reduce of (image | I_ang) (the analogue of intensity_step_struct::sum_reduce,
             src/RayTraceStructures.cpp:1603-1646).
"""
from __future__ import annotations

import numpy as np

from .problem import Problem, shard_columns


# --------------------------------------------------------------------------- one process per GPU
def shard(problem: Problem, rank: int, world: int) -> Problem:
    return shard_columns(problem, rank, world)


def tile_columns(nx: int, rank: int, world: int) -> int:
    return len(range(rank, nx, world))


class Assembler:
    """Per-rank output buffer and the ONE collective per image that assembles it on rank `dst`.

    Every rank traces its shard into `buffer` = (tile | pad | I_ang) -- `image` and `iang` are views of
    it, handed to rt_hip_plan_run as device pointers -- and `assemble()` then issues
      ASE    : a gather of the buffers to rank dst (each peer over its own xGMI link), followed on dst
               by one interleave copy of the pixel-column tiles and one sum over the I_ang tails;
      seeded : a sum-reduce of the buffers (full image | I_ang) to rank dst.
    Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over gloo."""

    def __init__(self, problem: Problem, rank: int, world: int, device=None, group=None, dst: int = 0,
                 via_host: bool = False):
        import torch

        b = problem.beam
        self.problem, self.rank, self.world, self.group, self.dst = problem, rank, world, group, dst
        self.seeded = problem.seed is not None
        self.n_ang = b.na * b.nb
        self.ncol = tile_columns(b.nx, rank, world) if not self.seeded else b.nx
        self.ncol_max = tile_columns(b.nx, 0, world) if not self.seeded else b.nx
        self.n_tile = b.ny * self.ncol * b.nv
        self.n_tile_max = b.ny * self.ncol_max * b.nv
        dev = device if device is not None else torch.device("cpu")
        self.buffer = torch.zeros(self.n_tile_max + self.n_ang, dtype=torch.float64, device=dev)
        self.image = self.buffer[:self.n_tile]
        self.iang = self.buffer[self.n_tile_max:]
        # via_host: device buffers, collectives on host copies (rehearsal of the N-rank path over gloo on a
        # box whose ranks share one GPU; RCCL takes the device buffers directly)
        self.via_host = via_host and dev.type != "cpu"
        cdev = torch.device("cpu") if self.via_host else dev
        # the gather's receive buffer (world x (tile | I_ang) on rank dst) is allocated by the first assemble():
        # a run that leaves its tiles on their GPUs (bench.py --no-assemble, config 5) never pays for it
        self.recv = None
        self._recv_dev = cdev
        self.last_collective = None  # what the last assemble() ran, for describe()

    def describe(self) -> str:
        n = (self.n_tile_max + self.n_ang) * 8
        if self.seeded:
            plan = f"reduce(sum, f64) of {n} B (image | I_ang) to rank {self.dst}"
        else:
            plan = (f"gather of {n} B (pixel-column tile | I_ang) per rank to rank {self.dst}, one interleave copy, "
                    f"one I_ang sum")
        ran = self.last_collective or "none yet"
        return f"{plan}; last run: {ran}"

    def assemble(self, buffer=None):
        """Returns (image, I_ang) flat tensors on rank dst, (None, None) elsewhere."""
        import torch
        import torch.distributed as dist

        buf = self.buffer if buffer is None else buffer
        b = self.problem.beam
        if self.world == 1:
            return buf[:self.n_tile], buf[self.n_tile_max:]
        if self.via_host:
            buf = buf.cpu()
        backend = dist.get_backend(self.group) + (" via host copies" if self.via_host else "")
        if self.seeded:
            dist.reduce(buf, dst=self.dst, op=dist.ReduceOp.SUM, group=self.group)
            self.last_collective = f"torch.distributed.reduce(SUM) over {backend}, world {self.world}"
            return (buf[:self.n_tile], buf[self.n_tile_max:]) if self.rank == self.dst else (None, None)
        if self.rank == self.dst and self.recv is None:
            self.recv = torch.empty((self.world, self.n_tile_max + self.n_ang), dtype=torch.float64, device=self._recv_dev)
        parts = list(self.recv.unbind(0)) if self.rank == self.dst else None
        dist.gather(buf, gather_list=parts, dst=self.dst, group=self.group)
        self.last_collective = f"torch.distributed.gather over {backend}, world {self.world}"
        if self.rank != self.dst:
            return None, None
        iang = self.recv[:, self.n_tile_max:].sum(0)
        if b.nx % self.world == 0:
            # equal tiles: column i = r + world * c  <=>  [rank][ny][c][k] -> [ny][c][rank][k], one copy
            full = self.recv[:, :self.n_tile_max].view(self.world, b.ny, self.ncol_max, b.nv).permute(1, 2, 0, 3).reshape(-1)
        else:
            out = torch.empty((b.ny, b.nx, b.nv), dtype=buf.dtype, device=buf.device)
            for r in range(self.world):
                nc = tile_columns(b.nx, r, self.world)
                out[:, r::self.world, :] = self.recv[r, :b.ny * nc * b.nv].view(b.ny, nc, b.nv)
            full = out.reshape(-1)
        return full, iang


def assemble(problem: Problem, tile_image, tile_iang, rank: int, world: int, group=None, dst: int = 0):
    """Assemble the final image on rank `dst` from per-rank results held in separate tensors
    (tile_image / tile_iang: device tensors under RCCL, CPU tensors under gloo).
    Returns (image, I_ang) as flat tensors on rank dst, (None, None) elsewhere."""
    a = Assembler(problem, rank, world, device=tile_image.device, group=group, dst=dst)
    a.image.copy_(tile_image.reshape(-1))
    a.iang.copy_(tile_iang.reshape(-1))
    img, ang = a.assemble()
    if img is None:
        return None, None
    return img.clone(), ang.clone()

"""Two ranks sharing the one GPU of the test box: each rank traces its pixel-column
shard with the HIP plan, the tiles are assembled with the same code bench.py uses
(`multigpu.assemble`), here over gloo on host tensors (RCCL needs one device per
rank; the driver exercises that at N = 2/4/8).  The result must equal the
single-rank image."""
import importlib
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
rt = importlib.import_module("raytrace-miniapp_amd")
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, scale, out_path):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    rtw = importlib.import_module("raytrace-miniapp_amd")
    mg = importlib.import_module("raytrace-miniapp_amd.multigpu")
    be = importlib.import_module("raytrace-miniapp_amd.backend")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = rtw.scale_problem(rtw.datfile.load(ROOT / "tests" / "golden" / f"{name}.dat.xz"), scale)
    mine = mg.shard(full, rank, world)
    with be.Plan(mine, device=0) as plan:
        res = plan.set_ray_grid().run().fetch()
    assert res["failure_code"] == 0
    img, ang = mg.assemble(full, torch.from_numpy(res["image"]), torch.from_numpy(res["I_ang"]), rank, world)
    if rank == 0:
        np.savez(out_path, image=img.numpy(), I_ang=ang.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,scale", [("ASE_small", 1.0), ("seed_small", 0.05)])
def test_two_rank_shards_assemble_to_the_single_rank_image(tmp_path, hip, name, scale):
    import torch.multiprocessing as mp
    out = tmp_path / "r0.npz"
    mp.spawn(_worker, args=(2, _free_port(), name, scale, str(out)), nprocs=2, join=True)
    got = np.load(out)
    full = rt.scale_problem(rt.datfile.load(ROOT / "tests" / "golden" / f"{name}.dat.xz"), scale)
    with hip.Plan(full) as plan:
        want = plan.set_ray_grid().run().fetch()
    for key in ("image", "I_ang"):
        assert np.linalg.norm(got[key] - want[key]) <= 1e-12 * np.linalg.norm(want[key])
    assert np.linalg.norm(want["image"]) > 0

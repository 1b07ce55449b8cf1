"""N>1 path on CPU: two gloo ranks own disjoint Z-order tiles, render them (CPU checker as the stand-in
renderer -- the HIP path needs a GPU) and reduce the float framebuffer to rank 0 exactly as bench.py does."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_binding as ob
    from pearray_amd import distributed, scene, tiling
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp = 64, 48, 3
    o = ob.OracleScene(scene.cornell_box(W, H, spp=spp))
    tiles = tiling.tiles_for_rank(W, H, rank, world, tile=16)
    o.set_tiles(tiles)
    o.render(spp, threads=2)
    xyz, smp, _ = o.output()
    fx, fs = torch.from_numpy(xyz.copy()), torch.from_numpy(smp.astype(np.int32))
    distributed.reduce_framebuffer(fx, fs, dst=0)
    total = distributed.sum_scalar(float(tiling.owned_pixel_count(tiles) * spp))
    t_max = distributed.max_scalar(0.5 + rank)
    per_rank = distributed.gather_scalars(10.0 + rank)      # bench.py's per-rank render times (config.rank_render_ms_per_step)
    if rank == 0:
        np.savez(out_path, xyz=fx.numpy(), smp=fs.numpy(), total=total, tmax=t_max, per_rank=np.array(per_rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharding_reduces_to_the_full_frame(tmp_path):
    out = str(tmp_path / "reduced.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from pearray_amd import scene
    whole = ob.OracleScene(scene.cornell_box(64, 48, spp=3))
    whole.render(3, threads=4)
    xyz, smp, _ = whole.output()
    got = np.load(out)
    assert np.array_equal(got["xyz"], xyz) and np.array_equal(got["smp"], smp.astype(np.int32))
    assert got["total"] == 64 * 48 * 3 and got["tmax"] == 1.5 and got["per_rank"].tolist() == [10.0, 11.0]

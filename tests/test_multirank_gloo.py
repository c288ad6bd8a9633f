"""World-size-2 test of the N>1 path on CPU (gloo): islands are sharded across ranks with NO data-path
collective, every rank steps its own block, and the integrated state is all-gathered - the result must equal
the un-sharded world bit for bit. The CPU oracle stands in for the stepper here (no GPU in this container);
the sharding / gather logic (edyn_amd.parallel, edyn_amd.scenes) is exactly what bench.py runs over RCCL."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world_size, port, steps, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from edyn_amd import scenes
    from edyn_amd.parallel import shard_range, gather_state, pack_state
    from oracle import binding as ob
    total_sites = 6
    first, count = shard_range(total_sites, rank, world_size)
    scene = scenes.mini_piles(3, 2, first_site=first, num_sites=count)
    w = ob.World(vel_iters=10, order=ob.ORDER_COLOURED)
    w.add_bodies(scene)
    counts = [64 * shard_range(total_sites, r, world_size)[1] for r in range(world_size)]
    gathered = None
    for _ in range(steps):
        w.step(1)
        pos, orn, lv, av = w.get_state()
        local = torch.from_numpy(pack_state(pos[1:], orn[1:], lv[1:], av[1:]))   # dynamic bodies only (the plane is replicated)
        gathered = gather_state(local, counts)
    if rank == 0:
        np.save(out_path, gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_islands_match_single_world(tmp_path):
    from edyn_amd import scenes
    from edyn_amd.parallel import pack_state
    from oracle import binding as ob
    steps = 8
    out = str(tmp_path / "gathered.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, steps, out), nprocs=2, join=True)
    got = np.load(out)
    w = ob.World(vel_iters=10, order=ob.ORDER_COLOURED)
    w.add_bodies(scenes.mini_piles(3, 2))
    w.step(steps)
    pos, orn, lv, av = w.get_state()
    ref = pack_state(pos[1:], orn[1:], lv[1:], av[1:])
    assert got.shape == ref.shape == (6 * 64, 13)
    assert np.array_equal(got, ref)
    assert w.get_stats()["num_islands"] == 6

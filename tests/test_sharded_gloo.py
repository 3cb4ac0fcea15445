"""CPU suite: the multi-rank host path (slam_amd/sharded.py + slam_amd/filter.py) under gloo,
world sizes 2 and 4, with the oracle-backed test double standing in for the GPU library. The
sharded run must reproduce the one-rank run bit for bit: global slot numbering, the
all-gathered scores and pose blocks, the redundant weights/resample on every rank; in
per-particle-map mode (FastSLAM, S3P:52) also every map slab after the fused update and the
cross-rank migration of pfh_migration_plan, executed over gloo send/recv."""
import zlib

import pytest
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _scenario():
    from slam_amd import synth
    dims = (48, 48, 32)
    Pg, M, steps = 96, 150, 4
    sc = synth.scene(dims, Pg, M, seed=4)
    rng = np.random.default_rng(9)
    noise = rng.normal(0, 0.02, (steps, Pg, 4)).astype(np.float32)
    u = rng.random((steps, Pg))
    return dims, Pg, steps, sc, noise, u


def _run(rank, world, port, q, per_particle=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    from oracle import binding as O
    from oracle_backend import OracleHandle
    from slam_amd.filter import ParticleMapStep
    from slam_amd.sharded import ShardedBackend
    dims, Pg, steps, sc, noise, u = _scenario()
    P = Pg // world
    lo = rank * P
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    h = OracleHandle(P, Pg, lo, grid=O.ref_grid(*dims), per_particle=per_particle, world=sc["grid"],
                     dist=dist if world > 1 else None)
    be = ShardedBackend(h, dist) if world > 1 else h
    st = ParticleMapStep(be, P, Pg, lo)
    st.set_state(sc["x"][lo:lo + P], sc["y"][lo:lo + P], sc["z"][lo:lo + P], sc["yaw"][lo:lo + P],
                 np.full(Pg, 1.0 / Pg))
    log = []
    for t in range(steps):
        slots = st.step(sc["beams"], (5.02, 4.99, 1.5, 0.01), (1.2, 1.2, 0.8, 0.35),
                        noise[t, lo:lo + P], u[t])
        log.append((slots.copy(), st.best, st.feedback.copy()))
    x, y, z, yaw = be.get_poses()
    w, _ = be.get_weights()
    crcs = [zlib.crc32(h.maps[i].tobytes()) for i in range(h.maps.shape[0])] if per_particle else []
    q.put((rank, world, np.stack([x, y, z, yaw]), w, log, crcs, getattr(h, "last_transfers", None)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _compare(world, per_particle, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p1 = ctx.Process(target=_run, args=(0, 1, port, q, per_particle)); p1.start()
    single = q.get(timeout=180); p1.join(60)
    procs = [ctx.Process(target=_run, args=(r, world, port + 1, q, per_particle)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    poses = np.concatenate([r[2] for r in res], axis=1)
    assert np.array_equal(poses, single[2])
    for r in res:
        assert np.array_equal(r[3], single[3])                    # whole weight vector on every rank
        for (s_a, b_a, f_a), (s_b, b_b, f_b) in zip(r[4], single[4]):
            assert np.array_equal(s_a, s_b) and b_a == b_b and np.array_equal(f_a, f_b)
    if per_particle:
        assert sum((r[5] for r in res), []) == single[5]          # every slab, after fuse and migration
        assert any(r[6] and r[6][0] > 0 for r in res)             # slabs did cross ranks
    return res


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_equal_one_rank(world):
    _compare(world, False, 29611 + 10 * world)


@pytest.mark.parametrize("world", [2, 4])
def test_per_particle_maps_migrate_between_ranks(world):
    _compare(world, True, 29711 + 10 * world)

"""GPU test (-m gpu) of the RCCL map migration inside pf_resample (S3P:805-812 across ranks: per-particle map
slabs whose ancestor lives on another rank move by ncclSend / ncclRecv groups), executed on the ONE GPU a test
box has: a world-1 communicator and the test switch SLAM_PF_VIRTUAL_SHARDS, which makes the handle treat its
shard as V equal blocks of slots ("virtual shards") that exchange their slabs exactly as ranks do -- the plan of
every block from pfh_migration_plan, the groups cut every 512 transfer numbers, the slab pointer arithmetic,
ncclSend and ncclRecv with this rank as the peer. Before this test those lines had never run at all (VERDICT
round 2, item 6). Every slab is checked against the permutation the returned slot_anc[] defines."""
import os

import numpy as np
import pytest

from slam_amd import capi

pytestmark = pytest.mark.gpu


def _run(P, dims, V, seed, heavy):
    """Resample once on per-particle maps that are all different; returns (slot_anc, maps before, maps after)."""
    if V:
        os.environ["SLAM_PF_VIRTUAL_SHARDS"] = str(V)
    else:
        os.environ.pop("SLAM_PF_VIRTUAL_SHARDS", None)
    h = capi.Handle(capi.default_config(n_particles=P, map_mode=capi.PF_MAP_PER_PARTICLE, grid_x=dims[0], grid_y=dims[1],
                                        grid_z=dims[2], local_x=0, local_y=0, local_z=0, max_beams=64))
    try:
        h.comm_init(capi.comm_unique_id(), 0, 1)
        rng = np.random.default_rng(seed)
        n = dims[0] * dims[1] * dims[2]
        before = rng.integers(0, 101, (P, n), dtype=np.uint8)
        for i in range(P):
            h.write_map(i, before[i])
        w = rng.random(P) ** 4
        w[rng.choice(P, heavy, replace=False)] *= 50.0          # a few ancestors with many children, most slots replaced
        h.set_weights(w / w.sum())
        slots, best, nohit, amb = h.resample(rng.random(P))
        assert nohit == 0
        after = np.stack([h.read_map(i) for i in range(P)])
        return slots, before, after
    finally:
        h.close()
        os.environ.pop("SLAM_PF_VIRTUAL_SHARDS", None)


@pytest.mark.parametrize("P,V", [(64, 2), (96, 4), (1536, 2)])
def test_map_migration_through_rccl_to_self(P, V):
    dims = (16, 16, 8) if P > 256 else (32, 32, 16)
    slots0, before0, after0 = _run(P, dims, 0, 7, 3)            # the one-GPU path: every slab copied by k_apply_maps
    assert np.array_equal(after0, before0[slots0])
    slots, before, after = _run(P, dims, V, 7, 3)               # the same step with the slabs of other blocks sent and received
    assert np.array_equal(slots, slots0) and np.array_equal(before, before0)
    Pv = P // V
    moved = int(np.sum((slots // Pv) != (np.arange(P) // Pv)))
    assert moved > (512 if P > 1000 else 8)                     # (1536 slots: more than one 512-transfer group)
    assert np.array_equal(after, before[slots])
    # the plan itself: every virtual shard's lists, played through per-pair FIFO order, name exactly these transfers
    total = 0
    for v in range(V):
        n_tr, sends, recvs = capi.migration_plan(slots, Pv, v)
        total = n_tr
        for dst, peer, seq in recvs:
            assert slots[v * Pv + dst] // Pv == peer
    assert total == moved

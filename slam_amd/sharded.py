"""One rank of a multi-GPU filter: particles sharded in contiguous equal blocks, one process
per GPU, torch.distributed (backend "nccl" = RCCL over xGMI) for the single exchange step.

Per step and rank:
  measure   local shard only, no communication (S3P:859-908 has no cross-particle dependence)
  C1        all-gather of the (sum_match, num_match) pairs, 8 B per particle
  weights   every rank computes min / weights / sum / CDF of ALL particles redundantly with the
            same fixed-shape reductions, so all ranks agree bit for bit (S3P:920-937)
  C2        all-gather of the 72-B pose blocks (shared-grid mode: the whole particle state), from
            which each rank picks the ancestors of its own slots (S3P:805-812)
  resample  redundantly on every rank; slot numbering is global, so slot_anc[] is identical to
            the one-GPU run.
The shared grid is replicated; every rank applies the identical update_map.
Per-particle maps (FastSLAM mode) live with their particle: pf_resample copies the slabs whose
ancestor is on the same rank and moves the others with RCCL send/receive groups itself, which
needs the library's own communicator (native=True on the nccl backend).
"""
import numpy as np


class ShardedBackend:
    def __init__(self, handle, dist, group=None, native=True):
        """handle: slam_amd.capi.Handle created with n_particles = shard size,
        n_particles_global, particle_offset = rank * shard size. dist: torch.distributed.
        native: let the library do the two all-gathers itself with RCCL on its own stream
        (pf_comm_init; torch.distributed then only carries the communicator id at set-up). Used
        when the backend is nccl and every rank managed to create its communicator; otherwise the
        collectives go through torch.distributed, as they do on gloo."""
        import torch
        self.torch = torch
        self.h = handle
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.P = handle.P
        self.Pg = handle.Pg
        if self.P * self.world != self.Pg:
            raise ValueError("shards must be equal: n_particles * world != n_particles_global")
        # zero-copy views of the handle's device buffers (fixed addresses): made once, the
        # __cuda_array_interface__ import costs tens of microseconds per call
        s = self._scores = handle.scores_tensor()
        p = self._poses = handle.poses_tensor()
        # rank blocks concatenated along dim 0: [world*P, 2] and [world*9, P]
        self._scores_all = torch.empty((self.world * s.shape[0], s.shape[1]), dtype=s.dtype,
                                       device=s.device)
        self._poses_all = torch.empty((self.world * p.shape[0], p.shape[1]), dtype=p.dtype,
                                      device=p.device)
        self.native = False
        if native and hasattr(handle, "comm_init") and dist.get_backend(group) == "nccl":
            self.native = self._init_native()

    def _init_native(self):
        from . import capi
        torch, dist = self.torch, self.dist
        ok = 1
        try:
            box = [capi.comm_unique_id() if self.rank == 0 else None]
        except Exception:
            box, ok = [None], 0
        src = dist.get_global_rank(self.group, 0) if self.group else 0
        dist.broadcast_object_list(box, src=src, group=self.group)
        if box[0] is None:
            return False                      # rank 0 has no RCCL: every rank sees that
        try:
            self.h.comm_init(box[0], self.rank, self.world)
        except Exception:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self._scores_all.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    # pass-through (local, no communication)
    def set_beams(self, *a): self.h.set_beams(*a)
    def measure(self, add): self.h.measure(add)
    def predict(self, x, y, z, yaw): self.h.predict(x, y, z, yaw)
    def get_poses(self): return self.h.get_poses()
    def set_weights(self, w): self.h.set_weights(w)
    def get_weights(self): return self.h.get_weights()
    def set_scan(self, *a): self.h.set_scan(*a)
    def update_map(self, *a): self.h.update_map(*a)

    def _flush_poses(self):
        # pf_predict_device may have left a pose block in place (one-GPU handle): pf_poses_device
        # makes the library copy it into the buffer the cached view points at
        flush = getattr(self.h, "poses_device", None)
        if flush is not None:
            flush()

    def _sync_device(self, t):
        if t.is_cuda:
            self.torch.cuda.current_stream(t.device).synchronize()

    def _before_collective(self):
        # handle on its own stream: its work must be complete before the collective reads it;
        # handle on torch's current stream (pf_set_stream): stream order already says so
        if not getattr(self.h, "external_stream", False):
            self.h.synchronize()

    def _after_collective(self, t):
        if not getattr(self.h, "external_stream", False):
            self._sync_device(t)

    def weights(self):
        if self.native:
            return self.h.weights(None)       # C1 inside the library
        self._before_collective()
        self.dist.all_gather_into_tensor(self._scores_all, self._scores, group=self.group)   # C1
        self._after_collective(self._scores_all)
        self.h.weights(self._scores_all.data_ptr())

    def resample(self, u=None, want_slots=True):
        if self.native:
            return self.h.resample(u, None, want_slots)   # C2 inside the library
        self._flush_poses()
        self._before_collective()
        self.dist.all_gather_into_tensor(self._poses_all, self._poses, group=self.group)     # C2
        self._after_collective(self._poses_all)
        return self.h.resample(u, self._poses_all.data_ptr(), want_slots)

    def resample_async(self, u=None, want_slots=False):
        if self.native:
            return self.h.resample_async(u, None, want_slots)
        self._flush_poses()
        self._before_collective()
        self.dist.all_gather_into_tensor(self._poses_all, self._poses, group=self.group)     # C2
        self._after_collective(self._poses_all)
        return self.h.resample_async(u, self._poses_all.data_ptr(), want_slots)

    def resample_result(self):
        return self.h.resample_result()

    def best_pose(self, best):
        """(x, y, z, yaw) of global slot `best` after resampling, on every rank."""
        r, i = divmod(int(best), self.P)
        t = self.torch.empty(4, dtype=self.torch.float64, device=self._poses_all.device)
        if r == self.rank:
            x, y, z, yaw = self.h.get_poses()
            t.copy_(self.torch.tensor([x[i], y[i], z[i], yaw[i]], dtype=self.torch.float64))
        self.dist.broadcast(t, src=self.dist.get_global_rank(self.group, r) if self.group else r,
                            group=self.group)
        return tuple(float(v) for v in t.cpu().numpy())

"""TEST DOUBLE: an object with slam_amd.capi.Handle's methods whose loops are the CPU oracle.

Lives under tests/ on purpose. It lets the CPU suite drive the product's host-side code
(slam_amd/filter.py, slam_amd/sharded.py) against the golden vectors and under gloo without a
GPU. The product never imports it; the product's only backend is the HIP library.
"""
import ctypes as C

import numpy as np
import torch

from oracle import binding as O


class OracleHandle:
    def __init__(self, P, Pg=None, offset=0, grid=None, per_particle=True, world=None, prior=50, dist=None):
        self.dist = dist          # torch.distributed of a sharded run: carries the map slabs between ranks
        self.P = P
        self.Pg = Pg if Pg is not None else P
        self.offset = offset
        self.grid = grid or O.ref_grid()
        self.elements = self.grid.gx * self.grid.gy * self.grid.gz
        self.per_particle = per_particle
        n_maps = P if per_particle else 1
        self.maps = np.full((n_maps, self.elements), prior, np.uint8)
        if world is not None:
            self.maps[:] = np.asarray(world, np.uint8).reshape(1, -1)
        self.pose = np.zeros((9, P), np.float64)
        self.pose[0] = 7.5; self.pose[1] = 7.5; self.pose[2] = 1.5
        self.w = np.full(self.Pg, 1.0 / self.Pg)
        self.logw = np.zeros(self.Pg, np.float32)
        self.best = 0
        self.scores = np.zeros((P, 2), np.int32)
        self.external_stream = False

    # state
    def predict(self, x, y, z, yaw):
        self.pose[0] = x; self.pose[1] = y; self.pose[2] = z; self.pose[3] = yaw

    def get_poses(self):
        return [self.pose[k].copy() for k in range(4)]

    def set_weights(self, w):
        self.w[:] = w

    def get_weights(self):
        return self.w.copy(), self.logw.copy()

    def read_map(self, i=0):
        return self.maps[i if self.per_particle else 0].copy()

    # measurement
    def set_beams(self, bx, by, bz, bv):
        self.beams = tuple(np.array(a) for a in (bx, by, bz, bv))

    def measure(self, add_points):
        fuse = bool(add_points) and self.per_particle
        s, c = O.measure(self.grid, self.pose[0].copy(), self.pose[1].copy(), self.pose[2].copy(),
                         self.pose[3].copy(), self.maps.reshape(-1),
                         self.elements if self.per_particle else 0, *self.beams, fuse)
        self.scores[:, 0] = s; self.scores[:, 1] = c

    def get_scores(self):
        return self.scores[:, 0].copy(), self.scores[:, 1].copy()

    def scores_tensor(self):
        return torch.from_numpy(self.scores)

    def poses_tensor(self):
        return torch.from_numpy(self.pose)

    def synchronize(self):
        pass

    @staticmethod
    def _from_ptr(ptr, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        buf = (C.c_char * n).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def weights(self, scores_global_ptr=None):
        if scores_global_ptr:
            sc = self._from_ptr(scores_global_ptr, (self.Pg, 2), np.int32)
        else:
            assert self.Pg == self.P
            sc = self.scores
        _, self.logw = O.weights(self.w, sc[:, 0].copy(), int(sc[-1, 1]))

    def resample(self, u, poses_global_ptr=None, want_slots=True):
        d, sa, best, nohit = O.resample(self.w, u, self.best)
        self.best = best
        if poses_global_ptr:
            world = self.Pg // self.P
            allp = self._from_ptr(poses_global_ptr, (world, 9, self.P), np.float64)
            new = self.pose.copy()
            for i in range(self.P):
                a = sa[self.offset + i]
                if a != self.offset + i:
                    new[:, i] = allp[a // self.P, :, a % self.P]
            self.pose[...] = new      # in place: the buffer address is fixed, like the device one
            if self.per_particle:
                self._migrate_maps(sa)
        else:
            rec = np.ascontiguousarray(self.pose.T)
            O.permute(sa, rec)
            self.pose[...] = rec.T
            if self.per_particle:
                O.permute(sa, self.maps)
        lw = self.logw.copy()
        O.permute(sa, lw)
        self.logw = lw
        return sa, best, nohit, 0

    def _migrate_maps(self, sa):
        """S3P:810 across ranks, the way pf_resample does it inside the library: slabs whose ancestor is
        local are copied, the others travel by the transfer lists of pfh_migration_plan (here over
        torch.distributed send/recv instead of RCCL: receives posted first, one tag per transfer)."""
        from slam_amd import capi
        rank = self.offset // self.P
        old = self.maps.copy()
        for i in range(self.P):
            a = int(sa[self.offset + i])
            if a != self.offset + i and self.offset <= a < self.offset + self.P:
                self.maps[i] = old[a - self.offset]
        total, sends, recvs = capi.migration_plan(sa, self.P, rank)
        self.last_transfers = (total, len(sends), len(recvs))
        if not sends and not recvs:
            return
        bufs, work = [], []
        for dst, peer, seq in recvs:
            t = torch.empty(self.elements, dtype=torch.uint8)
            bufs.append((dst, t))
            work.append(self.dist.irecv(t, src=peer, tag=int(seq)))
        for src, peer, seq in sends:
            work.append(self.dist.isend(torch.from_numpy(old[src].copy()), dst=peer, tag=int(seq)))
        for wk in work:
            wk.wait()
        for dst, t in bufs:
            self.maps[dst] = t.numpy()


class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class OracleLocHandle:
    """Test double of the PF_MAP_SHARED handle that holds the binary localisation grid (S3P:64) and
    its keyframe ring: the methods slam_amd.frontend.FrameLoop calls, loops = the CPU oracle."""

    def __init__(self, dims=(200, 200, 80), resolution=0.05):
        self.dims, self.resolution = dims, resolution
        self.grid = np.zeros(dims, np.uint8)
        self.ring = O.RingState(dims, resolution)

    def depth_to_cloud(self, image, width, height, pool=4, focal=385.0 / 4.0):
        return O.depth_to_cloud(image, width, height, pool, focal)

    def ring_paint(self):
        self.ring.paint(self.grid)

    def scan_vote(self, points, pos, zyx, **kw):
        return O.scan_vote(self.grid, self.resolution, points, pos, zyx, **kw)

    def ring_update(self, cloud, pos_scan, zyx):
        return self.ring.update(self.grid, cloud, pos_scan, zyx)

    def ring_state(self, with_idx=False):
        r = self.ring
        return (r.count.copy(), r.life.copy(), np.array(r.r.pos_update[:]), np.array(r.r.euler_update[:]),
                r.idx.reshape(r.r.n_slots, r.r.cap).copy() if with_idx else None)

    def read_map(self, i=0):
        return self.grid.reshape(-1).copy()

    def write_map(self, i, src):
        self.grid.reshape(-1)[:] = src


class OracleMapper:
    """Test double of the handle whose int8 local map (S3P:65) build_local_map() writes."""

    def __init__(self, dims=(200, 200, 80), resolution=0.05):
        self.cfg = _Cfg(local_x=dims[0], local_y=dims[1], local_z=dims[2])
        self.resolution = resolution
        self.local = np.zeros(dims[0] * dims[1] * dims[2], np.int8)

    def set_scan(self, points, ray_idx):
        self.points = np.array(points, np.float64).reshape(-1, 3)
        self.rays = np.array(ray_idx, np.int32)

    def update_map(self, target, params, pos, zyx):
        c = self.cfg
        cfg = O.UpdateCfg(c.local_x, c.local_y, c.local_z, self.resolution, 1, params.hit_inc, params.hit_clamp,
                          params.miss_inc, params.miss_clamp, params.max_range)
        O.update_map(cfg, self.local, pos, zyx, self.points, self.rays)

    def write_local_map(self, src):
        self.local[:] = np.asarray(src, np.int8).reshape(-1)

    def read_local_map(self):
        return self.local.copy()

    def reset_local_map(self):
        self.local[:] = 0

"""Host-driver mirror of the reference's particle_map() (S3P:827-962).

The caller-side lines stay on the host, as in the reference: the snapshot that yields the beam
list, the motion model with its noise, the add_points policy and the correction feedback. The
loops go to the GPU through the C ABI (include/slam_pf.h) in the reference's own order:
measure -> predict -> weights -> resample.

`backend` is a slam_amd.capi.Handle (one GPU) or a slam_amd.sharded.ShardedBackend (one rank
of a multi-GPU job). Nothing here computes a result the reference computes in its loops.
"""
import numpy as np

from . import capi

LOCAL_RESET = (5.0, 5.0, 1.5, 0.0)        # S3P:841-844


class ParticleMapStep:
    def __init__(self, backend, n_local, n_global=None, offset=0):
        self.be = backend
        self.P = n_local
        self.Pg = n_global if n_global is not None else n_local
        self.offset = offset
        # host mirror of the local shard's poses (S3P:47-50)
        self.x = np.full(self.P, 7.5); self.y = np.full(self.P, 7.5)
        self.z = np.full(self.P, 1.5); self.yaw = np.zeros(self.P)
        self.add_points = 1                                   # S3P:823
        self.last_update = np.array([7.5, 7.5, 1.5, 0.0])     # S3P:824-825
        self.best = 0                                         # S3P:756
        self.feedback = np.zeros(4)                           # S3P:60-61
        self.last_nohit = 0
        self.last_ambiguous = 0

    def set_state(self, x, y, z, yaw, w_global=None):
        self.x[:] = x; self.y[:] = y; self.z[:] = z; self.yaw[:] = yaw
        self.be.predict(self.x, self.y, self.z, self.yaw)
        if w_global is not None:
            self.be.set_weights(w_global)

    def step(self, beams, local_pose, global_pose, noise, u):
        """One particle_map() call.

        beams: (bx, by, bz, bv) from the caller's snapshot (S3P:833-846, 868-877).
        local_pose: (x, y, z, yaw) of the local frame at snapshot time (S3P:835-836).
        global_pose: (x, y, z, yaw) dead-reckoned pose at snapshot time (S3P:837-838).
        noise: float32 [n_local, 4], the normal draws of S3P:911-914 for the LOCAL shard.
        u: float64 [n_global], the resampling thresholds (S3P:773), the same on every rank.
        """
        be = self.be
        be.set_beams(*beams)
        be.measure(self.add_points)                                        # S3P:859-908
        dl = (local_pose[0] - 5, local_pose[1] - 5, local_pose[2] - 1.5)   # S3P:911-913
        capi.host_predict(self.x, self.y, self.z, self.yaw, dl, local_pose[3], noise)
        be.predict(self.x, self.y, self.z, self.yaw)
        be.weights()                                                       # S3P:920-937
        self.add_points = capi.add_points_policy(global_pose, self.last_update)   # S3P:941-953
        slots, best, nohit, amb = be.resample(u)                           # S3P:954
        self.best, self.last_nohit, self.last_ambiguous = best, nohit, amb
        if self.Pg == self.P:
            for a in (self.x, self.y, self.z, self.yaw):                   # S3P:805-812
                capi.permute(slots, a)
            bp = (self.x[best], self.y[best], self.z[best], self.yaw[best])
        else:
            self.x, self.y, self.z, self.yaw = be.get_poses()
            bp = be.best_pose(best)
        self.feedback = np.array(bp) - np.asarray(global_pose, np.float64)  # S3P:956-959
        return slots

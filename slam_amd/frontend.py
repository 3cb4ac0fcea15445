"""Host-driver mirror of the reference's Slam::callback() (S3P:168-205): what the caller keeps on
the host per depth frame — the members of class Slam that carry the scan pose, the vote's last
hypothesis, the IMU angle bookkeeping, the local frame's pose and the dead-reckoned global pose —
while the loops go to the GPU through the C ABI (include/slam_pf.h) in the reference's order:

    pf_depth_to_cloud                  resize_depth_to_array + img_to_cloud      S3P:173-175, 562-612
    pf_ring_paint, pf_scan_vote        scan_localization                         S3P:300-376
    pfh_vote_hypothesis / _epilogue                                              S3P:366-404
    pf_ring_update                     scan_localization_map_update              S3P:419-529
    pf_set_scan, pf_update_map         build_local_map                           S3P:621-704
    pfh_update_realtime_pos            update_realtime_pos                       S3P:712-728

With device_cloud=True the cloud (S3P:126) never leaves the GPU: the pf_*_device forms of the same calls work on
one device buffer, the two handles share `loc`'s stream, and per frame only the point count, the vote's sums and
the ring's count come back to the host (results identical, tests/test_gpu_frontend.py).

`loc` is a PF_MAP_SHARED handle holding the binary localisation grid (S3P:64) with its keyframe
ring; `mapper` is the handle whose int8 local map (S3P:65) build_local_map() writes — the filter's
own handle in the reference's layout. Both are slam_amd.capi.Handle objects (or, in the CPU suite,
test doubles with the same methods). Nothing here computes a result the reference computes in its
loops. driver/step_loop.cpp is the same caller in C++.
"""
import numpy as np

from . import capi

VOTE = dict(n_yaw=5, yaw_k0=-6, yaw_kstep=3, n_t=5, length=0.25)      # S3P:275-279, 328-348
N_SCAN_POINTS = 500                                                  # S3P:261
N_RAYS = 2400                                                        # S3P:668


class FrameLoop:
    def __init__(self, loc, mapper, width=640, height=480, pool=4, focal=385.0 / 4.0, resolution=0.05,
                 map_size=(10, 10, 4), replay=False, device_cloud=False):
        self.loc, self.mapper = loc, mapper
        self.device_cloud = bool(device_cloud)
        # replay: the index draws keep the reference's inclusive upper bound n (S3P:317, 666), for seeds known not to
        # draw it (the golden fixtures); default: draws in [0, n-1], which any seed serves
        self.replay = replay
        self.width, self.height, self.pool, self.focal = width, height, pool, focal
        self.resolution = resolution
        half = [map_size[0] // 2, map_size[1] // 2, map_size[2] // 2]   # S3P:269-271: integer halves
        # members of class Slam
        self.pos_scan = np.array(half, np.float64)                   # S3P:282
        self.yaw_scan = 0.0                                          # S3P:281
        self.vote_loc = np.array(half + [0.0], np.float64)           # S3P:265, 269-271: Tx_loc..rot_z_loc
        self.gyr_zyx = np.zeros(3)                                   # S3P:150
        self.gyr_yaw_old = 0.0                                       # S3P:141
        self.yaw_diff_scan = 0.0                                     # S3P:142
        self.pos_diff_scan = np.zeros(3)                             # S3P:143
        self.yaw_diff = 0.0                                          # S3P:139
        self.rot_y, self.rot_x = 0.0, 0.1                            # S3P:267-268
        # globals S3P:56-61
        self.local_pos = np.array([map_size[0] / 2, map_size[1] / 2, 1.5])
        self.local_yaw = 0.0
        self.pos = np.array([7.5, 7.5, 1.5])
        self.yaw = 0.0
        self.feedback = np.zeros(4)                                  # pos_xyz_diff, rot_z_diff
        self.update_params = capi.UpdateParams(10, 50, -3, -50, 3.0)  # S3P:637, 650-655, 695-700
        self.n = 0
        # cloudpoints (S3P:126): rows beyond the current count keep what earlier frames left there
        self.capacity = (width // pool) * (height // pool)
        if self.device_cloud:
            import torch      # device memory only: the buffer the four pf_*_device calls share
            self.cloud = None
            self._cloud_dev = torch.zeros((self.capacity + 1) * 3, dtype=torch.float64,
                                          device=torch.device("cuda", int(loc.cfg.device)))
            torch.cuda.synchronize()
            self.cloud_ptr = self._cloud_dev.data_ptr()
            mapper.set_stream(loc.stream())              # one stream orders the buffer's writer and its readers
        else:
            self.cloud = np.zeros((self.capacity + 1, 3), np.float64)
        self.last_vote = -1
        self.last_flags = 0

    def close(self):
        """device_cloud: the mapper goes back to its own stream (call before `loc`, whose stream it borrowed, is closed)."""
        if self.device_cloud and self.mapper is not None:
            self.mapper.set_stream(None)
            self.device_cloud = False

    # ---- S3P:173-185 ----
    def depth(self, image):
        if self.device_cloud:
            self.n = self.loc.depth_to_cloud_device(image, self.width, self.height, self.cloud_ptr, self.capacity,
                                                    self.pool, self.focal)
            return None
        c = self.loc.depth_to_cloud(image, self.width, self.height, self.pool, self.focal)
        self.n = len(c)
        self.cloud[:self.n] = c
        return self.cloud[:self.n]

    def set_gyr(self, zyx=None):
        """zyx: what sync_filter() would leave (S3P:244-246); None: no IMU sample (S3P:184)."""
        if zyx is None:
            self.gyr_zyx[0] = self.yaw + self.yaw_diff_scan
        else:
            self.gyr_zyx[:] = zyx

    # ---- scan_localization(): S3P:300-410 ----
    def scan_localization(self, seed):
        self.rot_y, self.rot_x = float(self.gyr_zyx[1]), float(self.gyr_zyx[2])      # S3P:301-302
        self.loc.ring_paint()                                                        # S3P:305-312
        # S3P:315-319 draws in [0, n] inclusive, one row past the cloud; replay=True keeps that (fixture seeds are
        # screened so that row n is never drawn); otherwise the draw is in [0, n-1] like driver/step_loop.cpp's
        pick = capi.draw_indices(seed, N_SCAN_POINTS, self.n if self.replay else self.n - 1)
        zyx = (self.yaw_scan, self.rot_y, self.rot_x)
        if self.device_cloud:                                                        # S3P:320-322 on the device
            _, best = self.loc.scan_vote_device(self.cloud_ptr, self.n, pick, self.pos_scan, zyx, **VOTE)
        else:
            sample = self.cloud[pick]                                                # S3P:320-322
            _, best = self.loc.scan_vote(sample, self.pos_scan, zyx, **VOTE)         # S3P:328-376
        hyp = capi.vote_hypothesis(best, self.pos_scan, zyx, self.resolution, **VOTE)
        if hyp is not None:                                                          # S3P:365-372
            self.vote_loc[:] = hyp
        self.last_vote = best
        (self.pos_scan, self.yaw_scan, self.gyr_yaw_old, self.pos_diff_scan, self.yaw_diff_scan,
         self.yaw_diff) = capi.vote_epilogue(self.pos_scan, self.yaw_scan, self.vote_loc,
                                             float(self.gyr_zyx[0]), self.gyr_yaw_old)   # S3P:379-404
        return pick

    # ---- scan_localization_map_update(): S3P:419-529 ----
    def map_update(self):
        zyx = (self.yaw_scan, self.rot_y, self.rot_x)
        if self.device_cloud:
            self.last_flags, self.pos_scan = self.loc.ring_update_device(self.cloud_ptr, self.n, self.pos_scan, zyx)
        else:
            self.last_flags, self.pos_scan = self.loc.ring_update(self.cloud[:self.n], self.pos_scan, zyx)
        return self.last_flags

    # ---- build_local_map(): S3P:621-710 (the caller holds its mutex around this, S3P:622/708) ----
    def build_local_map(self, seed):
        self.local_pos = self.local_pos + self.pos_diff_scan                         # S3P:623-625
        self.local_yaw = self.local_yaw + self.yaw_diff                              # S3P:626
        rays = capi.draw_indices(seed, N_RAYS, self.n if self.replay else self.n - 1)   # S3P:664-669 (inclusive in replay)
        if np.any(rays >= self.n):
            raise ValueError("a ray index names the row behind the cloud (S3P:666 draws it; the reference then "
                             "marches to a stale point): replay=True is for seeds screened against that")
        if self.device_cloud:
            self.mapper.set_scan_device(self.cloud_ptr, self.n, rays)
        else:
            self.mapper.set_scan(self.cloud[:self.n], rays)
        self.mapper.update_map(capi.PF_GRID_LOCAL, self.update_params, self.local_pos,
                               (self.local_yaw, self.rot_y, self.rot_x))             # S3P:628-704
        self.pos, self.yaw = capi.update_realtime_pos(self.pos, self.yaw, self.pos_diff_scan,
                                                      self.yaw_diff, self.feedback)  # S3P:706, 712-728
        return rays

    def frame(self, image, gyr_zyx, seed_vote, seed_rays):
        """One callback() (S3P:168-205) without the ROS message handling."""
        self.depth(image)
        self.set_gyr(gyr_zyx)
        self.scan_localization(seed_vote)
        self.map_update()
        self.build_local_map(seed_rays)

    # ---- the snapshot of particle_map(): S3P:833-846 (under the caller's mutex) ----
    def snapshot(self, origin=(100, 100, 30)):
        """Returns (beams, local_pose[4], global_pose[4]) and resets the local map and frame."""
        cfg = self.mapper.cfg
        local = self.mapper.read_local_map().reshape(cfg.local_x, cfg.local_y, cfg.local_z)
        local_pose = np.array([*self.local_pos, self.local_yaw])
        global_pose = np.array([*self.pos, self.yaw])
        self.mapper.reset_local_map()                                                # S3P:840
        self.local_pos = np.array([5.0, 5.0, 1.5]); self.local_yaw = 0.0             # S3P:841-844
        return capi.snapshot_beams(local, origin), local_pose, global_pose

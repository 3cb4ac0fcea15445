"""GPU test (-m gpu) of pf_scan_vote_static, the one entry point that allocates nothing (its kernels'
scratch is a file-scope device array): its output against pf_scan_vote's AND the oracle's (pfo_scan_vote) on
the same inputs, bit for bit, on two calls in a row, the second with a smaller input; that a warmed-up
call leaves the device's free memory and the caller's current device as they were; and the size limits."""
import numpy as np
import pytest

from slam_amd import capi
from oracle import binding as O
import golden_util as G

pytestmark = pytest.mark.gpu


def test_static_scan_vote_equals_scan_vote(golden_dir):
    import torch
    dev = torch.device("cuda:0")
    g = G.load(golden_dir, "vote.npz")
    dims = (200, 200, 80)
    h = capi.Handle(capi.default_config(n_particles=1, map_mode=capi.PF_MAP_SHARED, grid_x=dims[0], grid_y=dims[1],
                                        grid_z=dims[2], local_x=0, local_y=0, local_z=0, max_beams=64))
    try:
        # first call: the reference's lattice (5 yaw x 5^3) on 3000 points of a reference case;
        # second call, right behind it: fewer points, a smaller lattice, another case
        calls = [("c2_match_tilt", 3000, dict(n_yaw=5, yaw_k0=-6, yaw_kstep=3, n_t=5, length=0.25)),
                 ("c6_wrap_pos", 500, dict(n_yaw=3, yaw_k0=-3, yaw_kstep=3, n_t=4, length=0.2))]
        outs = []
        stream = torch.cuda.Stream(device=dev)
        keep = []
        capi.scan_vote_static_warmup(0)            # resolves the scratch array's address: from here on nothing is looked up
        free_before = None
        for name, n, lat in calls:
            grid = np.zeros(G.NL, np.uint8); grid[g[name + "_grid_idx"]] = 1
            cloud = G.cloud_of(g[name + "_cloud_mm"])[:n]
            before = g[name + "_before"]
            pos, zyx = before[0:3], np.array([before[3], g[name + "_gyr"][1], g[name + "_gyr"][2]])
            t_grid = torch.from_numpy(grid).to(dev)
            t_pts = torch.from_numpy(np.ascontiguousarray(cloud)).to(dev)
            t_sums = torch.full((lat["n_yaw"] * lat["n_t"] ** 3,), -7, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            keep.append((t_grid, t_pts, t_sums))
            if free_before is None:
                # one throw-away call first: the first launch on a new stream makes the HIP runtime create the
                # stream's hardware queue and load the kernels' code object (2 MiB of device memory on this image),
                # which no entry point can avoid; what is checked is that the calls behind it allocate nothing
                t_tmp = torch.zeros_like(t_sums)
                capi.scan_vote_static(0, stream.cuda_stream, t_grid.data_ptr(), dims, 0.05, t_pts.data_ptr(), len(cloud),
                                      pos, zyx, t_tmp.data_ptr(), **lat)
                stream.synchronize()
                keep.append(t_tmp)
                free_before = torch.cuda.mem_get_info(dev)[0]
            capi.scan_vote_static(0, stream.cuda_stream, t_grid.data_ptr(), dims, 0.05, t_pts.data_ptr(), len(cloud),
                                  pos, zyx, t_sums.data_ptr(), **lat)
            if len(outs) == 0:
                # nothing was allocated by the call (hipMemGetInfo's free bytes are what they were)
                assert torch.cuda.mem_get_info(dev)[0] == free_before
            assert torch.cuda.current_device() == 0
            outs.append((name, grid, cloud, pos, zyx, lat, t_sums))
        stream.synchronize()
        for name, grid, cloud, pos, zyx, lat, t_sums in outs:
            h.write_map(0, grid)
            want, best = h.scan_vote(cloud, pos, zyx, **lat)
            got = t_sums.cpu().numpy()
            assert np.array_equal(got, want), name
            assert want.max() > 0
            # ... and the oracle's restatement of S3P:328-376 on the same case
            ref, ref_best = O.scan_vote(grid.reshape(dims), 0.05, cloud, pos, zyx, **lat)
            assert np.array_equal(got, ref), name
            assert best == ref_best, name
    finally:
        h.close()


def test_static_scan_vote_rejects_sizes_beyond_its_scratch():
    import torch
    dev = torch.device("cuda:0")
    t_grid = torch.zeros(8 * 8 * 8, dtype=torch.uint8, device=dev)
    t_pts = torch.zeros((8, 3), dtype=torch.float64, device=dev)
    t_sums = torch.zeros(16 * 512, dtype=torch.int32, device=dev)
    args = (0, 0, t_grid.data_ptr(), (8, 8, 8), 0.05, t_pts.data_ptr())
    pos, zyx = np.zeros(3), np.zeros(3)
    for n, lat in ((4097, {}), (8, dict(n_yaw=17)), (8, dict(n_t=9)), (-1, {})):
        with pytest.raises(capi.PfError):
            capi.scan_vote_static(*args, n, pos, zyx, t_sums.data_ptr(), **lat)
    capi.scan_vote_static(*args, 8, pos, zyx, t_sums.data_ptr(), n_yaw=16, n_t=8)     # the maxima themselves pass
    torch.cuda.synchronize()

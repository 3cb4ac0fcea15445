#!/usr/bin/env python3
"""bench.py — particle·beam likelihood evaluations per second (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload at N=1: BASELINE.json configs[1] — 10k particles x 4k-beam scan on a 256^3 shared
uint8 grid. N>1: the same per GPU (weak scaling): particles sharded in equal blocks, grid
replicated, one RCCL all-gather of the scores and one of the pose blocks per step.

A step = one pass of the hot path on inputs resident in HBM, in the reference's order:
measure (K1) -> update_map (K5, every beam cast) -> predict -> weights (K3) ->
resample (K4).
The caller-side work (the motion model with its host-libm trig, the beam list, the scan, the
thresholds) happens before the timed region, as the contract's "inputs already resident"
asks: the predicted pose blocks of the steps are computed by pf_predict beforehand, kept in
HBM, and handed over per step by pf_predict_device, so every measure sees a fresh, realistic
cloud (resampled duplicates of last step's noisy poses) instead of a collapsing one.
DESIGN.md quotes the upload-inclusive rate. One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=10000, help="per GPU")
    ap.add_argument("--beams", type=int, default=4096)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frame-leg", action="store_true",
                    help="skip the per-frame path (depth -> cloud -> vote -> ring -> local map) timed after the step")
    ap.add_argument("--frame-only", action="store_true", help="run only the per-frame leg and print its block (profiling aid)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-update", action="store_true", help="leave K5 out of the step")
    ap.add_argument("--time-all-kernels", action="store_true",
                    help="HIP events around K3/K4/K5 too (kernel_us); costs host time in the step")
    ap.add_argument("--no-exchange-check", action="store_true",
                    help="several ranks: skip the start-up comparison of the in-library exchange with torch.distributed")
    ap.add_argument("--watchdog-seconds", type=float, default=480.0,
                    help="several ranks only: end the job with exit code 124 if it has not finished by then (0: never)")
    ap.add_argument("--torch-collectives", action="store_true",
                    help="sharded runs: all-gathers through torch.distributed instead of the library's own RCCL calls")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="read every step's results before the next step is enqueued (pf_resample) instead of "
                         "one step later (pf_resample_async / pf_resample_result)")
    ap.add_argument("--resident-thresholds", action="store_true",
                    help="keep one set of resampling thresholds resident instead of handing pf_resample new ones "
                         "every step (the default, as the reference draws them per step, S3P:769: 8 B per "
                         "particle per step)")
    ap.add_argument("--static-sensor", action="store_true",
                    help="K5 from the same sensor pose every step (cells saturate at their clamps) instead of "
                         "a pose that moves along a loop")
    ap.add_argument("--host-inclusive", action="store_true",
                    help="hand over the caller's inputs from host memory every step: pf_set_beams (beam list of the "
                         "step's snapshot), pf_set_scan, pf_predict (host poses: libm trig + upload) and the "
                         "thresholds, instead of inputs resident in HBM; reported in config, never the headline")
    ap.add_argument("--settle-seconds", type=float, default=1.0,
                    help="untimed settling before the warmup steps: 12000 steps per unit (a step count, so that every rank enqueues "
                         "the same collectives; about this many seconds on a fresh box, whose first process enqueues slowly)")
    ap.add_argument("--profile-steps", type=int, default=20,
                    help="steps of the second, untimed pass that times K3/K4/K5 with HIP events (0: skip)")
    ap.add_argument("--no-fuse-kernels", action="store_true",
                    help="skip the untimed third pass that times K2 (measure + fuse on per-particle maps) for "
                         "roofline_other_kernels: BASELINE configs[3]'s shape and the reference's own dense beam list")
    ap.add_argument("--force-sharded", action="store_true",
                    help="rehearsal: use the multi-GPU code path (RCCL all-gathers) even at N=1")
    return ap.parse_args()


def cpu_baseline(sc, dims, P, M, seconds):
    """The oracle (a port of the reference's loops, oracle/pf_oracle.c) timed on this host, one
    thread — the reference runs this path on one thread (S3P:1107). Bounded sample: whole steps
    of the same workload (measure + weights + resample + map update) until `seconds` elapse."""
    from oracle import binding as O
    g = O.ref_grid(*dims)
    grid = sc["grid"].reshape(-1).copy()
    ucfg = O.UpdateCfg(dims[0], dims[1], dims[2], 0.05, 0, 10, 100, -3, 0, 3.0)
    w = np.full(P, 1.0 / P)
    u = np.random.default_rng(3).random(P)
    pos = sc["true_pose"][:3].copy(); zyx = np.array([sc["true_pose"][3], 0.0, 0.0])
    reps, t0 = 0, time.perf_counter()
    t_measure = 0.0
    while True:
        a = time.perf_counter()
        s, c = O.measure(g, sc["x"], sc["y"], sc["z"], sc["yaw"], grid, 0, *sc["beams"], 0)
        t_measure += time.perf_counter() - a
        ww = w.copy()
        O.weights(ww, s, int(c[-1]))
        O.resample(ww, u, 0)
        O.update_map(ucfg, grid.reshape(dims), pos, zyx, sc["points"], sc["ray_idx"])
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds or reps >= 50:
            break
    # (ii) of SURVEY 8d: the same step with the measure loop split over the host cores this job
    # may use (particles are independent, S3P:859; ctypes releases the GIL), the rest serial
    from concurrent.futures import ThreadPoolExecutor
    cores = max(1, int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1))
    cores = min(cores, 64)
    cuts = np.linspace(0, P, cores + 1).astype(int)
    flat = grid.reshape(-1)

    def part(k):
        lo, hi = cuts[k], cuts[k + 1]
        return O.measure(g, sc["x"][lo:hi], sc["y"][lo:hi], sc["z"][lo:hi], sc["yaw"][lo:hi], flat, 0,
                         *sc["beams"], 0)

    reps_mt, t1 = 0, time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while True:
            res = list(ex.map(part, range(cores)))
            s = np.concatenate([r[0] for r in res]); c = np.concatenate([r[1] for r in res])
            ww = w.copy()
            O.weights(ww, s, int(c[-1]))
            O.resample(ww, u, 0)
            O.update_map(ucfg, grid.reshape(dims), pos, zyx, sc["points"], sc["ray_idx"])
            reps_mt += 1
            el_mt = time.perf_counter() - t1
            if el_mt >= seconds / 2 or reps_mt >= 200:
                break
    return {"value": reps * P * M / el, "unit": "evals/s", "cores": 1, "kind": "port",
            "measure_only_value": reps * P * M / t_measure,
            "all_cores": {"value": reps_mt * P * M / el_mt, "unit": "evals/s", "cores": cores,
                          "sample": f"{reps_mt} whole steps in {el_mt:.1f} s, measure loop split over "
                                    f"{cores} threads by particle, weights/resample/update serial"},
            "sample": f"{reps} whole steps of the N=1 workload ({P} particles x {M} beams, "
                      f"{dims[0]}^3 grid) in {el:.1f} s, 1 thread, gcc -O2"}


def reference_own_case(seconds):
    """The reference itself beside the port: its particle_map() (S3P:827-962, the g++ -O2 build of its own
    text that oracle/build_ref.sh makes, oracle/_ref/libs3p_ref.so) timed on this host at its own sizes,
    the only ones it compiles for: 50 particles with a 300x300x80 map each, the local map an 8-frame
    snapshot of 138 710 non-zero cells (tests/golden/local_map_8frames_beams.npz). One thread, as the
    reference runs it (S3P:1107). None when the build is not in the tree."""
    import ctypes as C
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "oracle", "_ref", "libs3p_ref.so")
    fx = os.path.join(here, "tests", "golden", "local_map_8frames_beams.npz")
    if not (os.path.exists(so) and os.path.exists(fx)):
        return None
    r = C.CDLL(so)
    r.ref_map_local_ptr.restype = C.POINTER(C.c_int8)
    r.ref_set_particle.argtypes = [C.c_int] + [C.c_double] * 5
    r.ref_set_local_pose.argtypes = [C.c_double] * 4
    r.ref_set_global_pose.argtypes = [C.c_double] * 4
    r.ref_set_policy.argtypes = [C.c_int] + [C.c_double] * 4 + [C.c_int]
    b = np.load(fx)
    LX, LY, LZ = 200, 200, 80
    local = np.zeros((LX, LY, LZ), np.int8)
    local[b["bx"].astype(int) + 100, b["by"].astype(int) + 100, b["bz"].astype(int) + 30] = b["bv"]   # S3P:875-877
    n_beams = int(np.count_nonzero(local))
    P = 50
    r.ref_alloc_particles(P); r.ref_init_maps()
    rng = np.random.default_rng(1)
    for i in range(P):
        r.ref_set_particle(i, 7.5 + rng.normal(0, 0.03), 7.5 + rng.normal(0, 0.03), 1.5, rng.normal(0, 0.02), 1.0 / P)
    dst = np.ctypeslib.as_array(r.ref_map_local_ptr(), shape=(LX * LY * LZ,))
    buf = C.create_string_buffer(1 << 16)
    out = {}
    for add in (0, 1):
        reps, el = 0, 0.0
        while el < seconds / 2 and reps < 200:
            dst[:] = local.reshape(-1)                              # the call empties map_local (S3P:840)
            r.ref_set_local_pose(5.0, 5.0, 1.5, 0.0); r.ref_set_global_pose(7.5, 7.5, 1.5, 0.0)
            r.ref_set_policy(add, 7.5, 7.5, 1.5, 0.0, 0)
            r.ref_set_seed(100 + reps)
            t0 = time.perf_counter()
            r.ref_particle_map(buf, len(buf))
            el += time.perf_counter() - t0
            reps += 1
        out[add] = (reps * P * n_beams / el, reps, el)
    # the port on the same case (measure loop only: the reference's call also predicts, weighs and resamples
    # its 50 particles, which is noise beside 7 million evaluations)
    from oracle import binding as O
    g = O.ref_grid(300, 300, 80)
    maps = np.full((P, 300 * 300 * 80), 50, np.uint8)
    rng = np.random.default_rng(1)
    px = np.array([7.5 + rng.normal(0, 0.03) for _ in range(P)]); py = px[::-1].copy()
    beams = (b["bx"].astype(np.int32), b["by"].astype(np.int32), b["bz"].astype(np.int32), b["bv"])
    t0 = time.perf_counter()
    O.measure(g, px, py, np.full(P, 1.5), np.zeros(P), maps.reshape(-1), maps.shape[1], *beams, 0)
    port = P * len(beams[0]) / (time.perf_counter() - t0)
    return {"value": out[0][0], "unit": "evals/s", "cores": 1, "kind": "reference",
            "with_map_fuse_value": out[1][0], "port_same_case_value": port,
            "sample": f"the reference's own particle_map() at its own sizes (50 particles x {n_beams} beams, per-particle "
                      f"300x300x80 maps): {out[0][1]} calls without the map fuse in {out[0][2]:.1f} s, {out[1][1]} with it in "
                      f"{out[1][2]:.1f} s; g++ -O2, 1 thread"}


def exchange_self_check(capi, ShardedBackend, dist, cfg, sc, Pg, stream):
    """One step (measure, weights, resample) on two fresh handles of this rank, one exchanging through the
    library's RCCL group and one through torch.distributed: weights, slots, best index and poses must be the
    same bits. Returns True when they are."""
    outs = []
    u = capi.draw_uniform(77, Pg)
    for native in (True, False):
        h = capi.Handle(cfg)
        try:
            h.set_stream(stream.cuda_stream)
            b = ShardedBackend(h, dist, native=native)
            if native and not b.native:
                return False
            h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
            h.set_weights(np.full(Pg, 1.0 / Pg)); h.set_beams(*sc["beams"])
            b.measure(0)
            h.predict(sc["x"] + 0.01, sc["y"] - 0.01, sc["z"], sc["yaw"] + 0.005)
            b.weights()
            w_pre, _ = h.get_weights()
            slots, best, nohit, amb = b.resample(u)
            w_post, _ = h.get_weights()
            outs.append((w_pre, np.asarray(slots), int(best), int(nohit), w_post, np.stack(h.get_poses())))
        finally:
            h.close()
    a0, a1 = outs
    return (np.array_equal(a0[0], a1[0]) and np.array_equal(a0[1], a1[1]) and a0[2:4] == a1[2:4]
            and np.array_equal(a0[4], a1[4]) and np.array_equal(a0[5], a1[5]))


K1_TRAFFIC_FILE = "profiles/r3_k1_traffic.json"          # committed rocprofv3 --pmc passes (tools/pmc_k1.sh + tools/mk_traffic_json.py)
K1_STATS_FILE = "profiles/r3_kernel_stats.csv"           # committed rocprofv3 --kernel-trace --stats summary of this command


def k1_counters(P, M, grid):
    """What the committed counter passes say about K1 (HBM-side bytes per launch, L2 hit rate), or {} when this run is not
    the workload they were taken on. They are NOT measured in this run: a --pmc pass is a run of its own."""
    try:
        with open(os.path.join(ROOT, K1_TRAFFIC_FILE)) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return {}
    if (P, M, grid) != (10000, 4096, 256):
        return {}
    return {"traffic": t.get("traffic_bytes_per_launch"), "l2_hit_rate": t.get("l2_hit_rate"),
            "counters_source": f"{K1_TRAFFIC_FILE} (committed rocprofv3 --pmc passes on this kernel, {t.get('kernel')}; not measured in this run)"}


def k1_rocprof_fraction(k1_bytes, P, M, grid):
    """K1's roofline fraction by rocprofv3's average kernel duration in the committed summary of this command."""
    if (P, M, grid) != (10000, 4096, 256):
        return {}
    try:
        import csv
        with open(os.path.join(ROOT, K1_STATS_FILE)) as f:
            for r in csv.DictReader(f):
                if "k_measure_tile" in r["Name"] and int(r["Calls"]) > 1000:
                    us = float(r["AverageNs"]) / 1e3
                    return {"frac_rocprof": k1_bytes / (us * 1e-6) / 8e12, "avg_launch_us_rocprof": us,
                            "rocprof_source": f"{K1_STATS_FILE} (committed rocprofv3 --kernel-trace --stats of python bench.py)"}
    except (OSError, ValueError, KeyError):
        pass
    return {}


def stream_copy_gbps(device, mib=1024, reps=10):
    """What this box's HBM does on a plain device-to-device copy (bytes read + bytes written per second):
    the measured figure SURVEY 8d asks for beside the 8 TB/s the roofline is quoted against."""
    import torch
    a = torch.empty(mib << 20, dtype=torch.uint8, device=device); b = torch.empty_like(a)
    b.copy_(a); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    return 2.0 * reps * (mib << 20) / (e0.elapsed_time(e1) * 1e-3) / 1e9


def host_cpu():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return {"model": model, "cpus_online": os.cpu_count(),
            "cpus_this_job_may_use": int(os.environ.get("OMP_NUM_THREADS", "0")) or None}


def fuse_kernels(device):
    """K2 (measure + fuse, S3P:859-908 with add_points on) timed on per-particle maps, outside the timed
    region, for the bench line's per-kernel rooflines: 2 B per evaluation (one byte read, one
    written). Two beam lists: BASELINE configs[3]'s shape (50 k particles x 128^3 maps, a sparse synthetic
    2048-beam scan: one beam per 64-byte line of a private map), and the list the reference itself fuses,
    tests/golden/local_map_8frames_beams.npz (its local map after 8 frames: 138 710 non-zero cells in 3 945
    columns), into 10 000 maps of its own shape, 300 x 300 x 80."""
    from slam_amd import capi, synth
    out = {}

    def timed(h, P, M, what, reps=3):
        h.measure(1); h.synchronize()
        h.profile(True, [capi.PF_K_MEASURE])
        for _ in range(reps):
            h.measure(1)
        h.synchronize()
        ms, n = h.kernel_time(capi.PF_K_MEASURE)
        h.profile(False)
        t = ms / n * 1e-3
        b = 2.0 * P * M
        out[what] = {"bytes": b, "achieved": b / t / 1e9, "unit": "GB/s", "frac": b / t / 8e12, "launch_us": t * 1e6,
                     "particles": P, "beams": M}

    # configs[3]
    P, M = 50000, 2048
    sc = synth.scene((128, 128, 128), P, M, seed=5)
    bx, by, bz, bv = sc["beams"]
    bv = np.random.default_rng(1).choice(np.array([-30, -3, 10, 30, 50], np.int8), len(bx))
    h = capi.Handle(capi.default_config(device=device, n_particles=P, grid_x=128, grid_y=128, grid_z=128,
                                        local_x=0, local_y=0, local_z=0, max_beams=len(bx)))
    h.broadcast_map(sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"]); h.set_beams(bx, by, bz, bv)
    timed(h, P, len(bx), "fuse_configs3_sparse_scan")
    h.close()
    # the reference's own list
    path = os.path.join(ROOT, "tests", "golden", "local_map_8frames_beams.npz")
    if os.path.exists(path):
        g = np.load(path)
        beams = (g["bx"].astype(np.int32), g["by"].astype(np.int32), g["bz"].astype(np.int32), g["bv"])
        P, M = 10000, len(beams[0])
        rng = np.random.default_rng(2)
        h = capi.Handle(capi.default_config(device=device, n_particles=P, max_beams=M))
        h.broadcast_map(np.full(300 * 300 * 80, 50, np.uint8))
        h.predict(7.5 + rng.normal(0, 0.06, P), 7.5 + rng.normal(0, 0.06, P), 1.5 + rng.normal(0, 0.03, P),
                  0.3 + rng.normal(0, 0.05, P))
        h.set_beams(*beams)
        timed(h, P, M, "fuse_reference_dense_list")
        h.close()
    return out


def frame_leg(reps=3):
    """The per-frame path beside the particle step (callback(), S3P:168-205): depth image -> cloud -> ring painting
    -> scan vote -> keyframe ring update -> local-map hits and ray march, on the 48 recorded frames of the
    reference's own run (tests/golden/frames.npz), through the same C ABI calls the parity tests use
    (slam_amd.frontend.FrameLoop). Outside the timed region of the step. `ms_per_frame` is host-inclusive wall time
    per callback() with the cloud kept in HBM between the stages (the depth image goes up, the point count, the
    vote's sums and the ring's count come back); `host_cloud` is the same with every stage handing the cloud to the
    host, as the reference's member array would; `kernel_us` holds the device time of the two stages the library
    times with events (vote, local-map update). Beside it the same call
    sequence of the reference itself (oracle/_ref: ref_depth_to_cloud .. ref_build_local_map, g++ -O2 build of
    its own text) on this host, one thread. Algorithmic bytes per stage, SURVEY 8d style: depth 2 B per pixel in
    + 24 B per point out; vote 1 B per look-up (hypotheses x points) + 24 B per rotated point; ring 28 B per
    scan point indexed + 5 B per painted index; local map 24 B per point + 2 B per visited voxel."""
    import ctypes as C
    from slam_amd import capi
    from slam_amd.frontend import FrameLoop
    fx = os.path.join(ROOT, "tests", "golden", "frames.npz")
    if not os.path.exists(fx):
        return None
    g = np.load(fx)
    F = int(g["frames"])
    images = [np.repeat(np.repeat(g["pooled"][f], 4, 0), 4, 1).astype("<u2").tobytes() for f in range(F)]
    out = {"frames": F, "source": "tests/golden/frames.npz (the reference's own 48-frame run: 640x480 depth, ~19 000 points, "
                                  "500-point x 625-hypothesis vote, 2400 rays)"}
    names = ("depth", "ring_paint+vote", "ring_update", "local_map")

    def run(device_cloud, want_bytes):
        loc = capi.Handle(capi.default_config(n_particles=1, map_mode=capi.PF_MAP_SHARED, grid_x=200, grid_y=200,
                                              grid_z=80, local_x=0, local_y=0, local_z=0, max_beams=64))
        mapper = capi.Handle(capi.default_config(n_particles=1, max_beams=1 << 18))
        try:
            loc.fill_maps(0); loc.ring_init()
            loc.profile(True, [capi.PF_K_VOTE]); mapper.profile(True, [capi.PF_K_UPDATE])
            fl = FrameLoop(loc, mapper, replay=True, device_cloud=device_cloud)
            stage = np.zeros(4); bytes_ = np.zeros(4)
            t_all = time.perf_counter()
            for f in range(F):
                t0 = time.perf_counter()
                fl.depth(images[f])
                t1 = time.perf_counter()
                fl.set_gyr(g["gyr"][f])
                fl.scan_localization(int(g["seed_vote"][f]))
                t2 = time.perf_counter()
                fl.map_update()
                t3 = time.perf_counter()
                rays = fl.build_local_map(int(g["seed_rays"][f]))
                t4 = time.perf_counter()
                stage += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
                if want_bytes:
                    n = fl.n
                    # visited voxels of the ray march (S3P:676, 685: round(20 |r|) - 1 steps per ray) + the hits within 3 m
                    d = fl.cloud[:n] * 1.0
                    rng_ = np.sqrt((d ** 2).sum(1))           # sensor-frame range = range in the local frame
                    steps = np.maximum(np.round(rng_[rays] * 20.0) - 1, 0).sum()
                    counts = loc.ring_state()[0]
                    bytes_ += (2.0 * 640 * 480 + 24.0 * n, 625.0 * 500 + 24.0 * 5 * 500, 28.0 * n + 5.0 * float(np.sum(counts)),
                               24.0 * n + 2.0 * (float(np.count_nonzero(rng_ <= 3.0)) + float(steps)))
            mapper.synchronize(); loc.synchronize()
            wall = time.perf_counter() - t_all
            kv, nv = loc.kernel_time(capi.PF_K_VOTE); ku, nu = mapper.kernel_time(capi.PF_K_UPDATE)
            cur = {"ms_per_frame": wall / F * 1e3, "stages_us": {k: float(v) / F * 1e6 for k, v in zip(names, stage)},
                   "kernel_us": {"vote": kv / max(nv, 1) * 1e3, "local_map": ku / max(nu, 1) * 1e3}}
            if want_bytes:
                out["algorithmic_bytes_per_frame"] = {k: float(v) / F for k, v in zip(names, bytes_)}
            return cur
        finally:
            mapper.set_stream(None)              # device_cloud: it borrowed loc's stream
            loc.close(); mapper.close()

    # the cloud handed over through host memory between the stages, as the reference's member array is (S3P:126) ...
    host = min((run(False, rep == 0) for rep in range(reps)), key=lambda c: c["ms_per_frame"])
    # ... and kept in one device buffer (pf_*_device): per frame the count, the vote's sums and the ring's count come back
    best = min((run(True, False) for rep in range(reps)), key=lambda c: c["ms_per_frame"])
    out.update(best)
    out["cloud"] = "device-resident (pf_depth_to_cloud_device, pf_scan_vote_device, pf_ring_update_device, pf_set_scan_device)"
    out["host_cloud"] = {"ms_per_frame": host["ms_per_frame"], "stages_us": host["stages_us"],
                         "what": "the same frames with pf_depth_to_cloud .. pf_set_scan: the cloud crosses PCIe four times per frame"}
    b = out["algorithmic_bytes_per_frame"]
    out["roofline"] = {"vote": {"bytes": b["ring_paint+vote"], "launch_us": out["kernel_us"]["vote"],
                                "frac": b["ring_paint+vote"] / (out["kernel_us"]["vote"] * 1e-6) / 8e12 if out["kernel_us"]["vote"] > 0 else None},
                       "local_map": {"bytes": b["local_map"], "launch_us": out["kernel_us"]["local_map"],
                                     "frac": b["local_map"] / (out["kernel_us"]["local_map"] * 1e-6) / 8e12 if out["kernel_us"]["local_map"] > 0 else None}}
    out["note"] = ("every stage is launch- and copy-bound at the reference's sizes: the kernels take a few us of a call whose "
                   "host side draws indices, uploads the image or a few hundred bytes and waits for a count or 625 sums")
    # the same loop without the Python around it: driver/step_loop.cpp replays the 48 frames (and the 6 filter steps
    # between them, outside its callback timer) and reports the wall time per callback()
    exe = os.path.join(ROOT, "driver", "step_loop")
    if os.path.exists(exe):
        import struct, subprocess, tempfile
        with tempfile.NamedTemporaryFile(suffix=".replay") as fp:
            fp.write(struct.pack("<iii", F, int(g["per_step"]), int(g["P"])))
            fp.write(np.ascontiguousarray(g["pooled"], "<u2").tobytes())
            fp.write(np.ascontiguousarray(g["gyr"], "<f8").tobytes())
            for k in ("seed_vote", "seed_rays", "step_seed"):
                fp.write(np.ascontiguousarray(g[k]).astype("<u4").tobytes())
            fp.flush()
            drv = {}
            for mode, key in (("replay-device", "us_per_frame"), ("replay", "us_per_frame_host_cloud")):
                best_us = None
                for rep in range(reps):
                    r = subprocess.run([exe, mode, fp.name], capture_output=True, text=True, timeout=120)
                    if r.returncode != 0:
                        break
                    for line in r.stderr.splitlines():
                        if line.startswith("callback():"):
                            us = float(line.split("us each,")[1].split("us")[0])     # the frames after the first filter step
                            best_us = us if best_us is None else min(best_us, us)
                if best_us is not None:
                    drv[key] = best_us
            if drv:
                drv["what"] = ("driver/step_loop.cpp (the C++ caller) on the same 48 frames, a fresh process per run; wall time per "
                               "callback() body over the 40 frames after the first filter step (the first eight load code objects "
                               "and allocate scratch)")
                out["cpp_driver"] = drv
    # the reference's own callback sequence on this host
    so = os.path.join(ROOT, "oracle", "_ref", "libs3p_ref.so")
    if os.path.exists(so):
        r = C.CDLL(so)
        r.ref_set_gyr.argtypes = [C.c_double] * 3
        el = 0.0
        for rep in range(reps):
            r.ref_slam_reset()
            t_rep = 0.0
            for f in range(F):
                raw = np.frombuffer(images[f], np.uint8)
                t0 = time.perf_counter()
                r.ref_depth_to_cloud(raw.ctypes.data_as(C.c_void_p), len(raw))                 # S3P:173-175
                r.ref_set_gyr(*[float(v) for v in g["gyr"][f]])
                r.ref_set_seed(int(g["seed_vote"][f]))
                r.ref_scan_localization()                                                      # S3P:187
                r.ref_scan_localization_map_update()                                           # S3P:189
                r.ref_set_seed(int(g["seed_rays"][f]))
                r.ref_build_local_map()                                                        # S3P:191
                t_rep += time.perf_counter() - t0
            el = t_rep if rep == 0 else min(el, t_rep)
        out["cpu_reference"] = {"ms_per_frame": el / F * 1e3, "kind": "reference", "cores": 1,
                                "what": "callback() body of the reference's own text (oracle/_ref), same 48 frames and seeds"}
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` launched plainly: one child process per GPU, started before this
    process has touched the GPU, the same rank environment torch.distributed.run would set. Rank 0's
    stdout (the one JSON line) passes through."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = pr.wait() or rc
    sys.exit(rc)


def update_traffic(sc, pos_list, res=0.05, max_range=3.0):
    """Algorithmic bytes of one K5 launch: 24 B per endpoint + 2 B per visited voxel
    (hits within range + the steps of every ray, S3P:676-685), averaged over the sensor poses."""
    pts = sc["points"]; n = np.linalg.norm(pts, axis=1)
    rays = n[sc["ray_idx"]]
    steps = np.maximum(np.rint(rays / res) - 1, 0).sum()
    hits = int((n <= max_range).sum())
    return 24.0 * len(pts) + 2.0 * (hits + steps)


def main():
    a = parse()
    if a.frame_only:
        print(json.dumps({"frame": frame_leg()}), flush=True)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    import torch
    import torch.distributed as dist
    from slam_amd import capi, synth
    from slam_amd.sharded import ShardedBackend

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or a.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        if world > 1 and a.watchdog_seconds > 0:
            # a job of several ranks that stops making progress (a collective one rank never enters) would
            # otherwise sit there until whoever launched it gives up: end it with a message instead
            import threading

            def give_up():
                sys.stderr.write(f"bench.py: rank {rank} of {world} did not finish within {a.watchdog_seconds:.0f} s "
                                 "(a collective that never returned?): giving up\n")
                sys.stderr.flush()
                os._exit(124)
            wd = threading.Timer(a.watchdog_seconds, give_up)
            wd.daemon = True
            wd.start()

    P, M, dims = a.particles, a.beams, (a.grid, a.grid, a.grid)
    Pg = P * world
    # the scene (grid, scan, beams) is the same on every rank; each rank owns its own particles
    sc = synth.scene(dims, P, M, seed=0)
    rng = np.random.default_rng(1000 + rank)
    tp = sc["true_pose"]
    sc["x"] = tp[0] + rng.normal(0, 0.03, P); sc["y"] = tp[1] + rng.normal(0, 0.03, P)
    sc["z"] = tp[2] + rng.normal(0, 0.015, P); sc["yaw"] = tp[3] + rng.normal(0, 0.02, P)
    M = len(sc["beams"][0])

    cfg = capi.default_config(device=local_rank, n_particles=P, n_particles_global=Pg,
                              particle_offset=rank * P, map_mode=capi.PF_MAP_SHARED,
                              grid_x=dims[0], grid_y=dims[1], grid_z=dims[2],
                              local_x=0, local_y=0, local_z=0, max_beams=max(M, 1),
                              max_points=max(len(sc["points"]), 1))
    h = capi.Handle(cfg)
    if sharded:
        # the handle enqueues on torch's current stream: the RCCL all-gathers are ordered against
        # it, so the step needs no host synchronisation between kernels and collectives
        work_stream = torch.cuda.Stream(device=local_rank)
        torch.cuda.set_stream(work_stream)
        h.set_stream(work_stream.cuda_stream)
    be = ShardedBackend(h, dist, native=not a.torch_collectives) if sharded else h
    collectives_note = None
    if sharded and be.native and not a.no_exchange_check:
        # The library's own RCCL exchange has only ever run on one rank before a job like this one: check one
        # whole step of it against the same step through torch.distributed (two handles, same inputs) before
        # anything is timed, and use the torch exchange if the two differ or the library's fails.
        ok = 1
        try:
            ok = int(exchange_self_check(capi, ShardedBackend, dist, cfg, sc, Pg, work_stream))
        except Exception as e:                                # a failed call, not a hang (the watchdog has those)
            sys.stderr.write(f"bench.py: rank {rank}: exchange self-check raised {type(e).__name__}: {e}\n")
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{local_rank}")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            be = ShardedBackend(h, dist, native=False)
            collectives_note = "torch.distributed (the in-library exchange failed its start-up check against it)"
        else:
            collectives_note = "in-library RCCL group (start-up check against torch.distributed: identical)"
    h.write_map(0, sc["grid"])
    h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
    h.set_weights(np.full(Pg, 1.0 / Pg))
    h.set_beams(*sc["beams"])
    h.set_scan(sc["points"], sc["ray_idx"])
    u = capi.draw_uniform(3, Pg)                     # the same thresholds on every rank
    u_fresh = None if a.resident_thresholds else [capi.draw_uniform(100 + k, Pg) for k in range(8)]
    upd = capi.UpdateParams(10, 100, -3, 0, 3.0)
    # the sensor of the map update moves along a loop (0.5 m radius, 2 cm and 0.01 rad per step), so
    # K5 keeps meeting cells that are not yet at their clamps; identical on every rank
    NPOSE = 1 if a.static_sensor else 157
    ang = 2 * np.pi * np.arange(NPOSE) / NPOSE
    sensor_pos = [tp[:3] + (0.0 if a.static_sensor else 0.5) * np.array([np.cos(t) - 1.0, np.sin(t), 0.0]) for t in ang]
    sensor_zyx = [np.array([tp[3] + (0.0 if a.static_sensor else 0.01 * k), 0.0, 0.0]) for k in range(NPOSE)]
    # predicted pose blocks (S3P:911-914 outcome), resident in HBM: NBLK different noise draws
    NBLK = 8
    blocks = torch.empty((NBLK, 9, P), dtype=torch.float64, device=f"cuda:{local_rank}")
    for b in range(NBLK):
        r2 = np.random.default_rng(5000 + 97 * rank + b)
        h.predict(tp[0] + r2.normal(0, 0.03, P), tp[1] + r2.normal(0, 0.03, P),
                  tp[2] + r2.normal(0, 0.015, P), tp[3] + r2.normal(0, 0.02, P))
        h.synchronize()
        blocks[b].copy_(h.poses_tensor())
    torch.cuda.synchronize()
    block_ptrs = [blocks[b].data_ptr() for b in range(NBLK)]
    step_no = [0]
    host_poses = None
    if a.host_inclusive:
        host_poses = []
        for b in range(NBLK):
            r2 = np.random.default_rng(5000 + 97 * rank + b)
            host_poses.append((tp[0] + r2.normal(0, 0.03, P), tp[1] + r2.normal(0, 0.03, P),
                               tp[2] + r2.normal(0, 0.015, P), tp[3] + r2.normal(0, 0.02, P)))
        # four beam lists / scans of the same scene (the list changes with every snapshot, S3P:833-846)
        host_beams = [tuple(np.ascontiguousarray(np.roll(v, 7 * k)) for v in sc["beams"]) for k in range(4)]
        host_scans = [(np.ascontiguousarray(np.roll(sc["points"], 5 * k, axis=0)), sc["ray_idx"]) for k in range(4)]

    def enqueue(first=False):
        if a.host_inclusive:
            k = step_no[0]
            h.set_beams(*host_beams[k % 4])
            h.set_scan(*host_scans[k % 4])
        be.measure(0)
        if not a.no_update:
            # the grid update only has to follow this step's measure and precede the next one
            h.update_map(capi.PF_GRID_SHARED, upd, sensor_pos[step_no[0] % NPOSE], sensor_zyx[step_no[0] % NPOSE])
        if a.host_inclusive:
            h.predict(*host_poses[step_no[0] % NBLK])
        else:
            h.predict_device(block_ptrs[step_no[0] % NBLK])
        step_no[0] += 1
        be.weights()

    def thresholds():
        return u_fresh[step_no[0] % 8] if u_fresh else None

    def step(first=False):
        enqueue()
        return be.resample(u if first else thresholds(), want_slots=False)

    def fence():
        h.synchronize()
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()

    step(first=True)                                 # makes the thresholds resident
    # untimed settling before the W warmup steps: the first process on a fresh box enqueues at less
    # than half its steady rate for the first few hundred steps (host side: the kernels' own times
    # are the same from the first launch)
    # (a fixed number of steps, not a time: every rank must enqueue the same collectives)
    for _ in range(int(a.settle_seconds * 12000)):
        step()
    for _ in range(a.warmup):
        step()
    # K1 is timed live with HIP events on its stream (roofline); the other classes only on request
    h.profile(True, None if a.time_all_kernels else [capi.PF_K_MEASURE])
    fence()
    t0 = time.perf_counter()
    amb_total = nohit_total = 0
    if a.no_pipeline:
        for _ in range(a.steps):
            _, _, nohit, amb = step()
            amb_total += amb; nohit_total += nohit
    else:
        # every step's work and every step's results, the results read one step late: step i+1 is
        # enqueued before the host waits for step i (its inputs here do not depend on them, and the
        # reference feeds the filter's correction back asynchronously too, S3P:956-959 / 712-716)
        for i in range(a.steps):
            enqueue()
            be.resample_async(thresholds(), want_slots=False) if sharded else h.resample_async(thresholds(), None, False)
            if i:
                _, _, nohit, amb = be.resample_result()
                amb_total += amb; nohit_total += nohit
        _, _, nohit, amb = be.resample_result()
        amb_total += amb; nohit_total += nohit
    fence()
    el = time.perf_counter() - t0
    h.profile(False)
    if sharded:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    k1_ms, k1_n = h.kernel_time(capi.PF_K_MEASURE)
    ko_ms, ko_n = 0.0, 0
    k3_ms = k4_ms = k5_ms = 0.0
    k3_n = k4_n = k5_n = 0
    moved = 0
    if a.profile_steps > 0:
        # second, untimed pass: HIP events around K3, K4 and K5 as well (eight more event records per
        # step cost ~20 us of host time, so not in the timed region); K5 runs on its own here (in the
        # timed region it rides in K3's launches)
        if a.time_all_kernels:
            k3_ms, k3_n = h.kernel_time(capi.PF_K_WEIGHTS); k4_ms, k4_n = h.kernel_time(capi.PF_K_RESAMPLE)
            k5_ms, k5_n = h.kernel_time(capi.PF_K_UPDATE)
        h.profile(True, [capi.PF_K_WEIGHTS, capi.PF_K_RESAMPLE, capi.PF_K_UPDATE, capi.PF_K_ORDER])
        for i in range(a.profile_steps):
            be.measure(0)
            if not a.no_update:
                h.update_map(capi.PF_GRID_SHARED, upd, sensor_pos[step_no[0] % NPOSE], sensor_zyx[step_no[0] % NPOSE])
                h.synchronize()                              # runs the recorded update now, alone
            h.predict_device(block_ptrs[step_no[0] % NBLK])
            step_no[0] += 1
            be.weights()
            slots, _, _, _ = be.resample(thresholds(), want_slots=True)
            moved += int((slots != np.arange(Pg)).sum())
        fence()
        h.profile(False)
        a3, b3 = h.kernel_time(capi.PF_K_WEIGHTS); a4, b4 = h.kernel_time(capi.PF_K_RESAMPLE)
        a5, b5 = h.kernel_time(capi.PF_K_UPDATE)
        ko_ms, ko_n = h.kernel_time(capi.PF_K_ORDER)
        ko_n = a.profile_steps                               # two launches per step (ordering, score reduction)
        k3_ms += a3; k3_n += b3; k4_ms += a4; k4_n += b4; k5_ms += a5; k5_n += b5

    if rank == 0:
        evals = float(Pg) * M * a.steps
        k1_avg_s = k1_ms / max(k1_n, 1) * 1e-3
        # algorithmic bytes of one K1 launch: one grid byte per eval, 48 B pose in and
        # 8 B scores out per particle, 13 B per beam read once
        k1_bytes = P * M * 1 + 56 * P + 13 * M
        achieved = k1_bytes / k1_avg_s / 1e9 if k1_avg_s > 0 else 0.0
        out = {
            "metric": "particle-beam likelihood evals/sec", "value": evals / el, "unit": "evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C2 per GPU: {P} particles x {M}-beam scan, {a.grid}^3 shared "
                                   f"uint8 grid; step = measure+weights+resample"
                                   + ("" if a.no_update else "+update_map(all beams cast)"),
                       "particles_per_gpu": P, "particles_total": Pg, "beams": M,
                       "grid": list(dims), "order": "fast" if Pg > 2048 else "exact",
                       "collectives": (collectives_note or ("rccl in-library" if getattr(be, "native", False) else "torch.distributed nccl")) if sharded else "none",
                       "thresholds": "resident" if a.resident_thresholds else "new every step (S3P:769), staged in pinned memory",
                       "inputs": "host memory every step (pf_set_beams, pf_set_scan, pf_predict, thresholds)"
                                 if a.host_inclusive else "resident in HBM",
                       "sensor": "static" if a.static_sensor else f"moves along a {NPOSE}-pose loop",
                       "results_read": "every step, before the next is enqueued" if a.no_pipeline
                                       else "every step, one step late (pf_resample_async)",
                       "ambiguous_draws": int(amb_total), "nohit_draws": int(nohit_total)},
            "roofline": {"bound": "hbm", "kernel": "k_measure_tile", "achieved": achieved,
                         "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": None, "l2_hit_rate": None,
                         **k1_counters(P, M, a.grid), **k1_rocprof_fraction(k1_bytes, P, M, a.grid),
                         "measured_stream_copy": stream_copy_gbps(f"cuda:{local_rank}"),
                         "algorithmic_bytes_per_launch": k1_bytes,
                         "avg_launch_us": k1_avg_s * 1e6, "launches": int(k1_n),
                         "k1_evals_per_s": P * M / k1_avg_s if k1_avg_s > 0 else 0.0},
        }
        # the measure stage as a whole: lane ordering + K1 + score reduction (they exist only for K1)
        stage_s = (k1_ms / max(k1_n, 1) + ko_ms / max(ko_n, 1)) * 1e-3
        out["roofline"]["measure_stage_us"] = stage_s * 1e6
        out["roofline"]["measure_stage_frac"] = k1_bytes / stage_s / 8e12 if stage_s > 0 else 0.0
        ku = {"measure": k1_ms / max(k1_n, 1) * 1e3, "order_and_reduce": ko_ms / max(ko_n, 1) * 1e3}
        other = {}
        if k3_n:
            ku["weights"] = k3_ms / k3_n * 1e3
            b3 = 28.0 * Pg                                   # 16 P in + 12 P out
            other["weights"] = {"bytes": b3, "achieved": b3 / (k3_ms / k3_n * 1e-3) / 1e9, "unit": "GB/s",
                                "frac": b3 / (k3_ms / k3_n * 1e-3) / 8e12}
        if k4_n:
            ku["resample"] = k4_ms / k4_n * 1e3
            b4 = 20.0 * Pg + 40.0 * moved / max(a.profile_steps, 1)   # 16 P in + 4 P out + 40 B per replaced slot
            other["resample"] = {"bytes": b4, "achieved": b4 / (k4_ms / k4_n * 1e-3) / 1e9, "unit": "GB/s",
                                 "frac": b4 / (k4_ms / k4_n * 1e-3) / 8e12,
                                 "replaced_slots_per_step": moved / max(a.profile_steps, 1)}
        if k5_n:
            ku["update"] = k5_ms / k5_n * 1e3
            b5 = update_traffic(sc, sensor_pos)
            other["update"] = {"bytes": b5, "achieved": b5 / (k5_ms / k5_n * 1e-3) / 1e9, "unit": "GB/s",
                               "frac": b5 / (k5_ms / k5_n * 1e-3) / 8e12}
        if world == 1 and a.profile_steps > 0 and not a.no_fuse_kernels:
            try:                                             # K2, untimed third pass on handles of its own
                other.update(fuse_kernels(local_rank))
            except Exception as e:                           # (e.g. the maps do not fit beside another job's memory)
                other["fuse_error"] = str(e)
        out["kernel_us"] = ku
        out["roofline_other_kernels"] = other
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, dims, P, M, a.cpu_seconds)
            try:
                own = reference_own_case(min(a.cpu_seconds, 8.0))
            except Exception as e:                            # a baseline beside the baseline: never fatal
                own = {"error": f"{type(e).__name__}: {e}"}
            if own:
                out["cpu_baseline"]["reference_own_case"] = own
            out["cpu_baseline"]["host"] = host_cpu()
        if not a.no_frame_leg and a.gpus == 1 and not a.force_sharded:
            try:
                out["frame"] = frame_leg()
            except Exception as e:                         # (the step's line is the contract; this block is extra)
                out["frame"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    h.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

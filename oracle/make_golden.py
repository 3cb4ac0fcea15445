#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by RUNNING THE REFERENCE ITSELF.

Runs only in the build container: it loads oracle/_ref/libs3p_ref.so, which
oracle/build_ref.sh compiles from the text of /root/reference/Slam_3d_particle.cpp
(S3P:12-110, 731-962) where it lies. The fixtures hold data only: the inputs this script
synthesises, the libstdc++ draws the reference consumed (as plain arrays) and the values the
reference's own particle_map() / resample() left in its globals and on std::cout.

    python oracle/make_golden.py            # writes tests/golden/*.npz (about a minute)
"""
import ctypes as C
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(HERE, "_ref", "libs3p_ref.so")

PX, PY, PZ = 300, 300, 80        # S3P:40-42
LX, LY, LZ = 200, 200, 80        # S3P:30-32
ORIGIN = (100, 100, 30)          # S3P:875-877
RES = 0.05


def load_ref():
    r = C.CDLL(REF)
    r.ref_particle_map_ptr.restype = C.POINTER(C.c_uint8)
    r.ref_map_local_ptr.restype = C.POINTER(C.c_int8)
    r.ref_set_particle.argtypes = [C.c_int] + [C.c_double] * 5
    r.ref_set_local_pose.argtypes = [C.c_double] * 4
    r.ref_set_global_pose.argtypes = [C.c_double] * 4
    r.ref_set_policy.argtypes = [C.c_int] + [C.c_double] * 4 + [C.c_int]
    r.ref_rot_matrix.argtypes = [C.c_void_p] + [C.c_double] * 3
    return r


def world_map(rng):
    """A box room with two box obstacles in the particle-map frame: 100 = wall, 0 = free
    interior, 50 = unknown (outside), a few noisy cells."""
    m = np.full((PX, PY, PZ), 50, np.uint8)
    x0, x1, y0, y1, z0, z1 = 60, 240, 70, 230, 6, 70
    m[x0:x1 + 1, y0:y1 + 1, z0:z1 + 1] = 0
    m[x0, y0:y1 + 1, z0:z1 + 1] = 100; m[x1, y0:y1 + 1, z0:z1 + 1] = 100
    m[x0:x1 + 1, y0, z0:z1 + 1] = 100; m[x0:x1 + 1, y1, z0:z1 + 1] = 100
    m[x0:x1 + 1, y0:y1 + 1, z0] = 100; m[x0:x1 + 1, y0:y1 + 1, z1] = 100
    m[180:200, 120:150, z0:40] = 100
    m[100:115, 170:200, z0:30] = 100
    noise = rng.integers(0, PX * PY * PZ, 4000)
    m.reshape(-1)[noise] = rng.integers(0, 101, 4000).astype(np.uint8)
    return m


def local_scan(world, pose, rng, n_occ, n_free, n_weak):
    """An int8 local map (S3P:65) that a sensor at `pose` (x,y,z in metres, yaw) could have
    built: occupied cells near world walls within 3 m, weak cells, negative free cells."""
    x, y, z, yaw = pose
    T = np.array([x, y, z]) / RES
    c, s = np.cos(yaw), np.sin(yaw)
    occ = np.argwhere(world == 100).astype(np.float64)
    d = occ - T
    near = occ[(np.linalg.norm(d, axis=1) < 60) & (np.abs(d[:, 2]) < 28)]
    pick = near[rng.choice(len(near), size=min(n_occ + n_weak, len(near)), replace=False)]
    d = pick - T
    lx = np.rint(c * d[:, 0] + s * d[:, 1]).astype(np.int64) + ORIGIN[0]
    ly = np.rint(-s * d[:, 0] + c * d[:, 1]).astype(np.int64) + ORIGIN[1]
    lz = np.rint(d[:, 2]).astype(np.int64) + ORIGIN[2]
    local = np.zeros((LX, LY, LZ), np.int8)
    ok = (lx >= 0) & (lx < LX) & (ly >= 0) & (ly < LY) & (lz >= 0) & (lz < LZ)
    lx, ly, lz = lx[ok], ly[ok], lz[ok]
    vals = rng.choice(np.array([30, 40, 50], np.int8), size=len(lx))
    weak = rng.random(len(lx)) < n_weak / max(1, n_occ + n_weak)
    vals[weak] = rng.choice(np.array([10, 20], np.int8), size=int(weak.sum()))
    local[lx, ly, lz] = vals
    # free cells: points on segments from the sensor cell towards occupied cells
    k = rng.integers(0, len(lx), n_free)
    t = rng.random(n_free) * 0.9
    fx = np.rint(ORIGIN[0] + (lx[k] - ORIGIN[0]) * t).astype(np.int64)
    fy = np.rint(ORIGIN[1] + (ly[k] - ORIGIN[1]) * t).astype(np.int64)
    fz = np.rint(ORIGIN[2] + (lz[k] - ORIGIN[2]) * t).astype(np.int64)
    free_vals = (-3 * rng.integers(1, 17, n_free)).astype(np.int8)
    empty = local[fx, fy, fz] == 0
    local[fx[empty], fy[empty], fz[empty]] = free_vals[empty]
    return local


def set_particles(r, st):
    for i in range(st.shape[0]):
        r.ref_set_particle(i, *[float(v) for v in st[i]])


def get_particles(r, P):
    out = np.empty((P, 5), np.float64)
    buf = (C.c_double * 5)()
    for i in range(P):
        r.ref_get_particle(i, buf)
        out[i] = buf[:]
    return out


def pmap(r, i):
    return np.ctypeslib.as_array(r.ref_particle_map_ptr(i), shape=(PX * PY * PZ,))


def parse_weights(text):
    lines = text.split("\n")
    assert lines[0].strip() == "weights", lines[0]
    vals = lines[1].split()
    # the closing "weights" marker follows the last weight on the same line (S3P:937-939)
    assert vals[-1] == "weights", vals[-3:]
    return np.array([float(v) for v in vals[:-1]], np.float64)


def run_step(r, P, world, state, local, local_pose, global_pose, policy, seed):
    """One call of the reference's particle_map(). Returns a dict of outputs."""
    for i in range(P):
        pmap(r, i)[:] = world.reshape(-1)
    set_particles(r, state)
    np.ctypeslib.as_array(r.ref_map_local_ptr(), shape=(LX * LY * LZ,))[:] = local.reshape(-1)
    r.ref_set_local_pose(*[float(v) for v in local_pose])
    r.ref_set_global_pose(*[float(v) for v in global_pose])
    r.ref_set_policy(int(policy["add"]), *[float(v) for v in policy["last"]], int(policy["best"]))
    r.ref_set_seed(seed)
    buf = C.create_string_buffer(1 << 22)
    r.ref_particle_map(buf, len(buf))
    text = buf.value.decode()
    out = {}
    out["w_pre"] = parse_weights(text)
    out["state"] = get_particles(r, P)
    add = C.c_int(); best = C.c_int(); last = (C.c_double * 4)()
    r.ref_get_policy(C.byref(add), last, C.byref(best))
    out["add_after"] = add.value; out["best"] = best.value; out["last_after"] = np.array(last[:])
    fb = (C.c_double * 4)(); r.ref_get_feedback(fb); out["feedback"] = np.array(fb[:])
    lp = (C.c_double * 4)(); r.ref_get_local_pose(lp); out["local_pose_after"] = np.array(lp[:])
    out["local_after_nonzero"] = int(np.count_nonzero(
        np.ctypeslib.as_array(r.ref_map_local_ptr(), shape=(LX * LY * LZ,))))
    return out


def streams(r, P, seed):
    noise = np.empty((P, 4), np.float32)
    r.ref_noise_stream(C.c_uint(seed), P, noise.ctypes.data_as(C.c_void_p))
    u = np.empty(P, np.float64)
    r.ref_uniform_stream(C.c_uint(seed + 1), P, u.ctypes.data_as(C.c_void_p))
    return noise, u


def sparse(a, base):
    idx = np.flatnonzero(a.reshape(-1) != base.reshape(-1)).astype(np.int32)
    return idx, a.reshape(-1)[idx].copy()


def gen_helpers(r, rng):
    xyz = rng.integers(-50, 350, (200, 3)).astype(np.int32)
    flat_p = np.array([r.ref_index_conv_p(int(a), int(b), int(c)) for a, b, c in xyz], np.int32)
    flat_l = np.array([r.ref_index_conv(int(a), int(b), int(c)) for a, b, c in xyz], np.int32)
    js = rng.integers(0, LX * LY * LZ, 200).astype(np.int32)
    rev = np.empty((200, 3), np.int32); tmp = (C.c_int * 3)()
    for k, j in enumerate(js):
        r.ref_reverse_index_conv(tmp, int(j)); rev[k] = tmp[:]
    ang = rng.uniform(-3.2, 3.2, (50, 3))
    ang[:10, 1:] = 0.0
    mats = np.empty((50, 9), np.float64)
    for k in range(50):
        r.ref_rot_matrix(mats[k].ctypes.data_as(C.c_void_p), *[float(v) for v in ang[k]])
    pts = rng.integers(-120, 120, (50, 3)).astype(np.int32)
    rot = np.empty((50, 3), np.float64)
    for k in range(50):
        r.ref_rotate(rot[k].ctypes.data_as(C.c_void_p), pts[k].ctypes.data_as(C.c_void_p),
                     mats[k].ctypes.data_as(C.c_void_p))
    np.savez_compressed(os.path.join(OUT, "helpers.npz"), xyz=xyz, flat_p=flat_p, flat_l=flat_l,
                        js=js, rev=rev, ang=ang, mats=mats, pts=pts, rot=rot)


def gen_resample(r, rng):
    """resample() alone (S3P:757-821): weights set by the harness, particles tagged through
    pos_z, thresholds = the libstdc++ uniform stream of the seed the reference was handed."""
    cases = {}
    k = 0
    for P in (50, 100):
        r.ref_alloc_particles(P)
        for kind in ("flat", "skewed", "zeros", "onehot_like", "unnormalised"):
            for seed in (11 + k, 1011 + k):
                k += 1
                if kind == "flat":
                    w = np.full(P, 1.0 / P)
                elif kind == "skewed":
                    w = rng.random(P) ** 6; w /= w.sum()
                elif kind == "zeros":
                    w = rng.random(P); w[rng.random(P) < 0.4] = 0.0; w /= w.sum()
                elif kind == "onehot_like":
                    w = np.full(P, 1e-9); w[P // 3] = 1.0; w /= w.sum()
                else:
                    w = rng.random(P); w /= w.sum() * 0.999999   # sums slightly above 1
                st = np.zeros((P, 5)); st[:, 2] = np.arange(P); st[:, 4] = w
                set_particles(r, st)
                prev_best = int(rng.integers(0, P))
                r.ref_set_policy(0, 0.0, 0.0, 0.0, 0.0, prev_best)
                r.ref_set_seed(seed)
                r.ref_resample()
                after = get_particles(r, P)
                add = C.c_int(); best = C.c_int(); last = (C.c_double * 4)()
                r.ref_get_policy(C.byref(add), last, C.byref(best))
                u = np.empty(P, np.float64)
                r.ref_uniform_stream(C.c_uint(seed), P, u.ctypes.data_as(C.c_void_p))
                # keep only cases the reference defines: every draw strictly inside an interval
                cdf = np.concatenate([[0.0], np.cumsum(w)])  # screening only, not the fixture
                if not np.all((u > 0) & (u < cdf[-1] * (1 - 1e-12))):
                    continue
                name = f"P{P}_{kind}_s{seed}"
                cases[name + "_w"] = w; cases[name + "_u"] = u
                cases[name + "_slot_anc"] = after[:, 2].astype(np.int32)
                cases[name + "_w_after"] = after[:, 4]
                cases[name + "_best"] = np.int32(best.value)
                cases[name + "_prev_best"] = np.int32(prev_best)
    np.savez_compressed(os.path.join(OUT, "resample.npz"), **cases)
    print("resample cases:", len([c for c in cases if c.endswith("_u")]))


def gen_steps(r, rng):
    """Single calls of particle_map() (S3P:827-962)."""
    world = world_map(rng)
    true_pose = np.array([7.5, 7.5, 1.5, 0.3])
    for P, add, seed, wkind in ((50, 1, 101, "flat"), (50, 0, 202, "flat"), (100, 1, 303, "rand"),
                                (100, 0, 404, "rand")):
        r.ref_alloc_particles(P)
        local = local_scan(world, true_pose, rng, n_occ=900, n_free=250, n_weak=80)
        st = np.empty((P, 5))
        st[:, 0] = true_pose[0] + rng.normal(0, 0.06, P)
        st[:, 1] = true_pose[1] + rng.normal(0, 0.06, P)
        st[:, 2] = true_pose[2] + rng.normal(0, 0.03, P)
        st[:, 3] = true_pose[3] + rng.normal(0, 0.05, P)
        if wkind == "flat":
            st[:, 4] = 1.0 / P
        else:
            w = rng.random(P) + 0.1; st[:, 4] = w / w.sum()
        local_pose = np.array([5.12, 4.93, 1.52, 0.04])
        global_pose = np.array([7.9, 7.4, 1.5, 0.33])
        policy = dict(add=add, last=[7.5, 7.5, 1.5, 0.0], best=int(rng.integers(0, P)))
        out = run_step(r, P, world, st, local, local_pose, global_pose, policy, seed)
        noise, u = streams(r, P, seed)
        crcs = np.array([zlib.crc32(pmap(r, i).tobytes()) for i in range(P)], np.uint32)
        d = {}
        for i in (0, 1, P - 1):
            idx, val = sparse(pmap(r, i).copy(), world)
            d[f"map{i}_idx"] = idx; d[f"map{i}_val"] = val
        lidx = np.flatnonzero(local.reshape(-1)).astype(np.int32)
        np.savez_compressed(
            os.path.join(OUT, f"step_P{P}_add{add}.npz"),
            P=np.int32(P), add=np.int32(add), seed=np.int32(seed),
            world=world, local_idx=lidx, local_val=local.reshape(-1)[lidx],
            state_in=st, local_pose=local_pose, global_pose=global_pose,
            last_in=np.array(policy["last"]), best_in=np.int32(policy["best"]),
            noise=noise, u=u, w_pre=out["w_pre"], state_out=out["state"],
            add_after=np.int32(out["add_after"]), best=np.int32(out["best"]),
            last_after=out["last_after"], feedback=out["feedback"],
            local_pose_after=out["local_pose_after"],
            local_after_nonzero=np.int32(out["local_after_nonzero"]), map_crc=crcs, **d)
        print(f"step P={P} add={add}: beams={len(lidx)} best={out['best']} "
              f"max w_pre={out['w_pre'].max():.4g}")


def gen_step_trig(r):
    """One particle_map() call (S3P:827-962) whose particles all sit at yaws where glibc's sincos() and
    sin()/cos() disagree in the last place (own RNG; written on its own by `make_golden.py step_trig`): g++ -O2
    takes the rotation of S3P:860 and the predict of S3P:911-912 from sincos(), and the local-frame motion is
    large enough for that last place to show in the predicted poses."""
    libm = C.CDLL("libm.so.6")
    libm.cos.restype = libm.sin.restype = C.c_double
    libm.cos.argtypes = libm.sin.argtypes = [C.c_double]
    libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rng = np.random.default_rng(7900)
    world = world_map(rng)
    true_pose = np.array([7.5, 7.5, 1.5, 0.3])
    P, add, seed = 50, 1, 505
    yaws = []
    while len(yaws) < P:
        a = float(true_pose[3] + rng.normal(0, 0.05))
        sv, cv = C.c_double(), C.c_double()
        libm.sincos(a, C.byref(sv), C.byref(cv))
        if sv.value != libm.sin(a) or cv.value != libm.cos(a):
            yaws.append(a)
    r.ref_alloc_particles(P)
    local = local_scan(world, true_pose, rng, n_occ=900, n_free=250, n_weak=80)
    st = np.empty((P, 5))
    st[:, 0] = true_pose[0] + rng.normal(0, 0.06, P)
    st[:, 1] = true_pose[1] + rng.normal(0, 0.06, P)
    st[:, 2] = true_pose[2] + rng.normal(0, 0.03, P)
    st[:, 3] = yaws
    st[:, 4] = 1.0 / P
    local_pose = np.array([9.63, 0.71, 1.52, 0.04])          # a long local-frame motion: the last place of c, s reaches the poses
    global_pose = np.array([7.9, 7.4, 1.5, 0.33])
    policy = dict(add=add, last=[7.5, 7.5, 1.5, 0.0], best=int(rng.integers(0, P)))
    out = run_step(r, P, world, st, local, local_pose, global_pose, policy, seed)
    noise, u = streams(r, P, seed)
    crcs = np.array([zlib.crc32(pmap(r, i).tobytes()) for i in range(P)], np.uint32)
    d = {}
    for i in (0, 1, P - 1):
        idx, val = sparse(pmap(r, i).copy(), world)
        d[f"map{i}_idx"] = idx; d[f"map{i}_val"] = val
    lidx = np.flatnonzero(local.reshape(-1)).astype(np.int32)
    np.savez_compressed(
        os.path.join(OUT, "step_P50_trig.npz"),
        P=np.int32(P), add=np.int32(add), seed=np.int32(seed),
        world=world, local_idx=lidx, local_val=local.reshape(-1)[lidx],
        state_in=st, local_pose=local_pose, global_pose=global_pose,
        last_in=np.array(policy["last"]), best_in=np.int32(policy["best"]),
        noise=noise, u=u, w_pre=out["w_pre"], state_out=out["state"],
        add_after=np.int32(out["add_after"]), best=np.int32(out["best"]),
        last_after=out["last_after"], feedback=out["feedback"],
        local_pose_after=out["local_pose_after"],
        local_after_nonzero=np.int32(out["local_after_nonzero"]), map_crc=crcs, **d)
    print(f"step_trig P={P}: beams={len(lidx)} best={out['best']}")


def gen_run(r, rng, P=100, steps=200):
    """BASELINE.json configs[0]: P=100, ~1k-beam scans, 200 consecutive steps of the
    reference's particle_map(); maps and particles carry over from step to step."""
    world = world_map(rng)
    r.ref_alloc_particles(P)
    for i in range(P):
        pmap(r, i)[:] = world.reshape(-1)
    st = np.empty((P, 5)); st[:, 0] = 7.5; st[:, 1] = 7.5; st[:, 2] = 1.5; st[:, 3] = 0.0
    st[:, 4] = 1.0 / P
    set_particles(r, st)
    r.ref_set_policy(1, 7.5, 7.5, 1.5, 0.0, 0)
    true = np.array([7.5, 7.5, 1.5, 0.0])
    gpose = true.copy()
    rec = dict(local_idx=[], local_val=[], local_off=[0], local_pose=[], global_pose=[], noise=[],
               u=[], best=[], add_after=[], feedback=[], w_pre_crc=[], state_crc=[], w_pre_max=[])
    seed = 5000
    buf = C.create_string_buffer(1 << 22)
    for t in range(steps):
        ang = 2 * np.pi * t / steps
        step_xy = 0.06 * np.array([np.cos(ang), np.sin(ang)])
        dyaw = 0.012
        new_true = true + np.array([step_xy[0], step_xy[1], 0.0, dyaw])
        # motion since the last step, expressed in the frame the local map was started in
        c, s = np.cos(true[3]), np.sin(true[3])
        dl = np.array([c * step_xy[0] + s * step_xy[1], -s * step_xy[0] + c * step_xy[1], 0.0])
        local = local_scan(world, true, rng, n_occ=850, n_free=150, n_weak=50)
        local_pose = np.array([5 + dl[0], 5 + dl[1], 1.5 + dl[2], dyaw])
        gpose = new_true + rng.normal(0, 0.01, 4)
        np.ctypeslib.as_array(r.ref_map_local_ptr(), shape=(LX * LY * LZ,))[:] = local.reshape(-1)
        r.ref_set_local_pose(*[float(v) for v in local_pose])
        r.ref_set_global_pose(*[float(v) for v in gpose])
        r.ref_set_seed(seed)
        r.ref_particle_map(buf, len(buf))
        w_pre = parse_weights(buf.value.decode())
        noise, u = streams(r, P, seed)
        seed += 2
        add = C.c_int(); best = C.c_int(); last = (C.c_double * 4)()
        r.ref_get_policy(C.byref(add), last, C.byref(best))
        fb = (C.c_double * 4)(); r.ref_get_feedback(fb)
        lidx = np.flatnonzero(local.reshape(-1)).astype(np.int32)
        rec["local_idx"].append(lidx); rec["local_val"].append(local.reshape(-1)[lidx])
        rec["local_off"].append(rec["local_off"][-1] + len(lidx))
        rec["local_pose"].append(local_pose); rec["global_pose"].append(gpose)
        rec["noise"].append(noise); rec["u"].append(u)
        rec["best"].append(best.value); rec["add_after"].append(add.value)
        rec["feedback"].append(np.array(fb[:]))
        rec["w_pre_crc"].append(zlib.crc32(w_pre.tobytes())); rec["w_pre_max"].append(w_pre.max())
        rec["state_crc"].append(zlib.crc32(get_particles(r, P).tobytes()))
        true = new_true
        if t % 20 == 0:
            print(f"  run step {t}: beams={len(lidx)} best={best.value} add={add.value}", flush=True)
    final = get_particles(r, P)
    crcs = np.array([zlib.crc32(pmap(r, i).tobytes()) for i in range(P)], np.uint32)
    np.savez_compressed(
        os.path.join(OUT, "run_P100_200steps.npz"), P=np.int32(P), steps=np.int32(steps),
        world=world,
        local_idx=np.concatenate(rec["local_idx"]), local_val=np.concatenate(rec["local_val"]),
        local_off=np.array(rec["local_off"], np.int64),
        local_pose=np.array(rec["local_pose"]), global_pose=np.array(rec["global_pose"]),
        noise=np.array(rec["noise"]), u=np.array(rec["u"]),
        best=np.array(rec["best"], np.int32), add_after=np.array(rec["add_after"], np.int32),
        feedback=np.array(rec["feedback"]), w_pre_crc=np.array(rec["w_pre_crc"], np.uint32),
        w_pre_max=np.array(rec["w_pre_max"]), state_crc=np.array(rec["state_crc"], np.uint32),
        state_final=final, map_crc=crcs)


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/libs3p_ref.so missing: run oracle/build_ref.sh (needs /root/reference)")
    os.makedirs(OUT, exist_ok=True)
    r = load_ref()
    assert r.ref_p_map_elements() == PX * PY * PZ and r.ref_map_elements() == LX * LY * LZ
    if len(sys.argv) > 1 and sys.argv[1] == "step_trig":    # the later addition, on its own: nothing else is rewritten
        gen_step_trig(r)
        return
    rng = np.random.default_rng(20261111)
    gen_helpers(r, rng)
    gen_resample(r, rng)
    gen_steps(r, rng)
    if "--no-run" not in sys.argv:
        gen_run(r, rng)
    gen_step_trig(r)
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()

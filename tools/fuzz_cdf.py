#!/usr/bin/env python3
"""Development aid: the sequential-order CDF forms (pf_sequential_cdf how = 1 one workgroup, how = 2 many workgroups)
against numpy's sequential cumsum, bit for bit, on random weight vectors: sizes from 1 to 600 000 (on and beside the
segment boundaries 1024, 2048, 4096, 8192 and multiples of 8192), normalised and raw, flat, log-normal over many binades,
ties (multiples of a power of two near the sum's ulp), runs of zeros, a few huge or negative or non-finite elements.
    python tools/fuzz_cdf.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import capi

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
h = capi.Handle(capi.default_config(device=0, n_particles=64, n_particles_global=64, particle_offset=0,
                                    map_mode=capi.PF_MAP_SHARED, grid_x=16, grid_y=16, grid_z=16,
                                    local_x=0, local_y=0, local_z=0, max_beams=8, max_points=8))
edges = [1024, 2048, 4096, 8192, 16384, 24576, 65536, 262144]
bad = 0
for case in range(cases):
    rng = np.random.default_rng(seed0 * 104729 + case)
    r = rng.random()
    if r < 0.3:
        n = int(rng.choice(edges)) + int(rng.integers(-3, 4))
    elif r < 0.8:
        n = int(rng.integers(1, 120000))
    else:
        n = int(rng.integers(120000, 600000))
    n = max(n, 1)
    kind = int(rng.integers(0, 7))
    if kind == 0:
        w = rng.random(n) + 1e-3
    elif kind == 1:
        w = np.full(n, 1.0)
    elif kind == 2:
        w = np.exp(rng.normal(0, float(rng.choice([0.5, 3.0, 12.0])), n))
    elif kind == 3:                                   # ties: integers of a power of two
        w = rng.integers(0, 1 << int(rng.integers(1, 14)), n).astype(np.float64)
    elif kind == 4:                                   # runs of zeros
        w = rng.random(n) * (rng.random(n) < 0.2)
    elif kind == 5:                                   # a slowly growing ramp: binade crossings spread over the segments
        w = np.linspace(1.0, float(rng.choice([2.0, 50.0, 1e4])), n) * (1 + 1e-3 * rng.random(n))
    else:
        w = rng.random(n) ** 8
    if rng.random() < 0.7:
        s = w.sum()
        if s > 0:
            w = w / s
    if kind == 3:
        w = w * 2.0 ** -int(rng.integers(40, 60))
    if rng.random() < 0.15:                           # elements the scan cannot take
        for _ in range(int(rng.integers(1, 4))):
            w[int(rng.integers(0, n))] = float(rng.choice([-1e-9, np.nan, np.inf, 1e300, -np.inf, 7.0]))
    with np.errstate(all="ignore"):
        ref = np.concatenate([[0.0], np.cumsum(w)])
    for how in (1, 2):
        got = h.sequential_cdf(w, how=how)
        nan = np.isnan(ref)
        ok = np.array_equal(np.isnan(got), nan) and np.array_equal(got.view(np.uint64)[~nan], ref.view(np.uint64)[~nan])
        if not ok:
            bad += 1
            d = np.flatnonzero((got.view(np.uint64) != ref.view(np.uint64)) & ~(nan & np.isnan(got)))
            print(f"case {case}: n {n} kind {kind} how {how}: first difference at {d[:3]}", flush=True)
print(f"{cases} cases, {bad} mismatches")
h.close()

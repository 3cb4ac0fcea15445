#!/usr/bin/env python3
"""TEST INFRASTRUCTURE. Writes tests/golden/local_map_8frames_beams.npz: the beam list (S3P:868-877: the
non-zero cells of the local map minus (100, 100, 30), x, y, z order, with their int8 values) that the
reference's class Slam hands to its first particle_map() in tests/golden/frames.npz, i.e. its local map
after 8 callback() frames. The frames are replayed through the oracle's mapper; golden_util.check_frames
asserts after every frame that the local map's CRC equals the one the reference itself produced
(frames.npz, key local_crc), so the list is the reference's own, bit for bit. Data only.
    python oracle/make_dense_beams.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from slam_amd import capi
import golden_util as G
from oracle_backend import OracleLocHandle, OracleMapper

mapper = OracleMapper()
G.check_frames(os.path.join(ROOT, "tests", "golden"), OracleLocHandle(), mapper, None, frames=8)
local = mapper.read_local_map()
bx, by, bz, bv = capi.snapshot_beams(local.reshape(200, 200, 80), (100, 100, 30))
out = os.path.join(ROOT, "tests", "golden", "local_map_8frames_beams.npz")
np.savez_compressed(out, bx=bx.astype(np.int16), by=by.astype(np.int16), bz=bz.astype(np.int16), bv=bv.astype(np.int8))
print(out, len(bx), "beams,", int((bv > 20).sum()), "occupied,", os.path.getsize(out), "bytes")

"""GPU suite: the C++ host driver (driver/step_loop.cpp, the stand-in for the reference's main() +
Slam::callback() loop over the C ABI) replays the reference's own run. tests/golden/frames.npz holds 48
consecutive callback() frames and 6 particle_map() calls of the reference's class Slam (generator:
oracle/make_golden_frontend.py): the pooled depth images, the IMU angles, the seeds of every random draw,
and after every call the global pose, the best particle, the add_points policy and the correction
feedback. The driver is given the inputs as a replay file and must print the same numbers, bit for bit
(17 significant digits round-trip a double). This pins the driver's call order and host arithmetic the
way tests/test_gpu_frontend.py pins the Python mirror."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["replay", "replay-device"])
def test_cpp_driver_replays_the_reference_run(golden_dir, tmp_path, mode):
    """mode replay-device: the same run with the cloud kept in a device buffer between the stages (the pf_*_device
    calls, both handles on one stream)."""
    exe = os.path.join(ROOT, "driver", "step_loop")
    assert os.path.exists(exe), "driver/step_loop is not built (python -c 'import __graft_entry__ as g; g.build()')"
    g = np.load(os.path.join(golden_dir, "frames.npz"))
    F, per_step, P = int(g["frames"]), int(g["per_step"]), int(g["P"])
    path = tmp_path / "frames.replay"
    with open(path, "wb") as fp:
        fp.write(struct.pack("<iii", F, per_step, P))
        fp.write(np.ascontiguousarray(g["pooled"], "<u2").tobytes())
        fp.write(np.ascontiguousarray(g["gyr"], "<f8").tobytes())
        fp.write(np.ascontiguousarray(g["seed_vote"]).astype("<u4").tobytes())
        fp.write(np.ascontiguousarray(g["seed_rays"]).astype("<u4").tobytes())
        fp.write(np.ascontiguousarray(g["step_seed"]).astype("<u4").tobytes())
    out = subprocess.run([exe, mode, str(path)], capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stderr[-500:]
    assert "callback(): 48 frames" in out.stderr
    frames, steps = {}, {}
    for line in out.stdout.splitlines():
        t = line.split()
        if t[:1] == ["frame"]:
            frames[int(t[1])] = np.array([float(v) for v in t[3:7]])
        elif t[:1] == ["step"] and "feedback" in t:
            steps[int(t[1])] = (int(t[3]), int(t[5]), np.array([float(v) for v in t[7:11]]))
    assert sorted(frames) == list(range(F)) and sorted(steps) == list(range(F // per_step))
    for f in range(F):
        assert np.array_equal(frames[f], g["gpose"][f]), (f, frames[f], g["gpose"][f])
    for k in range(F // per_step):
        best, add, fb = steps[k]
        assert best == int(g["step_best"][k]) and add == int(g["step_add_after"][k]), k
        assert np.array_equal(fb, g["step_feedback"][k]), (k, fb, g["step_feedback"][k])

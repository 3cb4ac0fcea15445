import os, struct, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(root, "tests", "golden", "frames.npz"))
F, per_step, P = int(g["frames"]), int(g["per_step"]), int(g["P"])
with open(sys.argv[1], "wb") as fp:
    fp.write(struct.pack("<iii", F, per_step, P))
    fp.write(np.ascontiguousarray(g["pooled"], "<u2").tobytes())
    fp.write(np.ascontiguousarray(g["gyr"], "<f8").tobytes())
    fp.write(np.ascontiguousarray(g["seed_vote"]).astype("<u4").tobytes())
    fp.write(np.ascontiguousarray(g["seed_rays"]).astype("<u4").tobytes())
    fp.write(np.ascontiguousarray(g["step_seed"]).astype("<u4").tobytes())

#!/usr/bin/env python3
"""Development aid: ms per step and K1's average launch time of bench.py output files."""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"] * 1e3, 1), "us/step; K1", round(d["roofline"]["avg_launch_us"], 2), "us")

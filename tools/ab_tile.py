#!/usr/bin/env python3
"""A/B in one call: bench.py's frame rate over fixed tile heights of the compacted convs
(LCF_OPT_CONV_TILE; 0 = the library's choice), alternating.  Usage: tools/ab_tile.py [heights]"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from objectdetection_amd import _lib  # noqa: E402

tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "0,16,20,24,28,32").split(",")]
args = ["--steps", "400", "--warmup", "30", "--train-steps", "0", "--no-cpu-baseline", "--no-alt"]
for rep in range(2):
    for t in tiles:
        _lib.lib().lcf_set_option(_lib.OPT_CONV_TILE, t)
        buf = io.StringIO()
        with redirect_stdout(buf):
            bench.main(args)
        d = json.loads(buf.getvalue().strip().splitlines()[-1])
        print("tile", t, d["value"], d["ms_per_step"], d["roofline"]["frac"], flush=True)

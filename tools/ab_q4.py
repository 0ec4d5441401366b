#!/usr/bin/env python3
"""A/B in one call: bench.py's frame rate with the 3x3x3 convs of 32 / 64 channels on the
tile-compacted kernel (LCF_OPT_CONV_IMPL = 2) against the default selection (the quad-row kernel,
csrc/conv_q4.h), alternating.  Usage: tools/ab_q4.py"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from objectdetection_amd import _lib  # noqa: E402

args = ["--steps", "400", "--warmup", "30", "--train-steps", "0", "--no-cpu-baseline", "--no-alt"]
for rep in range(3):
    for impl in (2, 0):
        _lib.lib().lcf_set_option(_lib.OPT_CONV_IMPL, impl)
        buf = io.StringIO()
        with redirect_stdout(buf):
            bench.main(args)
        d = json.loads(buf.getvalue().strip().splitlines()[-1])
        print("impl", impl, d["value"], d["ms_per_step"], d["roofline"]["kernel"],
              d["roofline"]["frac"], flush=True)

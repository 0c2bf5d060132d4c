#!/usr/bin/env python
"""What does the decode step's launch CHAIN cost before any kernel moves a byte?  Replays hipGraphs of N dependent launches of a
(nearly) empty kernel at the step's launch count (171 at batch 256, 123 at batch 1) and typical grid shapes.
    python tools/launch_floor.py      (through gpurun)"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
from neutts import _hip  # noqa: E402

lib = _hip.load_library()


def chain(n, grid, block, iters=200):
    us = C.c_double()
    rc = lib.ntts_k_launch_chain_probe(n, grid, block, iters, C.byref(us))
    assert rc == 0, rc
    return us.value


for n, grid, block, what in ((171, 256, 256, "batch-256 step: 171 launches of a kernel that reads 4 bytes per workgroup, 256 workgroups"),
                             (171, 512, 256, "... 512 workgroups (the attention grid)"),
                             (171, 152, 256, "... 152 workgroups"),
                             (123, 152, 256, "batch-1 step: 123 launches, 152 workgroups"),
                             (123, 16, 256, "... 16 workgroups"),
                             (1, 256, 256, "a single launch per replay"),
                             (171, 256, -256, "171 launches, each thread: one HBM-cold 16-byte load -> one 16-byte store (1 MB per launch)"),
                             (171, 2048, -256, "... 8 MB per launch"),
                             (123, 64, -256, "123 launches, 64 workgroups, load -> store")):
    us = chain(n, grid, block)
    print(json.dumps({"launches": n, "grid": grid, "touch": block < 0, "us_per_replay": round(us, 1), "us_per_launch": round(us / n, 3), "what": what}), flush=True)

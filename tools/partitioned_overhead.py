#!/usr/bin/env python3
"""Host-side cost of the phase-split stage: the same 32^3 P4 case on ONE rank through hfx_run_steps (one call per
run) and through the five hfx_stage_partitioned phases per stage that the N>1 path uses (no partition faces here, so
the difference is the extra calls and launches, not communication)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hifiles-solver_amd"))
import ctypes as C

import hfx
import hfx_host as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for label in ("run_steps", "partitioned"):
    c = H.Case(n, order=4)
    c.to_device(0)
    ctx, e, faces, nb = c.handles()
    lib = hfx.lib()
    hfx.check(lib.hfx_ctx_set_fused_mode(ctx, C.c_int(3)))

    def run(k):
        if label == "run_steps":
            hfx.check(lib.hfx_run_steps(e, faces, C.c_int(nb), C.c_int(k), C.c_int(3)))
        else:
            c.run_partitioned(k)
    run(2)
    c.synchronize()
    t0 = time.perf_counter()
    run(20)
    c.synchronize()
    dt = time.perf_counter() - t0
    print("%-12s %.4f ms/stage" % (label, 1e3 * dt / 100))
    c.close()

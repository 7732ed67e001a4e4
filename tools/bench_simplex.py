#!/usr/bin/env python3
"""BASELINE.json configs[3]'s element families at bench size: P3 tetrahedra / triangular prisms through the general
(dense FP64 MFMA) per-method path.

The host mirror has no setup for these classes, so the workload is built from a fixture of the genuine reference
(tests/golden/{tet,pri}_p3_n2_deformed.npz: a small periodic box with its operators, metrics and face tables) tiled
`--tiles` times: identical, mutually disconnected periodic boxes -- the same per-element and per-face work as one
large mesh, every array at full size in HBM.  Prints one line per case: DOF-updates/s, ms per RK stage and the
per-method times with the effective HBM rate of the slowest contraction.

    python tools/bench_simplex.py [--tiles 2048] [--steps 4]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hifiles-solver_amd"))
sys.path.insert(0, ROOT)
import hfx  # noqa: E402
from bench import METHOD_NAMES  # noqa: E402


def tiled(d, tiles):
    """the fixture's element and face arrays repeated `tiles` times along the element / face axis"""
    ne, nu, nfp, nf, nd = [int(v) for v in d["sizes"][:5]]
    out = dict(d)
    ele_axis = {"detjac_upts": 1, "JGinv_upts": 3, "detjac_fpts": 1, "JGinv_fpts": 3, "tdA_fpts": 1, "norm_fpts": 1, "u_init": 1}
    for k, ax in ele_axis.items():
        out[k] = np.asfortranarray(np.concatenate([d[k]] * tiles, axis=ax))
    faces = []
    for t in range(3):
        if "int%d_L" % t in d:
            L, R = d["int%d_L" % t].astype(np.int64), d["int%d_R" % t].astype(np.int64)
            # tile-major: the faces of one tile stay together in the list (bench.tile_table)
            Lb = np.concatenate([L + k * (nfp * ne) for k in range(tiles)], axis=1)
            Rb = np.concatenate([R + k * (nfp * ne) for k in range(tiles)], axis=1)
            faces.append((np.asfortranarray(Lb.astype(np.int32)), np.asfortranarray(Rb.astype(np.int32))))
    return out, faces, ne * tiles


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--fused", type=int, default=4, help="4: the general fused stage (csrc/general.hip), 0: per method")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE")
    args = ap.parse_args()
    ctx = hfx.Context(0)
    for kv in args.opt:
        ctx.set_option(*kv.split("="))
    lib = hfx.lib()
    for name in ("tet_p3_n2_deformed", "pri_p3_n2_deformed"):
        d = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
        big, face_tabs, ne = tiled(d, args.tiles)
        sz = [int(v) for v in d["sizes"]]
        ctx.set_params(hfx.params_from(d))
        e = hfx.Eles(ctx, [ne] + sz[1:5], big, ele_type=sz[6], order=sz[5])
        faces = [hfx.IntInters(ctx, e, e, L, R) for L, R in face_tabs]
        e.upload(hfx.DISU_UPTS0, big["u_init"])
        hfx.run_steps(e, faces, 1, fused=args.fused)
        ctx.synchronize()
        t0 = time.perf_counter()
        hfx.run_steps(e, faces, args.steps, fused=args.fused)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        assert e.check_nan() == -1
        n_stage = sz[7]
        dof = ne * sz[1] * sz[3]
        fa = (C.c_void_p * len(faces))(*[f.h for f in faces])
        if args.fused == 4:
            ms = (C.c_double * 8)()
            names = (C.c_char * 256)()
            ea = (C.c_void_p * 1)(e.h)
            hfx.check(lib.hfx_time_general_kernels(ea, C.c_int(1), fa, C.c_int(len(faces)), C.c_int(10), ms, names))
            by = (C.c_double * 8)()
            hfx.check(lib.hfx_general_kernel_bytes(ea, C.c_int(1), by))
            times = {n: round(ms[i], 4) for i, n in enumerate(names.value.decode().split(","))}
            times["GBps"] = {n: round(by[i] / (ms[i] * 1e-3) / 1e9, 1) for i, n in enumerate(names.value.decode().split(",")) if ms[i] > 0}
        else:
            ms = (C.c_double * 11)()
            hfx.check(lib.hfx_time_methods(e.h, fa, C.c_int(len(faces)), C.c_int(5), ms))
            times = {n: round(ms[i], 4) for i, n in enumerate(METHOD_NAMES)}
        print(json.dumps({"case": name, "n_eles": ne, "n_upts": sz[1], "n_fpts": sz[2], "dof": dof,
                          "ms_per_rk_stage": round(1e3 * dt / (args.steps * n_stage), 4),
                          "dof_updates_per_s": dof * n_stage * args.steps / dt, "methods_ms": times}))
        for f in faces:
            f.close()
        e.close()
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Full-size parity fixture -- TEST INFRASTRUCTURE, container only.

Runs the GENUINE reference (oracle/_ref/ref_harness) on BASELINE.json configs[1] as it stands -- the Taylor-Green
vortex on the generated periodic 32^3 hexahedral mesh, P4, Navier-Stokes, HLLC + LDG, RK45 (about 150 s and 6 GB here)
-- for two time steps and stores a COMPACT fixture, tests/golden/hex_p4_n32_tgv.npz:

  * the reference's own solution-point abscissae (`loc_1d_upts`, the data/JacobiGQ.bin row the reference reads; it is
    not bit-symmetric, and on an axis-aligned mesh the LDG switch of src/inters.cpp:568-581,620-633 is decided by the
    rounding noise of the metrics built from it -- the host mirror takes these values through hfxh_case_desc.loc_1d_upts),
  * per field the L1 / L2 / Linf norms of disu_upts(0) and div_tconf_upts(0) after step 1 and step 2,
  * the full arrays of a fixed sample of 256 elements (box corners, edges, faces, interior).

    python oracle/capture_fullsize.py [n_cells]
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from capture_golden import BASE, GOLDEN, HARNESS, REF_HOME, read_dump  # noqa: E402
from gen_neu_mesh import write_neu  # noqa: E402


def sample_elements(n, count=256):
    """Element ids (ele = ix + n*(iy + n*iz), the generator's order): the 8 corners, points on edges and faces, and a
    fixed pseudo-random interior set."""
    ids = set()
    ends = (0, n - 1)
    for a in ends:
        for b in ends:
            for c in ends:
                ids.add(a + n * (b + n * c))
    mid = (n // 3, n // 2, (2 * n) // 3)
    for a in ends:
        for b in ends:
            for m in mid:
                ids.update((a + n * (b + n * m), a + n * (m + n * b), m + n * (a + n * b)))
    for a in ends:
        for m1 in mid:
            for m2 in mid:
                ids.update((a + n * (m1 + n * m2), m1 + n * (a + n * m2), m1 + n * (m2 + n * a)))
    rng = np.random.RandomState(0x48694669 & 0x7fffffff)
    while len(ids) < min(count, n ** 3):
        ids.add(int(rng.randint(0, n ** 3)))
    return np.array(sorted(ids), dtype=np.int32)


def norms(a):
    """(3, n_fields): L1 sum, L2 sum of squares, Linf over all points and elements, per field"""
    return np.stack([np.abs(a).sum(axis=(0, 1)), (a * a).sum(axis=(0, 1)), np.abs(a).max(axis=(0, 1))])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    order, steps = 4, 2
    keys = dict(BASE, order=order, n_steps=steps)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        # the 1-D abscissae: a tiny level-1 run dumps loc_upts
        write_neu(os.path.join(td, "mesh.neu"), 3, 3, amp=0.0)
        with open(os.path.join(td, "input"), "w") as f:
            for k, v in keys.items():
                f.write("%s %s\n" % (k, repr(v) if isinstance(v, float) else v))
        env = dict(os.environ, HIFILES_HOME=REF_HOME)
        subprocess.run([HARNESS, "input", "small.bin", "0", "1"], cwd=td, env=env, check=True, capture_output=True)
        small = read_dump(os.path.join(td, "small.bin"))
        out["loc_1d_upts"] = small["loc_upts"][0, :order + 1].copy()
        os.remove(os.path.join(td, "small.bin"))
        # the full-size run
        write_neu(os.path.join(td, "mesh.neu"), n, 3, amp=0.0)
        env["HFX_DUMP_DIV"] = "1"
        r = subprocess.run([HARNESS, "input", "dump.bin", str(steps), "0"], cwd=td, env=env, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout[-3000:] + r.stderr[-3000:])
            raise SystemExit("harness failed")
        sys.stderr.write(r.stderr[-300:])
        arrs = read_dump(os.path.join(td, "dump.bin"))
    sample = sample_elements(n)
    out["sample_eles"] = sample
    out["sizes"] = arrs["sizes"]
    rk = int(arrs["sizes"][7])
    for s in range(steps):
        u = arrs["u_step%d_stage%d" % (s, rk - 1)]
        dv = arrs["div_step%d" % s]
        out["u_norms_step%d" % s] = norms(u)
        out["div_norms_step%d" % s] = norms(dv)
        out["u_sample_step%d" % s] = u[:, sample, :].copy()
        out["div_sample_step%d" % s] = dv[:, sample, :].copy()
    out["u_init_sample"] = arrs["u_init"][:, sample, :].copy()
    meta = dict(name="hex_p4_n%d_tgv" % n, n=n, dims=3, amp=0.0, steps=steps, keys=keys,
                generator="oracle/capture_fullsize.py via oracle/_ref/ref_harness (genuine reference)")
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(GOLDEN, "hex_p4_n%d_tgv.npz" % n)
    np.savez_compressed(path, **out)
    print("%s %.1f kB" % (path, os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()

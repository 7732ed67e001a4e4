"""ASCII restart files (8f-3): the host mirror reads the genuine reference's file and writes the same bytes."""
import json
import os
import re

import numpy as np
import pytest

import hfx_host as H

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


@pytest.mark.parametrize("name", ["hex_p2_restart", "quad_p3_restart"])
def test_restart_round_trip_vs_reference_file(tmp_path, name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    meta = json.loads(bytes(d["meta_json"]).decode())
    text = bytes(d["restart_ascii"]).decode()
    # the reference's own (not bit-symmetric) 1-D nodes are part of the header: build the case on them
    nodes = [float(v) for v in re.search(r"Location of solution points in 1D\n([^\n]*)\n", text).group(1).split()]
    n = meta["n"] if isinstance(meta["n"], list) else [meta["n"]] * meta["dims"]
    c = H.Case(n + [1] * (3 - len(n)), xv=d["xv"], loc_1d_upts=nodes, dims=meta["dims"], order=meta["keys"]["order"])
    (tmp_path / "Rest_000000007_p0000.dat").write_text(text)
    c.read_restart(tmp_path, 7)
    last = int(d["sizes"][7]) - 1
    # 15 significant digits on disk
    assert rel(c.array("disu_upts0"), d["u_step0_stage%d" % last]) < 5e-15
    out = tmp_path / "out"
    out.mkdir()
    c.write_restart(out, 12)
    assert (out / "Rest_000000012_p0000.dat").read_text() == text
    c.close()


def test_restart_projects_between_orders(tmp_path):
    """A restart written at P2 read at P3: opp_r interpolates (src/eles.cpp:709-719); a quadratic field is exact."""
    lo = H.Case(3, order=2, amp=0.1)
    lo.write_restart(tmp_path, 1)
    hi = H.Case(3, order=3, amp=0.1)
    hi.read_restart(tmp_path, 1)
    # both cases evaluate the same analytic initial condition at their own points; interpolation error of a smooth field
    assert rel(hi.array("disu_upts0"), H.Case(3, order=3, amp=0.1).array("disu_upts0")) < 5e-2
    with pytest.raises(Exception):
        hi.read_restart(tmp_path, 99)  # missing file
    lo.close(); hi.close()

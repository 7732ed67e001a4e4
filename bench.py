#!/usr/bin/env python3
"""bench.py -- DOF-updates/s and ms/RK-stage of the HiFiLES hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cells CELLS] [--order P] [--mode split3|split|methods|dense]
                    [--workload tgv|tets|prisms|mixed] [--tiles T]

One "step" is one time step = 5 RK stages (RK45), each stage = CalcResidual + AdvanceSolution
(/root/reference/src/HiFiLES.cpp:201-217) over the whole mesh.  Workload at N=1: BASELINE.json
configs[1], the Taylor-Green vortex on a generated periodic 32^3 hexahedral mesh, P4, Navier-Stokes,
HLLC + LDG, fixed dt; inputs are resident in HBM before the timed region.  The timed region (K steps between
barriers + device synchronisation) is repeated 5 times and the MEDIAN is reported.
For N>1 the periodic box is split into N blocks of 32^3 elements on a process grid (2: 2x1x1, 4: 2x2x1, 8: 2x2x2;
weak scaling), one rank per GPU; every rank advances its block and exchanges partition-face solution and flux
records with its neighbours twice per RK stage, as the reference's mpi_inters do -- by grouped ncclSend / ncclRecv
(RCCL over xGMI) on libhfx's own communication stream, overlapped with the interior face kernels.  Started as
`python -m torch.distributed.run ... bench.py --gpus N` the process is one rank; started plainly as
`python bench.py --gpus N` it launches its N ranks itself as CHILD processes (before anything touches a GPU) and
relays rank 0's line.

Prints ONE JSON line (rank 0).  The oracle under oracle/ is used only for the `cpu_baseline` leg.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "hifiles-solver_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

DEFAULT_FUSED = "split3"  # which fused variant `--mode auto` runs
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector = matrix peak

METHOD_NAMES = ["extrapolate_solution", "calculate_gradient", "evaluate_invFlux", "common_invFlux", "correct_gradient",
                "evaluate_viscFlux", "extrapolate_totalFlux", "calculate_divergence", "common_viscFlux",
                "calculate_corrected_divergence", "AdvanceSolution"]


def method_bytes(c):
    """ALGORITHMIC HBM bytes per element of each per-method entry point (compulsory reads + writes
    of the arrays the method consumes / produces; operators are cache resident)."""
    nu, nfp, nf, nd = c.n_upts, c.n_fpts, c.n_fields, c.n_dims
    d = 8.0
    b = {
        "extrapolate_solution": d * (nu * nf + nfp * nf),
        "calculate_gradient": d * (nu * nf + nu * nf * nd),
        "evaluate_invFlux": d * (nu * nf + nu * nd * nd + nu * nf * nd),
        # per flux point (each belongs to one face side): u, normal (left only -> 1/2), tdA, tconf, delta, 4-byte index
        "common_invFlux": d * nfp * (nf + 0.5 * nd + 1 + nf + nf) + 4 * nfp,
        # opp_5: delta + r/w grad_upts ; opp_6: grad_upts -> grad_fpts ; two in-place transforms with metrics
        "correct_gradient": d * (nfp * nf + 2 * nu * nf * nd + nu * nf * nd + nfp * nf * nd
                                 + 2 * nu * nf * nd + nu * (nd * nd + 1) + 2 * nfp * nf * nd + nfp * (nd * nd + 1)),
        "evaluate_viscFlux": d * (nu * nf + nu * nf * nd + nu * nd * nd + 2 * nu * nf * nd),
        "extrapolate_totalFlux": d * (nu * nf * nd + nfp * nf),
        "calculate_divergence": d * (nu * nf * nd + nu * nf),
        "common_viscFlux": d * nfp * (nf + nf * nd + 0.5 * nd + 1 + 2 * nf) + 4 * nfp,
        "calculate_corrected_divergence": d * (2 * nfp * nf + nfp * nf + 2 * nu * nf),
        "AdvanceSolution": d * (nu * nf + nu + 4 * nu * nf),
    }
    return b


def stage_algorithmic_bytes(c):
    """Compulsory HBM traffic of one fused RK stage per element (generalised):
    state r/w u0,u1; volume metrics; disu_fpts write + own + neighbour read; grad_disu_fpts likewise;
    face metrics."""
    nu, nfp, nf, nd = c.n_upts, c.n_fpts, c.n_fields, c.n_dims
    dbl = 4 * nu * nf + nu * (nd * nd + 1) + 3 * nfp * nf + 3 * nfp * nf * nd + nfp * (nd + 1 + nd * nd + 1)
    return 8.0 * dbl


def cpu_baseline(order, threads):
    """The oracle (oracle/liboracle.so, the CPU restatement pinned against the genuine reference)
    timed on this host: the same TGV case on a bounded sample mesh."""
    import hfx_host as H
    import oracle_py as O
    orc = O.load()
    out = {}
    # bounded sample, about 10-20 s of CPU work in all: 8^3 for one step on one core, 16^3 for 12 steps on all cores
    for label, n_cells, nthr, nsteps in (("1core", 8, 1, 2), ("allcores", 16, threads, 12)):
        case = H.Case(n_cells, order=order)
        reg = case.registration()
        oc = O.Case(reg)
        e = oc.c_eles()
        f, nb = oc.c_faces()
        orc.orc_set_threads(nthr)
        t0 = time.perf_counter()
        for _ in range(nsteps):
            assert orc.orc_rk_step(C.byref(e), f, nb, C.byref(oc.params)) == -1
        dt = time.perf_counter() - t0
        dofs = case.n_eles * case.n_upts * case.n_fields * case.n_stages * nsteps
        out[label] = dict(value=dofs / dt, seconds=dt, n_cells=n_cells, steps=nsteps, threads=nthr)
        case.close()
    orc.orc_set_threads(1)
    return out


def reference_baseline(order, n_cells=8, n_steps=8):
    """The GENUINE reference (oracle/_ref/ref_harness, g++ -O3 of /root/reference/src as it stands, serial, BLAS=NO;
    built by `make -C oracle ref` in the build container and shipped prebuilt) on the same TGV case, reduced mesh:
    wall clock of its own RK loop (setup excluded), printed by the harness.  None where the binary is absent."""
    import re
    import subprocess
    import tempfile
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    harness = os.path.join(ref_dir, "ref_harness")
    if not (os.path.isfile(harness) and os.path.isfile(os.path.join(ref_dir, "data", "JacobiGQ.bin"))):
        return None
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from capture_golden import BASE
    from gen_neu_mesh import write_neu
    keys = dict(BASE, order=order, n_steps=n_steps)
    with tempfile.TemporaryDirectory() as td:
        write_neu(os.path.join(td, "mesh.neu"), n_cells, 3, amp=0.0)
        with open(os.path.join(td, "input"), "w") as f:
            for k, v in keys.items():
                f.write("%s %s\n" % (k, repr(v) if isinstance(v, float) else v))
        r = subprocess.run([harness, "input", "dump.bin", str(n_steps), "0"], cwd=td, env=dict(os.environ, HIFILES_HOME=ref_dir),
                           capture_output=True, text=True)
    m = re.search(r"RK loop (\d+) steps (\d+) stages ([0-9.eE+-]+) s", r.stderr)
    if r.returncode != 0 or not m:
        sys.stderr.write("reference baseline failed: %s\n" % r.stderr[-500:])
        return None
    secs = float(m.group(3))
    n_upts = (order + 1) ** 3
    dofs = n_cells ** 3 * n_upts * 5 * int(m.group(2)) * int(m.group(1))
    return dict(value=dofs / secs, seconds=secs, n_cells=n_cells, steps=int(m.group(1)))


def _min_over_ranks(dist, torch, v):
    t = torch.tensor([v], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t.item()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def rocprof_kernel_ms(mode, kernel):
    """average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this workload"""
    for rnd in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.txt" % (rnd, mode))
        if os.path.exists(path):
            for line in open(path):
                if kernel + "<" in line or kernel + "(" in line:
                    f = [x.strip() for x in line.split("|")]
                    return float(f[3]) * 1e-3, os.path.relpath(path, ROOT)
    return None, None


def tile_arrays(d, tiles, prefix=""):
    """element-wise arrays of one class of a fixture repeated `tiles` times along the element axis"""
    ele_axis = {"detjac_upts": 1, "JGinv_upts": 3, "detjac_fpts": 1, "JGinv_fpts": 3, "tdA_fpts": 1, "norm_fpts": 1, "u_init": 1}
    out = dict(d)
    for k, ax in ele_axis.items():
        out[k] = np.asfortranarray(np.concatenate([d[k]] * tiles, axis=ax))
    return out


def tile_table(T, plane, tiles):
    """a face table (n_fpts_per_inter, n_inters) of offsets into one tile's (fpt, ele) plane, for `tiles` tiles"""
    T = T.astype(np.int64)
    # TILE-MAJOR: column j of the result is face j % n_inters of tile j // n_inters (per-face data of the tiled mesh:
    # np.tile(x, tiles)) -- the faces of one tile stay together, as a mesh generator's face list keeps neighbouring faces
    # together; with the tiles interleaved every face of the list would lie 15 kB from the previous one
    return np.asfortranarray(np.concatenate([T + t * plane for t in range(tiles)], axis=1).astype(np.int32))


def general_workload(args):
    """BASELINE.json configs[3]'s element classes at bench size through the general fused stage (hfx_run_steps_blocks,
    fused = 4): the reference's own fixture of a small mesh -- P3 tetrahedra, P3 prisms, or the MIXED channel (prism
    layers on the walls, tetrahedra in the core, isothermal / adiabatic walls) -- with its operators, metrics and face
    tables, tiled `--tiles` times: identical, mutually disconnected copies, i.e. the per-element and per-face work of one
    large mesh with every array at full size in HBM.  The host mirror has the element classes of tetrahedra and prisms but no
    mesh preprocessor for them (connectivity, face matching: out of scope), which is why the mesh comes from
    a fixture of the genuine reference."""
    import torch  # noqa: F401  (maps torch's HIP runtime first, see tests/conftest.py)
    import hfx
    import mixed_util as MU
    name = {"tets": "tet_p3_n2_deformed", "prisms": "pri_p3_n2_deformed", "mixed": "mixed_p3_channel"}[args.workload]
    d = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
    tiles = args.tiles
    ctx = hfx.Context(0)
    lib = hfx.lib()
    if args.workload == "mixed":
        classes, per, faces, bdy = MU.split(d)
    else:
        classes, per = [int(d["sizes"][6])], None
        per = {classes[0]: d}
        faces = [(classes[0], classes[0], d["int%d_L" % t], d["int%d_R" % t]) for t in range(3) if "int%d_L" % t in d]
        bdy = []
    ctx.set_params(hfx.params_from(per[classes[0]]))
    for kv in args.opt:
        ctx.set_option(*kv.split("="))
    E, plane = {}, {}
    for c in classes:
        sz = [int(v) for v in per[c]["sizes"]]
        plane[c] = sz[0] * sz[2]
        big = tile_arrays(per[c], tiles)
        E[c] = hfx.Eles(ctx, [sz[0] * tiles] + sz[1:5], big, ele_type=sz[6], order=sz[5])
        E[c].upload(hfx.DISU_UPTS0, big["u_init"])
        if args.les_cs >= 0:
            # LES, WALE closure (evaluated inside the general stage's flux kernel): Jacobian_fpts from the host mirror's element class
            import hfx_host as H
            x1 = per[c]["loc_upts"][2, ::(sz[5] + 1) * (sz[5] + 2) // 2] if sz[6] == 3 else None
            S = H.Simplex(sz[6], sz[5], per[c]["shape"][:, :(4 if sz[6] == 2 else 6), :], viscous=1, loc_1d_upts=x1, SGS_model=1)
            J = np.asfortranarray(np.concatenate([S.array("Jacobian_fpts")] * tiles, axis=3))
            S.close()
            E[c].set_les(1, args.les_cs, 1.0, 0.41, 0.9, J)
    F = [hfx.IntInters(ctx, E[a], E[b], tile_table(L, plane[a], tiles), tile_table(R, plane[b], tiles)) for a, b, L, R in faces]
    for a, L, ids in bdy:
        F.append(hfx.BdyInters(ctx, E[a], tile_table(L, plane[a], tiles), np.tile(ids, tiles), hfx.bc_records(d["bc_flags"], d["bc_params"]),
                               float(np.ravel(d["bc_R_ref"])[0]), int(np.ravel(d["ramp_counter"])[0])))
    blocks = [E[c] for c in classes]
    fused = 4 if args.mode in ("auto", "general") else 0
    n_stages = int(per[classes[0]]["sizes"][7])
    hfx.run_steps_blocks(blocks, F, args.warmup, fused=fused)
    samples = []
    for _ in range(max(1, args.reps)):
        ctx.synchronize()
        t0 = time.perf_counter()
        hfx.run_steps_blocks(blocks, F, args.steps, fused=fused)
        ctx.synchronize()
        samples.append(time.perf_counter() - t0)
    elapsed = float(np.median(samples))
    for c in classes:
        assert E[c].check_nan() == -1
    dof = sum(E[c].n_eles * E[c].n_upts * E[c].n_fields for c in classes)
    value = dof * n_stages * args.steps / elapsed
    roof = None
    if fused:
        ea = (C.c_void_p * len(blocks))(*[e.h for e in blocks])
        fa = (C.c_void_p * max(1, len(F)))(*[f.h for f in F])
        ms = (C.c_double * 8)()
        by = (C.c_double * 8)()
        names = (C.c_char * 256)()
        hfx.check(lib.hfx_time_general_kernels(ea, C.c_int(len(blocks)), fa, C.c_int(len(F)), C.c_int(10), ms, names))
        hfx.check(lib.hfx_general_kernel_bytes(ea, C.c_int(len(blocks)), by))
        kn = names.value.decode().split(",")
        times = {n: ms[i] for i, n in enumerate(kn)}
        dom = max(times, key=times.get)
        i = kn.index(dom)
        achieved = by[i] / (ms[i] * 1e-3) / 1e9
        k_prof, k_src = rocprof_kernel_ms("general_" + args.workload, dom)
        roof = dict(bound="hbm", kernel=dom + " (all element blocks)", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                    traffic=None, algorithmic_bytes=by[i], kernel_ms=ms[i], kernel_ms_source="HIP events in this run",
                    kernel_ms_rocprof=k_prof, kernel_ms_rocprof_source=k_src, kernels_ms=times,
                    stage_algorithmic_bytes=sum(by[:4]), stage_hbm_frac=sum(by[:4]) / (1e-3 * sum(ms[:4])) / 1e9 / HBM_PEAK_GBS)
    cpu = None
    if not args.no_cpu:
        # the oracle on ONE tile (the reference is a serial code; the fixture was produced by the genuine reference itself)
        t0 = time.perf_counter()
        reps = 0
        if args.workload == "mixed":
            m = MU.MixedOracle(d)
            while time.perf_counter() - t0 < 10.0:
                for rk in range(n_stages):
                    assert m.CalcResidual() == -1
                    m.AdvanceSolution(rk)
                reps += 1
            dof1 = sum(m.case[c].n_eles * m.case[c].n_upts * 5 for c in m.classes)
        else:
            import oracle_py as O
            oc = O.Case(d)
            e1, (f1, nb1) = oc.c_eles(), oc.c_faces()
            O.load().orc_set_threads(1)
            while time.perf_counter() - t0 < 10.0:
                assert O.load().orc_rk_step(C.byref(e1), f1, nb1, C.byref(oc.params)) == -1
                reps += 1
            dof1 = oc.n_eles * oc.n_upts * oc.n_fields
        secs = time.perf_counter() - t0
        cpu = dict(value=dof1 * n_stages * reps / secs, unit="DOF-updates/s", cores=1, cpu_model=cpu_model(), kind="port",
                   sample="oracle (C restatement of the reference CPU path, pinned bit-exactly by this very fixture of the genuine "
                          "reference) on ONE tile of the workload (%d elements), %d time steps in %.1f s on 1 core" %
                          (sum(int(per[c]["sizes"][0]) for c in classes), reps, secs))
    desc = {"tets": "P3 tetrahedra, periodic box", "prisms": "P3 triangular prisms, periodic box",
            "mixed": "MIXED tetrahedron / prism channel with isothermal + adiabatic walls, P3 (BASELINE.json configs[3])"}[args.workload]
    line = {"metric": "DOF-updates/sec", "value": value, "unit": "DOF-updates/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_rk_stage": 1e3 * elapsed / args.steps / n_stages, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "timing": {"reps": len(samples), "statistic": "median", "seconds_per_rep": samples, "stages_per_rep": n_stages * args.steps},
            "config": {"workload": "%s: the reference's fixture %s tiled %d times, Navier-Stokes, HLLC+LDG, RK45%s" %
                                   (desc, name, tiles, ", LES WALE C_s %g" % args.les_cs if args.les_cs >= 0 else ""),
                       "n_eles": {str(c): E[c].n_eles for c in classes}, "dof": dof,
                       "path": "general fused stage (fused 4)" if fused else "per method", "multi_gpu": "none"},
            "roofline": roof, "cpu_baseline": cpu}
    print(json.dumps(line))
    for f in F:
        f.close()
    for c in classes:
        E[c].close()
    ctx.close()
    return 0


ALSO_LEGS = [
    # (name, what it is, extra arguments)
    ("mixed_p3_channel", "BASELINE.json configs[3]: mixed tetrahedron / prism channel, P3, general fused stage", ["--workload", "mixed"]),
    ("config5_overint_shock", "BASELINE.json configs[4]'s ingredients on one GPU: 32^3 P4 hexes, over-integration (7 points per direction) + "
     "shock capturing after every stage", ["--over-int-order", "6", "--shock-s0", "1e-3"]),
    ("les_wale", "32^3 P4 hexes, LES with the WALE closure (SURVEY 8f rank 4)", ["--les-cs", "0.325"]),
    ("mixed_les_wale", "the mixed tetrahedron / prism channel with the WALE closure, evaluated inside the general fused stage's flux kernel",
     ["--workload", "mixed", "--les-cs", "0.325"]),
    ("self_partition", "the 2x2x2 rank's share: 32^3 P4 with all 6 144 wrap-around faces as partition faces, exchanged over RCCL with the "
     "rank itself", ["--self-partition"]),
]


def also_legs(args):
    """Short legs of the other configurations as CHILD processes of the default run, after its timed region: each is this
    script with --steps 4 --reps 3 --no-cpu --no-also and prints its own line, of which the summary goes into `also`."""
    out = {}
    for name, what, extra in ALSO_LEGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "1", "--reps", "3", "--no-cpu", "--no-also", "--no-api-path"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=float(os.environ.get("HFX_BENCH_ALSO_TIMEOUT", "90")))
        except subprocess.TimeoutExpired:
            out[name] = {"error": "timed out", "what": what}
            continue
        lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
        if r.returncode != 0 or not lines:
            out[name] = {"error": "exit code %d: %s" % (r.returncode, r.stderr[-300:]), "what": what}
            continue
        d = json.loads(lines[-1])
        roof = d.get("roofline") or {}
        out[name] = {"what": what, "ms_per_rk_stage": d["ms_per_rk_stage"], "dof_updates_per_s": d["value"],
                     "dominant_kernel": roof.get("kernel"), "frac": roof.get("frac"), "dominant_kernel_ms": roof.get("kernel_ms"),
                     "path": d["config"].get("path"), "steps": d["steps"], "reps": d["timing"]["reps"],
                     "leg_wall_s": round(time.perf_counter() - t0, 1)}
        if d.get("partitioned_stage_ms"):
            out[name]["partitioned_stage_ms"] = d["partitioned_stage_ms"]
    return out


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of THIS process, which has not
    touched (and never touches) a GPU, and relay rank 0's JSON line.  Transports are tried in order: libhfx's own RCCL
    communicator, RCCL through torch.distributed, gloo with host staging."""
    order = [os.environ["HFX_BENCH_TRANSPORT"]] if os.environ.get("HFX_BENCH_TRANSPORT") else ["rccl", "torch", "gloo"]
    last = ""
    # time limits: the whole chain must fit the driver's limit for one bench run (600 s) with room for the line to be
    # relayed -- 540 s in all; the first transport (which also pays the first `import torch` of N processes on a fresh box)
    # gets up to 300 s, every later one at most 150 s of what is left.  HFX_BENCH_CHILD_TIMEOUT overrides the per-leg limit.
    t_start = time.perf_counter()
    total = float(os.environ.get("HFX_BENCH_TOTAL_TIMEOUT", "540"))
    for k, transport in enumerate(order):
        left = total - (time.perf_counter() - t_start)
        limit = float(os.environ["HFX_BENCH_CHILD_TIMEOUT"]) if os.environ.get("HFX_BENCH_CHILD_TIMEOUT") else (300.0 if k == 0 else 150.0)
        limit = min(limit, left)
        if limit < 20.0:
            last += "transport %s: not tried, %.0f s left of %.0f s\n" % (transport, left, total)
            sys.stderr.write(last)
            break
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
        env = dict(os.environ, HFX_BENCH_TRANSPORT=transport)
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit)
        except subprocess.TimeoutExpired as e:
            err = e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
            last = "transport %s: TIMED OUT after %.0f s (leg %d of %d)\n%s" % (transport, limit, k + 1, len(order), err[-2000:])
            sys.stderr.write(last + "\n")
            continue
        lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
        if r.returncode == 0 and lines:
            sys.stderr.write(r.stderr[-4000:])
            print(lines[-1])
            return 0
        last = "transport %s: exit code %d\n%s" % (transport, r.returncode, r.stderr[-4000:])
        sys.stderr.write(last + "\n")
    raise SystemExit("bench.py --gpus %d: every transport failed\n%s" % (args.gpus, last))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed region; the median is reported")
    ap.add_argument("--cells", "--n", dest="n", type=int, default=32, help="cells per direction per GPU")
    ap.add_argument("--order", type=int, default=4)
    ap.add_argument("--mode", default="auto", choices=["auto", "split", "split3", "methods", "dense", "general"])
    ap.add_argument("--workload", default="tgv", choices=["tgv", "tets", "prisms", "mixed"],
                    help="tgv: BASELINE.json configs[1] (default); tets / prisms / mixed: configs[3]'s element classes, see general_workload")
    ap.add_argument("--tiles", type=int, default=2048, help="copies of the fixture mesh (workloads tets / prisms / mixed)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    # BASELINE.json configs[4]'s ingredients on the same mesh (not the default workload): polynomial de-aliasing of the
    # inviscid flux and shock capturing after every stage
    ap.add_argument("--over-int-order", type=int, default=-1, help="over_int 1 with this over_int_order (cubature points per direction - 1)")
    ap.add_argument("--shock-s0", type=float, default=-1.0, help="shock_cap 1 with this sensor threshold s0")
    ap.add_argument("--les-cs", type=float, default=-1.0, help="LES 1 with the WALE closure and this C_s (split3: the closure in the flux "
                    "kernel; --mode split: the variant that keeps the corrected gradients in HBM for a pointwise closure kernel)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="hfx_ctx_set_option knob for A/B runs (split_grid_per_cu, xcd_order, dictionary_rows, flux_waves, "
                         "buffer_addressing, loader_wave, flux_stamps, tensor_ops); echoed into config.options")
    ap.add_argument("--no-api-path", action="store_true", help="skip the timing of the mirrored CalcResidual + AdvanceSolution loop")
    ap.add_argument("--no-also", action="store_true", help="skip the short legs of the other workloads (the `also` block of the default run)")
    ap.add_argument("--self-partition", action="store_true", help="N=1 only: the box's wrap-around faces become partition faces "
                    "to the rank itself, i.e. the partitioned stage with its RCCL exchanges priced on one GPU")
    args = ap.parse_args()

    if args.workload != "tgv":
        if args.gpus != 1:
            raise SystemExit("bench.py --workload %s runs on one GPU" % args.workload)
        return general_workload(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (start N ranks with torch.distributed.run, or run "
                         "`python bench.py --gpus N` without a launcher)" % (args.gpus, world))

    import torch
    dist = None
    transport = "none"
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rccl : libhfx's own communicator (grouped ncclSend / ncclRecv on its communication stream); torch.distributed
        #        over gloo is the control plane only (unique id, barriers, the MAX of the elapsed time)
        # torch: the exchange hook through torch.distributed's RCCL backend
        # gloo : host-staged exchange, ranks may share cards (rehearsal on a box with fewer GPUs than ranks)
        transport = os.environ.get("HFX_BENCH_TRANSPORT", "rccl")
        if transport == "gloo":
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if transport == "torch":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    torch.cuda.set_device(local_rank)

    import hfx
    import hfx_host as H

    pgrid = {1: None, 2: [2, 1, 1], 4: [2, 2, 1], 8: [2, 2, 2]}.get(world, [world, 1, 1])
    extra = {}
    if args.over_int_order >= 0:
        extra.update(over_int=1, over_int_order=args.over_int_order)
    if args.shock_s0 >= 0:
        extra.update(shock_cap=1, s0=args.shock_s0, expf_fac=36.0, expf_order=4, expf_cutoff=1, shock_det_field=0)
    if args.les_cs >= 0:
        extra.update(LES=1, SGS_model=1, C_s=args.les_cs, filter_ratio=1.0)
    # the reference's own solution-point abscissae (the data/JacobiGQ.bin row, carried as data by the full-size fixture):
    # with them the LDG switch on this axis-aligned mesh falls as in the reference (tests/test_fullsize_vs_reference.py)
    nodes = None
    fx = os.path.join(ROOT, "tests", "golden", "hex_p4_n32_tgv.npz")
    if args.order == 4 and os.path.exists(fx):
        nodes = np.load(fx)["loc_1d_upts"]
    selfp = [1, 1, 1] if (args.self_partition and world == 1) else None
    case = H.Case(args.n, order=args.order, rank=rank, pgrid=pgrid, loc_1d_upts=nodes, self_partition=selfp, **extra)
    case.to_device(local_rank)
    ctx, e, faces, nb = case.handles()
    lib = hfx.lib()
    options = {}
    for kv in args.opt:
        k, v = kv.split("=")
        hfx.check(lib.hfx_ctx_set_option(ctx, k.encode(), C.c_int(int(v))))
        options[k] = int(v)
    ex = None
    partitioned = world > 1 or selfp is not None
    if partitioned:
        if args.mode not in ("auto", "split", "split3", "methods"):
            raise SystemExit("a partitioned run takes the split fused path or the per-method path")
        if selfp is not None:
            transport = "rccl"
        pf = args.mode in ("auto", "split3")
        dev = torch.device("cuda", local_rank)

        def agreed(ok):
            """True when the step succeeded on EVERY rank (control plane)"""
            if dist is None:
                return bool(ok)
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        def trial():
            try:
                if args.mode == "methods":
                    case.CalcResidual()
                else:
                    case.run_partitioned(1)
                torch.cuda.synchronize()
                return True
            except Exception as err:  # noqa: BLE001
                sys.stderr.write("rank %d: partition-face exchange over %s failed: %s\n" % (rank, transport, err))
                return False

        if transport == "rccl":
            ok = True
            try:
                uid = [hfx.comm_unique_id() if rank == 0 else None]
            except Exception as err:  # noqa: BLE001
                sys.stderr.write("rank %d: %s\n" % (rank, err))
                uid, ok = [None], False
            if dist is not None:
                dist.broadcast_object_list(uid, src=0)
            ok = ok and uid[0] is not None
            if agreed(ok):
                try:
                    case.set_comm(uid[0])  # collective: ncclCommInitRank
                except Exception as err:  # noqa: BLE001
                    sys.stderr.write("rank %d: %s\n" % (rank, err))
                    ok = False
            if not (agreed(ok) and agreed(trial())):
                if selfp is not None:
                    raise SystemExit("--self-partition needs libhfx's RCCL transport")
                case.set_comm(None)
                transport = "torch"
        if transport != "rccl":
            import exchange
            group = None
            if transport == "torch" and dist.get_backend() != "nccl":
                # the launcher's control plane is gloo: a second group on RCCL for the data path
                try:
                    group = dist.new_group(backend="nccl", device_id=dev)
                except TypeError:
                    group = dist.new_group(backend="nccl")
            ex = exchange.for_case(case, group=group, device=dev, projected_flux=pf)
            case.set_exchange(ex)
            case.set_reduce_min(lambda v: float(_min_over_ranks(dist, torch, v)))
            if transport == "torch" and not agreed(trial()):
                ex.close()
                transport = "gloo"
                ex = exchange.for_case(case, device=dev, projected_flux=pf)  # host staged through the gloo control group
                case.set_exchange(ex)

    mode = args.mode
    if partitioned:
        mode = "split3" if mode == "auto" else mode
    elif mode in ("auto", "split", "split3"):
        probe = {"split": 2, "split3": 3}[DEFAULT_FUSED if mode == "auto" else mode]
        rc = lib.hfx_run_steps(e, faces, C.c_int(nb), C.c_int(0), C.c_int(probe))
        if mode != "auto" and rc != 0:
            raise SystemExit("fused path unavailable: " + lib.hfx_last_error().decode())
        if mode == "auto":
            mode = DEFAULT_FUSED if rc == 0 else "methods"
    if mode == "dense":
        hfx.check(lib.hfx_ctx_set_contract_mode(ctx, C.c_int(hfx.CONTRACT_DENSE)))
    fused = {"split": 2, "split3": 3}.get(mode, 0)
    if fused:
        hfx.check(lib.hfx_ctx_set_fused_mode(ctx, C.c_int(fused)))

    def barrier():
        if dist is not None:
            dist.barrier()

    def run(nsteps):
        if not partitioned:
            hfx.check(lib.hfx_run_steps(e, faces, C.c_int(nb), C.c_int(nsteps), C.c_int(fused)))
        elif fused:
            case.run_partitioned(nsteps)  # hfx_run_steps_partitioned (RCCL inside libhfx) or phases + exchange hook
        elif nsteps:
            case.run(nsteps)              # mirrored CalcResidual with the mpi_inters calls

    run(args.warmup)
    samples = []
    for _ in range(max(1, args.reps)):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        samples.append(el)
    elapsed = float(np.median(samples))

    # state sanity: the run must not have produced NaNs (src/eles.cpp:1781-1795)
    bad = C.c_long(0)
    hfx.check(lib.hfx_eles_check_nan(e, C.byref(bad)))
    assert bad.value == -1, "NaN in the residual at flat index %d" % bad.value

    n_stages = case.n_stages
    dof_per_rank = case.n_eles * case.n_upts * case.n_fields
    total_updates = dof_per_rank * world * n_stages * args.steps
    value = total_updates / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    ms_per_stage = ms_per_step / n_stages

    phases = None
    rccl_info = None
    part_roof = None
    if partitioned and fused and transport == "rccl":
        # per-phase and per-exchange times of the partitioned stage (HIP events on the compute / communication streams) in
        # its SERIALISED schedule (every kernel on the compute stream so that the events bracket the phases; the timed run
        # above puts the one-sided partition-face kernels on the communication stream); collective: every rank runs it,
        # rank 0 reports its own
        phases = case.time_partitioned(10)
        phases["schedule"] = "serialised (all kernels on the compute stream); `value` is timed on the overlapped schedule"
        # what RCCL itself reports for every rank's communicator: N ranks on N different devices
        mine = case.comm_info()
        if dist is not None:
            allinfo = [None] * world
            dist.all_gather_object(allinfo, mine)
        else:
            allinfo = [mine]
        rccl_info = {"rccl_nranks": mine["nranks"], "ranks": allinfo,
                     "distinct_devices": len({i["pci_bus_id"] for i in allinfo})}
        if rank == 0 and phases.get("flux_kernel", 0.0) > 0.0:
            share = (C.c_double * 8)()
            hfx.check(lib.hfx_fused_kernel_bytes(e, share))
            ach = share[1] / (phases["flux_kernel"] * 1e-3) / 1e9
            part_roof = dict(bound="hbm", kernel="split_flux_tensor_kernel (rank 0, partitioned stage)", achieved=ach, peak=HBM_PEAK_GBS,
                             unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None, algorithmic_bytes=share[1], kernel_ms=phases["flux_kernel"],
                             kernel_ms_source="HIP events around the kernel inside hfx_time_partitioned, this run")

    # ---- the reference's UNCHANGED call sequence: the mirrored CalcResidual + AdvanceSolution loop (csrc/host/solver.cpp makes
    # the seventeen method calls of src/solver.cpp:59-221 + src/HiFiLES.cpp:201-217 one by one through the C ABI), timed with
    # libhfx deferring them (whole stages run as the fused stage) and with every call launching its own kernels
    api_path = None
    if not partitioned and fused and not args.no_api_path:
        api_path = {}
        for label, on, nsteps in (("deferred", True, args.steps), ("per_method", False, max(2, args.steps // 4))):
            case.set_deferred(on)
            case.run(1)
            case.synchronize()
            nf0 = hfx.deferred_stats(ctx)[:2]
            t0 = time.perf_counter()
            case.run(nsteps)
            case.synchronize()
            el = time.perf_counter() - t0
            nf1 = hfx.deferred_stats(ctx)[:2]
            api_path[label] = {"ms_per_rk_stage": 1e3 * el / nsteps / case.n_stages, "steps": nsteps,
                               "stages_run_fused": nf1[0] - nf0[0], "records_replayed_call_by_call": nf1[1] - nf0[1]}
        case.set_deferred(True)

    roof = None
    cpu = None
    if rank == 0 and not partitioned:
        # ---- roofline of the dominant kernel, timed live with HIP events on the library's stream
        if fused:
            kt = (C.c_double * 8)()
            names = (C.c_char * 256)()
            hfx.check(lib.hfx_time_fused_kernels(e, faces, C.c_int(nb), C.c_int(10), kt, names))
            kn = names.value.decode().split(",")
            times = {n: kt[i] for i, n in enumerate(kn) if n}
            dom = max(times, key=times.get)
            per_ele = stage_algorithmic_bytes(case)
            # the stage's compulsory bytes are shared by the fused kernels in proportion to what each touches
            share = (C.c_double * 8)()
            hfx.check(lib.hfx_fused_kernel_bytes(e, share))
            bytes_launch = share[kn.index(dom)]
            achieved = bytes_launch / (times[dom] * 1e-3) / 1e9
            # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, collected in separate
            # rocprofv3 runs by tools/profile_round.sh and committed under profiles/); only valid for the
            # workload it was measured on
            traffic, traffic_source = None, None
            for rnd in ("r03", "r02", "r01"):
                tfile = os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (rnd, mode))
                if os.path.exists(tfile) and args.n == 32 and args.order == 4 and not extra:
                    traffic = json.load(open(tfile)).get(dom, {}).get("traffic_bytes_corrected")
                    traffic_source = "profiles/%s_%s_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this " \
                                     "command on another box, builder run; not measured in this run)" % (rnd, mode)
                    break
            k_prof, k_src = rocprof_kernel_ms(mode, dom) if (args.n == 32 and args.order == 4 and not extra) else (None, None)
            roof = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_source,
                        algorithmic_bytes=bytes_launch, kernel_ms=times[dom], kernel_ms_source="HIP events in this run",
                        kernel_ms_rocprof=k_prof, kernel_ms_rocprof_source=k_src,
                        kernels_ms=times, stage_algorithmic_bytes_per_element=per_ele,
                        stage_hbm_frac=per_ele * case.n_eles / (ms_per_stage * 1e-3) / 1e9 / HBM_PEAK_GBS)
        else:
            ms = (C.c_double * 11)()
            hfx.check(lib.hfx_time_methods(e, faces, C.c_int(nb), C.c_int(10), ms))
            mb = method_bytes(case)
            times = {n: ms[i] for i, n in enumerate(METHOD_NAMES)}
            if mode == "dense":
                # dense FP64 MFMA contractions: price the heaviest contraction method against the MFMA roof
                nu, nfp, nf, nd = case.n_upts, case.n_fpts, case.n_fields, case.n_dims
                flops = {"extrapolate_solution": 2.0 * nfp * nu * nf, "calculate_gradient": 2.0 * nu * nu * nf * nd,
                         "correct_gradient": 2.0 * (nu * nfp + nfp * nu) * nf * nd,
                         "extrapolate_totalFlux": 2.0 * nfp * nu * nf * nd, "calculate_divergence": 2.0 * nu * nu * nf * nd,
                         "calculate_corrected_divergence": 2.0 * nu * nfp * nf}
                dom = max(flops, key=lambda k: times[k])
                achieved = flops[dom] * case.n_eles / (times[dom] * 1e-3) / 1e12
                roof = dict(bound="mfma", kernel=dom, achieved=achieved, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=achieved / FP64_PEAK_TFLOPS, traffic=None, kernel_ms=times[dom], methods_ms=times)
            else:
                dom = max(times, key=times.get)
                achieved = mb[dom] * case.n_eles / (times[dom] * 1e-3) / 1e9
                roof = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=achieved / HBM_PEAK_GBS, traffic=None, kernel_ms=times[dom], methods_ms=times)
        if not args.no_cpu:
            # the GPU box gives one job a 16-CPU share whatever os.cpu_count() says
            threads = max(1, min(16, len(os.sched_getaffinity(0)), os.cpu_count() or 1))
            cb = cpu_baseline(args.order, threads)
            port_note = ("oracle (C restatement of the reference CPU path, OpenMP over elements) on a %d^3 P%d TGV mesh, "
                         "%d time steps: %.3g DOF-updates/s on %d cores; on 1 core (%d^3): %.3g" %
                         (cb["allcores"]["n_cells"], args.order, cb["allcores"]["steps"], cb["allcores"]["value"], threads,
                          cb["1core"]["n_cells"], cb["1core"]["value"]))
            rb = reference_baseline(args.order)
            if rb is not None:
                # the reference is a serial code (no OpenMP / threads anywhere in its src/): one core is all it uses
                cpu = dict(value=rb["value"], unit="DOF-updates/s", cores=1, cpu_model=cpu_model(), kind="reference",
                           sample="genuine reference (oracle/_ref/ref_harness: g++ -O3 of the reference's own sources, serial, "
                                  "BLAS=NO), its RK loop on a %d^3 P%d TGV mesh, %d time steps in %.1f s; beside it the %s" %
                                  (rb["n_cells"], args.order, rb["steps"], rb["seconds"], port_note),
                           port_value_allcores=cb["allcores"]["value"], port_cores=threads, port_value_1core=cb["1core"]["value"])
            else:
                cpu = dict(value=cb["allcores"]["value"], unit="DOF-updates/s", cores=threads, cpu_model=cpu_model(), kind="port",
                           sample=port_note, value_1core=cb["1core"]["value"])

    also = None
    if (rank == 0 and world == 1 and not args.no_also and not partitioned and not extra and args.n == 32 and args.order == 4
            and args.mode == "auto" and not args.opt):
        also = also_legs(args)

    if rank == 0:
        knobs = {k: v for k, v in os.environ.items() if k.startswith("HFX_") and k != "HFX_BENCH_CHILD_TIMEOUT"}
        tname = {"rccl": "RCCL p2p (grouped ncclSend / ncclRecv on libhfx's communication stream, hfx_run_steps_partitioned)",
                 "torch": "RCCL p2p through torch.distributed (exchange hook)",
                 "gloo": "gloo p2p, host staged (rehearsal)"}.get(transport, transport)
        if world > 1:
            mg = "one periodic box split into %s blocks, partition-face exchange over %s" % ("x".join(map(str, pgrid)), tname)
        elif selfp is not None:
            mg = "one rank that is its own neighbour in x, y and z (self-partition): the partitioned stage on one GPU, exchange over " + tname
        else:
            mg = "none"
        line = {
            "metric": "DOF-updates/sec", "value": value, "unit": "DOF-updates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_rk_stage": ms_per_stage, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "timing": {"reps": len(samples), "statistic": "median", "seconds_per_rep": samples, "stages_per_rep": n_stages * args.steps},
            "config": {"workload": "Taylor-Green vortex, %d^3 hexa per GPU, P%d, Navier-Stokes, HLLC+LDG, RK45, "
                                   "1 step = %d RK stages" % (args.n, args.order, n_stages) +
                                   (", over-integration order %d" % args.over_int_order if args.over_int_order >= 0 else "") +
                                   (", shock capturing s0 %g" % args.shock_s0 if args.shock_s0 >= 0 else "") +
                                   (", LES WALE C_s %g" % args.les_cs if args.les_cs >= 0 else ""),
                       "n_eles_per_gpu": case.n_eles, "dof_per_gpu": dof_per_rank, "path": mode,
                       "solution_points": "reference's data/JacobiGQ.bin row" if nodes is not None else "computed Gauss nodes",
                       "multi_gpu": mg, "options": options, "env_knobs": knobs},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if also is not None:
            line["also"] = also
        if api_path is not None:
            line["api_path_ms_per_rk_stage"] = {k: v["ms_per_rk_stage"] for k, v in api_path.items()}
            line["api_path"] = dict(api_path, note="the reference's unchanged call sequence (mirrored CalcResidual + AdvanceSolution, one C-ABI "
                                    "call per reference method): 'deferred' = libhfx records the calls and runs each whole stage as the fused "
                                    "stage (the host mirror's default), 'per_method' = every call launches its own kernels; `value` is "
                                    "hfx_run_steps, the same fused stages without the per-call recording")
        if phases is not None:
            line["partitioned_stage_ms"] = phases
        if rccl_info is not None:
            line["rccl"] = rccl_info
            line["rccl_nranks"] = rccl_info["rccl_nranks"]
        if part_roof is not None and roof is None:
            line["roofline"] = part_roof
        print(json.dumps(line))
    if ex is not None:
        ex.close()
    case.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def _json_only_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner on
    stdout when its first communicator comes up), so everything that is not the line goes to stderr: file descriptor 1 is
    pointed at stderr for the whole run and `print` writes to the saved descriptor."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(saved, "w", buffering=1)


if __name__ == "__main__":
    _json_only_stdout()
    sys.exit(main())

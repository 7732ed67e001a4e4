"""Workers of the partitioned (N-rank) tests: TEST INFRASTRUCTURE.

A periodic box is split into blocks on a process grid; rank r owns block r.  The invariant that pins
the partition-face path (oracle's orc_mpi_* on the CPU, libhfx's mpi_inters kernels on the GPU) is
PARTITION INVARIANCE: the N-rank result equals the 1-rank result of the same global box, where the
1-rank oracle is itself pinned bit-exactly against the genuine reference (test_oracle_vs_golden.py).
"""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hifiles-solver_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def pcoord(rank, pgrid):
    out = []
    for g in pgrid:
        out.append(rank % g)
        rank //= g
    return out


def global_index(n_local, pgrid, rank):
    """global element number of every local element (x fastest, like the box mesh)."""
    pc = pcoord(rank, pgrid)
    dims = len(n_local)
    G = [n_local[d] * pgrid[d] for d in range(dims)]
    idx = []
    rng = [range(n_local[d]) for d in range(dims)]
    if dims == 2:
        for j in rng[1]:
            for i in rng[0]:
                idx.append((i + pc[0] * n_local[0]) + G[0] * (j + pc[1] * n_local[1]))
    else:
        for k in rng[2]:
            for j in rng[1]:
                for i in rng[0]:
                    idx.append((i + pc[0] * n_local[0]) + G[0] * ((j + pc[1] * n_local[1]) + G[1] * (k + pc[2] * n_local[2])))
    return np.array(idx)


def init_pg(rank, world, port, backend="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def oracle_worker(rank, world, port, n_local, pgrid, cfg, n_steps, outdir):
    """N-rank oracle run: host-mirror partition tables + orc_mpi_* + gloo exchange on CPU tensors."""
    import torch
    import hfx_host as H
    import oracle_py as O
    from exchange import Exchange
    dist = init_pg(rank, world, port)
    try:
        O.load().orc_set_threads(1)
        c = H.Case(list(n_local), rank=rank, pgrid=list(pgrid), **cfg)
        reg = c.registration()
        L, Rlut, nout = c.mpi_faces()
        pc = O.PartitionedCase(reg, L, Rlut)
        t = {k: torch.from_numpy(v) for k, v in pc.buf.items()}
        bufs = {0: (t["out_disu"], t["in_disu"])}
        if pc.viscous:
            bufs[1] = (t["out_grad"], t["in_grad"])
        pc.exchange = Exchange(nout, rank, bufs)
        for _ in range(n_steps):
            pc.rk_step()
        np.save(os.path.join(outdir, "u_rank%d.npy" % rank), pc.arr["u0"])
        np.save(os.path.join(outdir, "div_rank%d.npy" % rank), pc.arr["div_tconf_upts"])
        c.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def single_rank_oracle(n_global, cfg, n_steps):
    import ctypes as C
    import hfx_host as H
    import oracle_py as O
    o = O.load()
    c = H.Case(list(n_global), **cfg)
    oc = O.Case(c.registration())
    e, (f, nb) = oc.c_eles(), oc.c_faces()
    for _ in range(n_steps):
        bad = o.orc_rk_step(C.byref(e), f, nb, C.byref(oc.params))
        assert bad < 0
    c.close()
    return oc.arr["u0"], oc.arr["div_tconf_upts"]


def assemble(outdir, name, n_local, pgrid, shape_global):
    world = int(np.prod(pgrid))
    out = np.zeros(shape_global, order="F")
    for r in range(world):
        a = np.load(os.path.join(outdir, "%s_rank%d.npy" % (name, r)))
        out[:, global_index(n_local, pgrid, r), :] = a
    return out


def gpu_worker(rank, world, port, n_local, pgrid, cfg, n_steps, outdir, mode, backend="gloo"):
    """N-rank libhfx run (all ranks on cuda:0 when the box has one GPU): mode "methods" = the mirrored
    CalcResidual with mpi_inters calls, mode "fused" = hfx_stage_partitioned phases."""
    import faulthandler
    faulthandler.enable()
    import torch
    import hfx_host as H
    import exchange
    ndev = torch.cuda.device_count()
    dev = rank % ndev
    torch.cuda.set_device(dev)
    dist = init_pg(rank, world, port, backend)
    try:
        c = H.Case(list(n_local), rank=rank, pgrid=list(pgrid), **cfg)
        c.to_device(dev)
        if mode == "fused2":
            import ctypes as C
            import hfx
            hfx.check(hfx.lib().hfx_ctx_set_fused_mode(c.handles()[0], C.c_int(2)))
        ex = exchange.for_case(c, device=torch.device("cuda", dev), projected_flux=(mode == "fused"))
        if ex is not None:
            c.set_exchange(ex)
        if mode == "methods":
            c.run(n_steps)
        else:
            c.run_partitioned(n_steps)
        c.sync_host()
        np.save(os.path.join(outdir, "u_rank%d.npy" % rank), c.array("disu_upts0"))
        np.save(os.path.join(outdir, "div_rank%d.npy" % rank), c.array("div_tconf_upts"))
        dist.barrier()
        if ex is not None:
            ex.close()
        c.close()
    finally:
        dist.destroy_process_group()


def spawn(fn, world, args):
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(fn, args=(world, port) + tuple(args), nprocs=world, join=True)

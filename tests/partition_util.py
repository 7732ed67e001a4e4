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


def case_kw(cfg):
    """cfg may carry `self_partition` (a Case keyword, not an input key)"""
    cfg = dict(cfg)
    sp = cfg.pop("self_partition", None)
    return cfg, sp


def oracle_worker(rank, world, port, n_local, pgrid, cfg, n_steps, outdir):
    """N-rank oracle run: host-mirror partition tables + orc_mpi_* + gloo exchange on CPU tensors."""
    import torch
    import hfx_host as H
    import oracle_py as O
    from exchange import Exchange
    dist = init_pg(rank, world, port)
    try:
        O.load().orc_set_threads(1)
        cfg, sp = case_kw(cfg)
        c = H.Case(list(n_local), rank=rank, pgrid=list(pgrid), self_partition=sp, **cfg)
        reg = c.registration()
        L, Rlut, nout = c.mpi_faces()
        pc = O.PartitionedCase(reg, L, Rlut)
        t = {k: torch.from_numpy(v) for k, v in pc.buf.items()}
        bufs = {0: (t["out_disu"], t["in_disu"])}
        if pc.viscous:
            bufs[1] = (t["out_grad"], t["in_grad"])
        if pc.les:
            bufs[2] = (t["out_sgsf"], t["in_sgsf"])
        pc.exchange = Exchange(nout, rank, bufs, seg=c.mpi_segments())
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
    cfg, _ = case_kw(cfg)  # the undivided box has interior faces where a self-partitioned rank has partition faces
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


def gpu_worker(rank, world, port, n_local, pgrid, cfg, n_steps, outdir, mode, backend="gloo", transport="torch"):
    """N-rank libhfx run (all ranks on cuda:0 when the box has one GPU): mode "methods" = the mirrored
    CalcResidual with mpi_inters calls, mode "fused" = hfx_stage_partitioned phases.
    transport "torch": the exchange hook through torch.distributed; "rccl": libhfx's own communicator (hfx_comm_*),
    the unique id broadcast over the process group."""
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
        import ctypes as C
        import hfx
        cfg, sp = case_kw(cfg)
        c = H.Case(list(n_local), rank=rank, pgrid=list(pgrid), self_partition=sp, **cfg)
        c.to_device(dev)
        if mode == "fused2":
            hfx.check(hfx.lib().hfx_ctx_set_fused_mode(c.handles()[0], C.c_int(2)))
        ex = None
        if transport == "rccl":
            uid = [hfx.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            c.set_comm(uid[0])
        else:
            ex = exchange.for_case(c, device=torch.device("cuda", dev), projected_flux=(mode == "fused"))
            if ex is not None:
                c.set_exchange(ex)
                c.set_reduce_min(lambda v: _allreduce_min(dist, v))
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


def _allreduce_min(dist, v):
    import torch
    t = torch.tensor([v], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


class ThreadTransport:
    """N ranks as N THREADS of one process (a GPU box allows few processes on its card, 8 ranks do not fit otherwise):
    the exchange hook of rank r waits at a barrier until every rank has packed, then PULLS its segments out of the peers'
    out buffers (device-to-device copies), and a second barrier keeps anyone from repacking before all have pulled.
    Also the MIN reduction of calc_time_step."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.out = [dict() for _ in range(world)]  # rank -> {kind: tensor}
        self.inn = [dict() for _ in range(world)]
        self.seg = [None] * world
        self.sync = [None] * world
        self.vals = [0.0] * world

    def register(self, rank, case, projected_flux):
        import torch
        import hfx
        import exchange
        h = case.mpi_handle()
        dev = torch.device("cuda", torch.cuda.current_device())
        les = bool(case.cfg.get("LES", 0))
        t = [exchange.device_tensor(*hfx.mpi_buffer(h, w), dev) for w in ((0, 1, 4, 5) if (projected_flux and not les) else (0, 1, 2, 3))]
        self.out[rank] = {0: t[0], 1: t[2]}
        self.inn[rank] = {0: t[1], 1: t[3]}
        if les:
            self.out[rank][2], self.inn[rank][2] = (exchange.device_tensor(*hfx.mpi_buffer(h, w), dev) for w in (6, 7))
        self.seg[rank] = case.mpi_segments()
        self.sync[rank] = case.synchronize

    def hook(self, rank):
        import torch

        def fn(kind, phase):
            if phase == 0:
                return
            self.sync[rank]()       # this rank's pack kernel has finished
            self.barrier.wait()     # ... and everybody else's
            n_faces = sum(c for _, _, _, c in self.seg[rank])
            rec = self.out[rank][kind].numel() // max(1, n_faces)
            for p, s, r, c in self.seg[rank]:
                # what lands in this segment's receive slot: the peer's segment for this rank -- or, for the faces a rank
                # shares with itself, this very segment's send slot (a group's records go to the mate group's slots)
                ps = s if p == rank else [q for q in self.seg[p] if q[0] == rank][0][1]
                self.inn[rank][kind][r * rec:(r + c) * rec].copy_(self.out[p][kind][ps * rec:(ps + c) * rec])
            torch.cuda.synchronize()
            self.barrier.wait()
        return fn

    def reduce_min(self, rank):
        def fn(v):
            self.vals[rank] = v
            self.barrier.wait()
            m = min(self.vals)
            self.barrier.wait()
            return m
        return fn


def threaded_gpu_run(world, n_local, pgrid, cfg, n_steps, mode):
    """All ranks of a partitioned box as threads of THIS process on cuda:0; returns [(u, div)] per rank."""
    import ctypes as C
    import threading
    import torch
    import hfx
    import hfx_host as H
    torch.cuda.set_device(0)
    T = ThreadTransport(world)
    out = [None] * world
    err = []
    cfg, sp = case_kw(cfg)

    def work(rank):
        try:
            torch.cuda.set_device(0)
            c = H.Case(list(n_local), rank=rank, pgrid=list(pgrid), self_partition=sp, **cfg)
            c.to_device(0)
            if mode == "fused2":
                hfx.check(hfx.lib().hfx_ctx_set_fused_mode(c.handles()[0], C.c_int(2)))
            T.register(rank, c, projected_flux=(mode == "fused"))
            c.set_exchange(T.hook(rank))
            c.set_reduce_min(T.reduce_min(rank))
            T.barrier.wait()
            if mode == "methods":
                c.run(n_steps)
            else:
                c.run_partitioned(n_steps)
            c.sync_host()
            out[rank] = (c.array("disu_upts0"), c.array("div_tconf_upts"))
            T.barrier.wait()
            c.close()
        except BaseException as e:  # noqa: BLE001
            err.append(e)
            T.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if err:
        raise err[0]
    return out


def assemble_arrays(parts, which, n_local, pgrid, shape_global):
    out = np.zeros(shape_global, order="F")
    for r, p in enumerate(parts):
        out[:, global_index(n_local, pgrid, r), :] = p[which]
    return out


def spawn(fn, world, args):
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(fn, args=(world, port) + tuple(args), nprocs=world, join=True)

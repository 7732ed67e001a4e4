"""N>1 path on the GPU: 2 and 4 ranks (all on the box's one card, exchange over gloo with host staging)
drive libhfx's partition-face kernels -- through the mirrored CalcResidual with the reference's
send/receive call order, and through hfx_stage_partitioned (split fused kernels) -- and must reproduce
the single-rank oracle of the global box.  "fused" = fused mode 3 (projected viscous flux on the wire),
"fused2" = fused mode 2 (corrected gradient on the wire, as the reference)."""
import os

import numpy as np
import pytest

import partition_util as PU

pytestmark = pytest.mark.gpu
CFG = dict(order=2, amp=0.05, length=6.2831853071795862, T_c_ic=300.0, dt=1e-4)


def rel(a, b):
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


@pytest.mark.parametrize("mode", ["methods", "fused", "fused2"])
@pytest.mark.parametrize("n_local,pgrid,kw", [
    ([2, 4, 4], [2, 1, 1], dict(riemann_solve_type=3)),
    ([4, 4, 2], [1, 1, 2], dict(riemann_solve_type=0, order=3)),
    ([2, 2, 4], [2, 2, 1], dict(riemann_solve_type=3)),
])
def test_gpu_partition_invariance(tmp_path, mode, n_local, pgrid, kw):
    cfg = dict(CFG)
    cfg.update(kw)
    world = int(np.prod(pgrid))
    PU.spawn(PU.gpu_worker, world, (n_local, pgrid, cfg, 2, str(tmp_path), mode))
    n_global = [n_local[d] * pgrid[d] for d in range(3)]
    u1, div1 = PU.single_rank_oracle(n_global, cfg, 2)
    u = PU.assemble(str(tmp_path), "u", n_local, pgrid, u1.shape)
    div = PU.assemble(str(tmp_path), "div", n_local, pgrid, div1.shape)
    assert rel(u, u1) < 1e-11
    assert rel(div, div1) < 5e-10


@pytest.mark.parametrize("mode", ["methods", "fused"])
def test_gpu_partition_dealiasing_and_shock_capturing(tmp_path, mode):
    """BASELINE.json configs[4]'s ingredients on a partitioned box (2 ranks): over-integration and shock capturing
    through the mirrored loop and through hfx_stage_partitioned must reproduce the 1-rank run of the same library
    (which the hex_p4_jet / overint / shock fixtures pin against the genuine reference)."""
    import ctypes as C
    import hfx
    import hfx_host as H
    n_local, pgrid = [2, 4, 4], [2, 1, 1]
    cfg = dict(CFG, order=3, riemann_solve_type=3, over_int=1, over_int_order=5, shock_cap=1, s0=1e30, expf_fac=36.0, expf_order=4,
               expf_cutoff=1, shock_det_field=0)
    # sensor threshold inside the widest gap of the initial sensor values: rounding cannot flip an element
    c = H.Case([4, 4, 4], **cfg)
    c.to_device(0)
    e = c.handles()[1]
    hfx.check(hfx.lib().hfx_eles_shock_capture(e))
    sens = np.zeros(c.n_eles)
    hfx.check(hfx.lib().hfx_eles_download(e, C.c_int(hfx.SENSOR), sens.ctypes.data_as(hfx.dp)))
    sens = np.sort(sens)
    c.close()
    ratio = sens[1:] / sens[:-1]
    k = int(np.argmax(ratio))  # some elements are filtered, the others are not
    assert ratio[k] > 1.001  # rounding differences between the runs are ~1e-13
    cfg["s0"] = float(np.sqrt(sens[k] * sens[k + 1]))
    one = H.Case([4, 4, 4], **cfg)
    one.to_device(0)
    one.run(2)
    one.sync_host()
    u1 = one.array("disu_upts0").copy()
    one.close()
    PU.spawn(PU.gpu_worker, 2, (n_local, pgrid, cfg, 2, str(tmp_path), mode))
    u = PU.assemble(str(tmp_path), "u", n_local, pgrid, u1.shape)
    assert rel(u, u1) < 1e-11


@pytest.mark.parametrize("mode", ["methods", "fused"])
def test_gpu_partition_with_boundary_faces(tmp_path, mode):
    """Partition faces and boundary faces in one run (walls on the y sides, periodic and split in x): the mirrored loop
    and hfx_stage_partitioned reproduce the 1-rank run of the same library (pinned by the bdy fixtures)."""
    import hfx_host as H
    P = 0.0008421095852102401 * 286.9 * 300.0
    bcs = [dict(type="isotherm_wall", T_static=310.0, u=3.0), dict(type="adiabat_wall", v=-2.0),
           dict(type="sub_out_char", p_static=P * 0.999)]
    cfg = dict(CFG, order=2, riemann_solve_type=3, bcs=bcs, sides={"y-": 0, "y+": 1, "z+": 2, "z-": 2})
    one = H.Case([4, 4, 4], **cfg)
    one.to_device(0)
    one.run(2)
    one.sync_host()
    u1 = one.array("disu_upts0").copy()
    one.close()
    PU.spawn(PU.gpu_worker, 2, ([2, 4, 4], [2, 1, 1], cfg, 2, str(tmp_path), mode))
    u = PU.assemble(str(tmp_path), "u", [2, 4, 4], [2, 1, 1], u1.shape)
    assert rel(u, u1) < 1e-11


def test_gpu_partition_quads(tmp_path):
    cfg = dict(CFG, dims=2, riemann_solve_type=0)
    n_local, pgrid = [4, 2], [1, 2]
    PU.spawn(PU.gpu_worker, 2, (n_local, pgrid, cfg, 1, str(tmp_path), "methods"))
    u1, div1 = PU.single_rank_oracle([4, 4], cfg, 1)
    u = PU.assemble(str(tmp_path), "u", n_local, pgrid, u1.shape)
    assert rel(u, u1) < 1e-11


@pytest.mark.parametrize("mode", ["methods", "fused", "fused2"])
def test_gpu_partition_invariance_8_ranks(mode):
    """BASELINE.json configs[2]'s decomposition: 8 ranks on a 2x2x2 process grid, three neighbours each.  The ranks are
    threads of this process (a GPU box allows few processes on its card); the exchange hook pulls the segments out of
    the peers' device buffers (partition_util.ThreadTransport)."""
    n_local, pgrid = [2, 2, 2], [2, 2, 2]
    cfg = dict(CFG, riemann_solve_type=3)
    parts = PU.threaded_gpu_run(8, n_local, pgrid, cfg, 2, mode)
    u1, div1 = PU.single_rank_oracle([4, 4, 4], cfg, 2)
    u = PU.assemble_arrays(parts, 0, n_local, pgrid, u1.shape)
    div = PU.assemble_arrays(parts, 1, n_local, pgrid, div1.shape)
    assert rel(u, u1) < 1e-11
    assert rel(div, div1) < 5e-10


@pytest.mark.parametrize("mode", ["methods", "fused", "fused2"])
@pytest.mark.parametrize("n_local,kw", [
    ([3, 4, 3], dict(riemann_solve_type=3, self_partition=[1, 0, 1])),
    ([4, 3, 3], dict(riemann_solve_type=0, order=3, self_partition=[0, 1, 0])),
])
def test_gpu_rccl_transport_self_partition(tmp_path, mode, n_local, kw):
    """libhfx's own transport (hfx_comm_*: grouped ncclSend / ncclRecv on the library's communication stream, ordered
    against the compute stream by events) under the real partition-face kernels, on the one rank a one-GPU box allows:
    the wrap-around faces of one or two periodic directions are partition faces whose neighbour is the rank itself.
    "methods": the mirrored CalcResidual (mpi_inters::send_* / receive_* -> hfx_mpi_inters_send_* / receive_*);
    "fused" / "fused2": hfx_run_steps_partitioned (the whole RK loop inside the library)."""
    cfg = dict(CFG)
    cfg.update(kw)
    PU.spawn(PU.gpu_worker, 1, (n_local, [1, 1, 1], cfg, 2, str(tmp_path), mode, "gloo", "rccl"))
    u1, div1 = PU.single_rank_oracle(n_local, cfg, 2)
    u = PU.assemble(str(tmp_path), "u", n_local, [1, 1, 1], u1.shape)
    div = PU.assemble(str(tmp_path), "div", n_local, [1, 1, 1], div1.shape)
    assert rel(u, u1) < 1e-11
    assert rel(div, div1) < 5e-10


LES = dict(riemann_solve_type=3, LES=1, SGS_model=1, C_s=0.325, filter_ratio=1.0)


@pytest.mark.parametrize("mode", ["methods", "fused"])
def test_gpu_partition_les_third_exchange(tmp_path, mode):
    """LES (WALE) on a partitioned box: the SGS flux at the partition faces is the third message of the stage
    (mpi_inters::send_sgsf_fpts / receive_sgsf_fpts, src/mpi_inters.cpp:339-397, src/solver.cpp:168-178,203-206).  Two ranks
    over the hook transport -- the mirrored CalcResidual, and hfx_stage_partitioned (a block with a closure runs the
    split variant that keeps the gradients) -- against the 1-rank oracle, which the LES fixtures pin against the reference."""
    n_local, pgrid = [2, 4, 4], [2, 1, 1]
    cfg = dict(CFG, **LES)
    PU.spawn(PU.gpu_worker, 2, (n_local, pgrid, cfg, 2, str(tmp_path), mode))
    u1, div1 = PU.single_rank_oracle([4, 4, 4], cfg, 2)
    u = PU.assemble(str(tmp_path), "u", n_local, pgrid, u1.shape)
    assert rel(u, u1) < 1e-11


@pytest.mark.parametrize("mode", ["methods", "fused"])
def test_gpu_rccl_les_self_partition(tmp_path, mode):
    """the same through the library's RCCL transport (hfx_mpi_inters_send / receive_sgsf_fpts; hfx_run_steps_partitioned
    with three messages per stage) on a self-partitioned rank"""
    n_local = [3, 4, 3]
    cfg = dict(CFG, self_partition=[1, 0, 1], **LES)
    PU.spawn(PU.gpu_worker, 1, (n_local, [1, 1, 1], cfg, 2, str(tmp_path), mode, "gloo", "rccl"))
    u1, div1 = PU.single_rank_oracle(n_local, cfg, 2)
    u = PU.assemble(str(tmp_path), "u", n_local, [1, 1, 1], u1.shape)
    assert rel(u, u1) < 1e-11


@pytest.mark.parametrize("mode", ["methods", "fused"])
@pytest.mark.parametrize("model,ftype", [(2, 0), (4, 1)])
def test_gpu_rccl_les_similarity_self_partition(tmp_path, mode, model, ftype):
    """the closures that filter the solution (WALE + similarity, similarity) on partitioned blocks: calc_sgs_terms at the
    first stage of every step -- CalcResidual's own call on the per-method path, phase 1 of hfx_stage_partitioned on the
    fused one -- and the three messages per stage over RCCL"""
    n_local = [3, 4, 3]
    cfg = dict(CFG, self_partition=[1, 0, 1], riemann_solve_type=3, LES=1, SGS_model=model, C_s=0.325, filter_ratio=2.0, filter_type=ftype)
    PU.spawn(PU.gpu_worker, 1, (n_local, [1, 1, 1], cfg, 2, str(tmp_path), mode, "gloo", "rccl"))
    u1, div1 = PU.single_rank_oracle(n_local, cfg, 2)
    u = PU.assemble(str(tmp_path), "u", n_local, [1, 1, 1], u1.shape)
    assert rel(u, u1) < 1e-11


@pytest.mark.parametrize("transport", ["rccl", "torch"])
def test_gpu_partitioned_cfl_time_step(tmp_path, transport):
    """dt_type 1 on the partitioned fused path: calc_time_step at the top of every step (src/HiFiLES.cpp:198), the MIN
    reduction over the ranks through the library's communicator or the caller's hook; equals the undivided per-method run."""
    import hfx_host as H
    n_local = [3, 4, 3]
    cfg = dict(CFG, riemann_solve_type=3, dt_type=1, CFL=0.3, dt=0.0)
    one = H.Case(n_local, **cfg)
    one.to_device(0)
    one.run(2)
    one.sync_host()
    u1 = one.array("disu_upts0").copy()
    one.close()
    PU.spawn(PU.gpu_worker, 1, (n_local, [1, 1, 1], dict(cfg, self_partition=[1, 0, 0]), 2, str(tmp_path), "fused", "gloo", transport))
    u = PU.assemble(str(tmp_path), "u", n_local, [1, 1, 1], u1.shape)
    assert rel(u, u1) < 1e-11


def _time_partitioned_worker(rank, world, port, outdir):
    import torch
    import hfx
    import hfx_host as H
    torch.cuda.set_device(0)
    c = H.Case([4, 4, 4], order=3, amp=0.05, self_partition=[1, 1, 1])
    c.to_device(0)
    c.set_comm(hfx.comm_unique_id())
    t = c.time_partitioned(5)
    comm = hfx.Comm(c.handles()[0], hfx.comm_unique_id(), 1, 0)
    red = comm.allreduce([3.0, -1.5], "min") + comm.allreduce([2.0], "sum")
    comm.close()
    info = c.comm_info()  # what RCCL itself reports: one rank, this device
    assert info["nranks"] == 1 and info["rank"] == 0 and info["device"] == 0 and info["pci_bus_id"], info
    c.close()
    np.save(outdir + "/t.npy", np.array([t[k] for k in sorted(t)] + red))


def test_gpu_time_partitioned_and_allreduce(tmp_path):
    PU.spawn(_time_partitioned_worker, 1, (str(tmp_path),))
    v = np.load(str(tmp_path / "t.npy"))
    assert np.all(v[:8] > 0.0)  # four phases, two exchanges, the stage, the flux kernel on its own
    assert list(v[8:]) == [3.0, -1.5, 2.0]


def _nccl_self_worker(rank, world, port, outdir):
    import torch
    import hfx
    from exchange import Exchange
    torch.cuda.set_device(0)
    dist = PU.init_pg(rank, world, port, "nccl")
    try:
        ctx = hfx.Context(0)
        stream = torch.cuda.ExternalStream(ctx.stream, device=torch.device("cuda", 0))
        nface, rec = 6, 10
        with torch.cuda.stream(stream):
            out = torch.arange(nface * rec, dtype=torch.float64, device="cuda") + 1.0
            inn = torch.zeros(nface * rec, dtype=torch.float64, device="cuda")
            g_out = -out.repeat(3)
            g_in = torch.zeros_like(g_out)
        stream.synchronize()
        ex = Exchange([nface], 0, {0: (out, inn), 1: (g_out, g_in)}, stream=stream, allow_self=True)
        ex(0, 0); ex(1, 0); ex(0, 1); ex(1, 1)
        stream.synchronize()
        ok = bool(torch.equal(inn, out) and torch.equal(g_in, g_out))
        ex.close()
        ctx.close()
        np.save(outdir + "/ok.npy", np.array([ok]))
    finally:
        dist.destroy_process_group()


def test_gpu_exchange_rccl_device_buffers(tmp_path):
    """The RCCL branch of the exchange (batched isend/irecv on device buffer slices, ordered on the hfx
    stream) with the one rank a 1-GPU box allows: every segment is sent to self."""
    PU.spawn(_nccl_self_worker, 1, (str(tmp_path),))
    assert np.load(str(tmp_path / "ok.npy"))[0]


def _fullsize_selfpartition_worker(rank, world, port, outdir):
    import torch
    import hfx
    import hfx_host as H
    torch.cuda.set_device(0)
    nodes = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hex_p4_n32_tgv.npz"))["loc_1d_upts"]
    a = H.Case(32, order=4, loc_1d_upts=nodes)
    a.to_device(0)
    a.run_steps_lib(2, fused=3)
    a.sync_host()
    ua = a.array("disu_upts0").copy()
    a.close()
    b = H.Case(32, order=4, loc_1d_upts=nodes, self_partition=[1, 1, 1])
    b.to_device(0)
    b.set_comm(hfx.comm_unique_id())
    b.run_partitioned(2)
    b.sync_host()
    ub = b.array("disu_upts0")
    err = np.abs(ua - ub).max() / np.abs(ua).max()
    b.close()
    np.save(outdir + "/err.npy", np.array([err, float(np.isfinite(ub).all())]))


def test_gpu_fullsize_selfpartition_equals_undivided(tmp_path):
    """BASELINE.json configs[2]'s per-GPU share (32^3 P4 hexes, all six sides partition faces: 6 144 faces to the rank itself
    over libhfx's RCCL transport, hfx_run_steps_partitioned) against the undivided block (hfx_run_steps, fused 3) after two
    time steps: the partitioned stage is the same computation at the size bench.py --self-partition / --gpus 8 runs it."""
    PU.spawn(_fullsize_selfpartition_worker, 1, (str(tmp_path),))
    err, finite = np.load(str(tmp_path / "err.npy"))
    assert finite == 1.0
    assert err < 1e-11


def _comm_stream_worker(rank, world, port, outdir):
    import ctypes as C
    import torch
    import hfx
    import hfx_host as H
    torch.cuda.set_device(0)
    out = []
    for knob in (1, 0):
        c = H.Case([4, 5, 3], order=3, amp=0.1, riemann_solve_type=3, self_partition=[1, 1, 0])
        c.to_device(0)
        hfx.check(hfx.lib().hfx_ctx_set_option(c.handles()[0], b"comm_stream_faces", C.c_int(knob)))
        c.set_comm(hfx.comm_unique_id())
        c.run_partitioned(3)
        c.sync_host()
        out.append(c.array("disu_upts0").copy())
        c.close()
    np.save(outdir + "/same.npy", np.array([float(np.array_equal(out[0], out[1])), float(np.isfinite(out[0]).all())]))


def test_gpu_partition_face_kernels_on_either_stream(tmp_path):
    """hfx_run_steps_partitioned with the one-sided partition-face kernels on the communication stream (default) and on the
    compute stream (option comm_stream_faces 0): the same kernels on the same data in the same order -- the same bits."""
    PU.spawn(_comm_stream_worker, 1, (str(tmp_path),))
    same, finite = np.load(str(tmp_path / "same.npy"))
    assert finite == 1.0 and same == 1.0


def test_gpu_rccl_dealiasing_shock_and_walls_self_partition(tmp_path):
    """the comm-stream form of hfx_run_steps_partitioned (partition-face kernels beside the interior ones) with everything a
    stage can carry besides: over-integration before the flux kernel, shock capturing after the update, boundary faces --
    a rank that is its own neighbour in x and z, walls in y, against the undivided run of the same library"""
    import ctypes as C
    import hfx
    import hfx_host as H
    n = [3, 4, 3]
    cfg = dict(CFG, order=3, riemann_solve_type=3, over_int=1, over_int_order=5, shock_cap=1, s0=1e30, expf_fac=36.0, expf_order=4,
               expf_cutoff=1, shock_det_field=0, bcs=[dict(type="isotherm_wall", T_static=310.0, u=3.0), dict(type="adiabat_wall", v=-2.0)],
               sides={"y-": 0, "y+": 1})
    c = H.Case(n, **cfg)
    c.to_device(0)
    e = c.handles()[1]
    hfx.check(hfx.lib().hfx_eles_shock_capture(e))
    sens = np.zeros(c.n_eles)
    hfx.check(hfx.lib().hfx_eles_download(e, C.c_int(hfx.SENSOR), sens.ctypes.data_as(hfx.dp)))
    sens = np.sort(sens)
    c.close()
    ratio = sens[1:] / sens[:-1]
    k = int(np.argmax(ratio))
    cfg["s0"] = float(np.sqrt(sens[k] * sens[k + 1]))
    one = H.Case(n, **cfg)
    one.to_device(0)
    one.run_steps_lib(2, fused=3)
    one.sync_host()
    u1 = one.array("disu_upts0").copy()
    one.close()
    PU.spawn(PU.gpu_worker, 1, (n, [1, 1, 1], dict(cfg, self_partition=[1, 0, 1]), 2, str(tmp_path), "fused", "gloo", "rccl"))
    u = PU.assemble(str(tmp_path), "u", n, [1, 1, 1], u1.shape)
    assert rel(u, u1) < 1e-11

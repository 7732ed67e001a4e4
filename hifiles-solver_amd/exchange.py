"""Partition-face exchange: the transport the reference gets from MPI_Isend / MPI_Irecv / MPI_Waitall
(/root/reference/src/mpi_inters.cpp:244-270, 304-332), done with torch.distributed point-to-point ops --
RCCL over xGMI on device buffers (backend "nccl"), or gloo on host tensors (CPU tests; device buffers
are staged through pinned host memory).  Plumbing only: packing and the one-sided flux kernels are libhfx.

The faces shared with one rank are contiguous in the out/in buffers (record = one face), so each
neighbour gets ONE message per exchange: out[start:start+count] -> its in[start':start'+count].
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


class _DevBuf:
    """__cuda_array_interface__ view of a raw device pointer owned by libhfx."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, n, device):
    return torch.as_tensor(_DevBuf(ptr, n), device=device)


def segments(nout_proc, rank, allow_self=False):
    """[(peer, first face sent, first face received, faces)] in buffer order, from the reference's Nout_proc table
    (faces of one rank contiguous, ranks ascending, src/mpi_inters.cpp:244-256)."""
    seg, start = [], 0
    for p, c in enumerate(np.asarray(nout_proc).tolist()):
        if c:
            assert allow_self or p != rank
            seg.append((p, start, start, c))
            start += c
    return seg


class Exchange:
    """out/in tensor pairs per kind (0 solution, 1 corrected gradient, 2 SGS flux); tensors are flat float64.

    CPU tensors go straight through the process group.  Device tensors go through the group directly
    when it is RCCL, and through pinned host staging when it is gloo."""

    def __init__(self, nout_proc, rank, bufs, group=None, stream=None, allow_self=False, seg=None):
        """seg: [(peer, send_first, recv_first, count)] (hfx_host.Case.mpi_segments); default: from nout_proc"""
        self.seg = list(seg) if seg is not None else segments(nout_proc, rank, allow_self)
        self.rank = rank
        self.n_faces = sum(c for _, _, _, c in self.seg)
        self.bufs = bufs  # {kind: (out, in)}
        self.group = group
        self.backend = dist.get_backend(group)
        self.stream = stream  # torch.cuda.ExternalStream of the hfx context (device buffers)
        self.pending = []
        self.stage = {}
        for kind, (o, i) in bufs.items():
            assert o.numel() == i.numel() and (self.n_faces == 0 or o.numel() % self.n_faces == 0)
            if o.is_cuda and self.backend != "nccl":
                self.stage[kind] = (torch.empty(o.numel(), dtype=o.dtype).pin_memory(),
                                    torch.empty(i.numel(), dtype=i.dtype).pin_memory())

    def _rec(self, kind):
        return self.bufs[kind][0].numel() // max(1, self.n_faces)

    def start(self, kind):
        o, i = self.bufs[kind]
        rec = self._rec(kind)
        if not self.seg:
            return
        if o.is_cuda and self.backend == "nccl":
            with torch.cuda.stream(self.stream):
                ops = []
                for p, s, r, c in self.seg:
                    ops.append(dist.P2POp(dist.isend, o[s * rec:(s + c) * rec], p, self.group))
                    ops.append(dist.P2POp(dist.irecv, i[r * rec:(r + c) * rec], p, self.group))
                self.pending.append((kind, dist.batch_isend_irecv(ops)))
            return
        if o.is_cuda:
            so, si = self.stage[kind]
            with torch.cuda.stream(self.stream):
                so.copy_(o, non_blocking=True)
            self.stream.synchronize()
            o, i = so, si
        reqs = []
        for p, s, r, c in self.seg:
            if p == self.rank:  # a rank's faces with itself (self-partition): a local copy stands in for the message
                i[r * rec:(r + c) * rec].copy_(o[s * rec:(s + c) * rec])
            else:
                reqs.append(dist.irecv(i[r * rec:(r + c) * rec], src=p, group=self.group, tag=kind))
        for p, s, r, c in self.seg:
            if p != self.rank:
                reqs.append(dist.isend(o[s * rec:(s + c) * rec], dst=p, group=self.group, tag=kind))
        self.pending.append((kind, reqs))

    def wait(self, kind):
        keep = []
        for k, reqs in self.pending:
            if k != kind:
                keep.append((k, reqs))
                continue
            o, i = self.bufs[k]
            if o.is_cuda and self.backend == "nccl":
                with torch.cuda.stream(self.stream):
                    for r in reqs:
                        r.wait()
            else:
                for r in reqs:
                    r.wait()
                if o.is_cuda:
                    with torch.cuda.stream(self.stream):
                        i.copy_(self.stage[k][1], non_blocking=True)
        self.pending = keep

    def close(self):
        """Drop the staging / aliasing tensors while the hfx context (its stream) is still alive: the
        pinned-memory allocator records an event on every stream a block was used on when it is freed."""
        if self.stream is not None:
            self.stream.synchronize()
        self.stage.clear()
        self.bufs = {}
        self.pending = []

    def __call__(self, kind, phase):
        (self.start if phase == 0 else self.wait)(kind)


def for_case(case, group=None, device=None, projected_flux=False):
    """Exchange wired to the device buffers of a partitioned hfx_host.Case that is on the device.
    projected_flux: kind 1 moves buffers 4/5 (what hfx_stage_partitioned sends in fused mode 3)."""
    import hfx
    L, Rlut, nout = case.mpi_faces()
    h = case.mpi_handle()
    if not h:
        return None
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    les = bool(case.cfg.get("LES", 0))
    if les:
        projected_flux = False  # a block with an LES closure runs the split variant that sends the gradient
    t = [device_tensor(*hfx.mpi_buffer(h, w), device) for w in ((0, 1, 4, 5) if projected_flux else (0, 1, 2, 3))]
    stream = torch.cuda.ExternalStream(case.stream(), device=device)
    p = case.params()
    bufs = {0: (t[0], t[1])}
    if p.viscous:
        bufs[1] = (t[2], t[3])
        if les:  # third message: the physical SGS flux (mpi_inters::send_sgsf_fpts)
            bufs[2] = tuple(device_tensor(*hfx.mpi_buffer(h, w), device) for w in (6, 7))
    return Exchange(nout, case.rank, bufs, group=group, stream=stream, seg=case.mpi_segments())

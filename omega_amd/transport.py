"""Halo transports: how the packed per-neighbour messages of Halo::exchange* travel.

Production: RCCL send/recv over xGMI, issued through torch.distributed (backend "nccl" is
RCCL on ROCm) as ONE grouped batch per exchange on the same HIP stream the pack / unpack
kernels run on -- stream-ordered, no host polling.  The message buffers are torch tensors
owned here and registered with the C++ Halo (omg_halo_set_buffers).

Test mode (backend "gloo"): the same C++ exchange path, but each message is staged through
host memory, so a 2-rank exchange can be exercised on a one-GPU box (both ranks on one GPU)
and on CPU-only hosts with host buffers.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class TorchTransport:
    def __init__(self, halo, per_cell: int, per_edge: int, per_vertex: int = 0, device="cuda", stream=None):
        """per_cell / per_edge / per_vertex: doubles per element in the largest exchange."""
        self.halo = halo
        self.tasks = halo.neighbors
        self.backend = dist.get_backend()
        self.device = torch.device(device)
        self.stream = stream  # torch.cuda.Stream the library's kernels run on (None = default)
        self.send, self.recv = [], []
        for i, _ in enumerate(self.tasks):
            nbytes = max(halo.required_bytes(i, per_cell, per_edge, per_vertex), 8)
            s = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            r = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            self.send.append(s)
            self.recv.append(r)
            halo.set_buffers(i, s.data_ptr(), r.data_ptr(), nbytes)
        halo.set_transport(self._exchange)

    def _exchange(self, tasks, send_ptrs, send_bytes, recv_ptrs, recv_bytes, stream_handle):
        direct = self.backend == "nccl"
        ops, staged = [], []
        # the C++ side says which HIP stream the pack / unpack kernels of THIS exchange run on (the stepper's
        # communication stream when the exchange is overlapped with interior compute): RCCL must order on it
        own = None if self.stream is None else int(self.stream.cuda_stream)
        if direct and stream_handle and int(stream_handle) != own:
            ctx = torch.cuda.stream(torch.cuda.ExternalStream(int(stream_handle)))
        else:   # the stream this transport was created for (sequential exchanges), or the default stream
            ctx = torch.cuda.stream(self.stream) if (self.stream is not None) else _Null()
        with ctx:
            for i, t in enumerate(tasks):
                assert send_ptrs[i] == self.send[i].data_ptr() and recv_ptrs[i] == self.recv[i].data_ptr()
                if direct:
                    if recv_bytes[i]:
                        ops.append(dist.P2POp(dist.irecv, self.recv[i][: recv_bytes[i]], t))
                    if send_bytes[i]:
                        ops.append(dist.P2POp(dist.isend, self.send[i][: send_bytes[i]], t))
                else:
                    hs = self.send[i][: send_bytes[i]].cpu()  # synchronises with the pack kernels
                    hr = torch.empty(recv_bytes[i], dtype=torch.uint8)
                    staged.append((i, hr))
                    if recv_bytes[i]:
                        ops.append(dist.P2POp(dist.irecv, hr, t))
                    if send_bytes[i]:
                        ops.append(dist.P2POp(dist.isend, hs, t))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()  # nccl: makes the current stream wait; gloo: blocks the host
            for i, hr in staged:
                if hr.numel():
                    self.recv[i][: hr.numel()].copy_(hr)
        return 0


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

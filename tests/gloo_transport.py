"""Halo wire for test rigs (test infrastructure: not part of the omega_amd package).

The production wire is inside the library: `omega_amd.RcclComm` + `Halo.use_rccl` (omega_amd/csrc/Rccl.cpp:
grouped ncclSend / ncclRecv over xGMI on the exchange's HIP stream, no Python on the path).

`GlooStagedTransport` below exists so that the same C++ exchange path (job-table pack kernel -> wire -> unpack
kernel, RK4 overlap on the communication stream) can be exercised with several ranks on a ONE-GPU box, where RCCL
refuses two ranks on one device: each message is staged through host memory and travels over torch.distributed's
gloo backend.  It synchronises the stream the exchange runs on before reading the send buffer and finishes the
host-to-device copies before returning, so it is correct on any stream (blocking or not) -- and serialises the
exchange, which is fine for a correctness rig.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

import omega_amd as oa


class GlooStagedTransport:
    def __init__(self, halo):
        assert dist.get_backend() == "gloo", "GlooStagedTransport is the host-staged test wire; production uses RcclComm"
        self.halo = halo
        halo.set_transport(self._exchange)

    def _exchange(self, tasks, send_ptrs, send_bytes, recv_ptrs, recv_bytes, stream_handle):
        L = oa.lib()
        # the pack kernel of this exchange was queued on `stream_handle` (0 / None = the default stream)
        oa._chk(L.omg_stream_synchronize(C.c_void_p(stream_handle) if stream_handle else None))
        ops, staged = [], []
        for i, t in enumerate(tasks):
            hs = np.empty(send_bytes[i], dtype=np.uint8)
            if send_bytes[i]:
                oa._chk(L.omg_copy_to_host(hs.ctypes.data_as(C.c_void_p), C.c_void_p(send_ptrs[i]), C.c_size_t(send_bytes[i])))
            hr = np.empty(recv_bytes[i], dtype=np.uint8)
            staged.append((i, hr))
            if recv_bytes[i]:
                ops.append(dist.P2POp(dist.irecv, torch.from_numpy(hr), t))
            if send_bytes[i]:
                ops.append(dist.P2POp(dist.isend, torch.from_numpy(hs), t))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for i, hr in staged:
            if hr.size:   # synchronous copy: complete before the unpack kernel is queued
                oa._chk(L.omg_copy_to_device(C.c_void_p(recv_ptrs[i]), hr.ctypes.data_as(C.c_void_p), C.c_size_t(hr.size)))
        return 0
